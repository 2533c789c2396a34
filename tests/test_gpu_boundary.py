"""GPU tests of the drop-in surface around the solver: the functions a maintainer of the reference would rebind
(`project_points`, `_apply_bundle_adjustment`, the verbose table, `load_calibration_data`) and the calling
conventions of include/sfmba.h (threads, int64 pixels).  Reference lines are cited per test."""
import contextlib
import io
import json
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import ba_oracle
    return ba_oracle


def test_project_points_matches_oracle_and_reference_fixtures(orc):
    """sfmba.project_points <-> /root/reference/sfm_lite/bundle_adjustment.py:20-32 (one camera row per point).
    Checked against the oracle's per-observation loop and, through residual + uv, against every residual
    fixture captured from the reference."""
    import sfmba
    rng = np.random.default_rng(5)
    n = 257
    pts = rng.normal(size=(n, 3)) + np.array([0.0, 0.0, 10.0])
    cams = np.hstack([rng.normal(0, 0.3, (n, 3)), rng.normal(0, 0.5, (n, 3))])
    cams[0, :3] = 0.0                                    # theta = 0: series branch
    cams[1, :3] = np.array([np.pi, 0.0, 0.0]) * (1 - 1e-9)
    for K in (sfmba.K_SCEAUX, np.array([[1000.0, 2.5, 500.0], [0.1, 990.0, 400.0], [1e-4, -2e-4, 1.0]])):
        got = sfmba.project_points(pts, cams, K)
        ref = np.asarray(orc.project_points_loop(pts, cams, K))
        assert got.shape == (n, 2)
        assert np.abs(got - ref).max() <= 1e-11 * max(3000.0, np.abs(ref).max())
    g = np.load(os.path.join(GOLDEN, "residual_cases.npz"))
    for k in range(int(g["n_cases"])):
        pre = f"c{k:02d}_"
        C, P, N = (int(v) for v in g[pre + "dims"])
        x, ci, pi = g[pre + "x"], g[pre + "ci"], g[pre + "pi"]
        cam_rows = x[:6 * C].reshape(C, 6)[ci]
        pt_rows = x[6 * C:].reshape(P, 3)[pi]
        proj = sfmba.project_points(pt_rows, cam_rows, g[pre + "K"])
        ref = g[pre + "r"].reshape(N, 2) + g[pre + "uv"]           # reference residual + pixels = its projection
        assert np.abs(proj - ref).max() <= 1e-11 * max(3000.0, np.abs(ref).max()), str(g[pre + "tag"])


def _synthetic_graph(seed=3, n_nodes=7, n_points=60):
    """What SFM._apply_bundle_adjustment reads from its graph (sfm.py:248-256): 4x4 poses, registered flags,
    the cloud, and (point, camera, pixel) tuples in Graph.pt3ds_pt2ds order (graph.py:186-191), with node 2
    unregistered and one (camera, point) pair seen twice (track sets merge, graph.py:86)."""
    import sfmba
    from sfmba import api
    rng = np.random.default_rng(seed)
    registered = [True] * n_nodes
    registered[2] = False
    H_list = []
    for k in range(n_nodes):
        H = np.eye(4)
        H[:3, :3] = api._matrix_from_rotvec(rng.normal(0, 0.15, 3))
        H[:3, 3] = rng.normal(0, 0.4, 3)
        H_list.append(H)
    X3d = rng.normal(size=(n_points, 3)) + np.array([0.0, 0.0, 9.0])
    K = sfmba.K_SCEAUX
    obs = []
    for p in range(n_points):
        cams = [c for c in rng.permutation(n_nodes)[:rng.integers(2, n_nodes)] if registered[c]]
        if p == 5:
            cams = cams + [cams[0]]                      # duplicated (camera, point) pair
        for c in sorted(cams):
            R, T = H_list[c][:3, :3], H_list[c][:3, 3]
            q = K @ (R @ (X3d[p] - T))                   # the BA model: T is the camera centre (bundle_adjustment.py:27)
            uv = np.trunc(q[:2] / q[2] + rng.normal(0, 0.5, 2)).astype(np.int64)     # integer pixels, graph.py:112
            obs.append((p, c, uv))
    H0 = [H.copy() for H in H_list]
    for k in range(n_nodes):                             # perturbed start
        H0[k][:3, :3] = api._matrix_from_rotvec(api._rotvec_from_matrix(H_list[k][:3, :3]) + rng.normal(0, 0.01, 3))
        H0[k][:3, 3] += rng.normal(0, 0.01, 3)
    return H0, registered, X3d + rng.normal(0, 0.01, X3d.shape), obs, K


def test_apply_bundle_adjustment_end_to_end(orc):
    """sfmba.apply_bundle_adjustment <-> SFM._apply_bundle_adjustment (/root/reference/sfm_lite/sfm.py:243-281),
    against the oracle's pack_problem -> trf_schur -> unpack_result on the same graph."""
    import sfmba
    H0, registered, X0, obs, K = _synthetic_graph()
    x0, n_cam, n_pts, ci, pi, uv, cmap = orc.pack_problem(H0, registered, X0, obs)
    px0, pn_cam, pn_pts, pci, ppi, puv, pcmap = sfmba.pack_cameras_points(H0, registered, X0, obs)
    assert (pn_cam, pn_pts, pcmap) == (n_cam, n_pts, cmap) and n_cam == 6
    assert np.array_equal(pci, ci) and np.array_equal(ppi, pi) and np.array_equal(puv, uv)
    assert np.abs(px0 - x0).max() < 1e-13
    o = orc.trf_schur(x0, n_cam, n_pts, ci, pi, uv, K, ftol=1e-10, linear="pcg", pcg_tol=1e-3, precond="schur_exact")     # 6 cameras: PCG in LDS
    H_ref, X_ref = orc.unpack_result(o.x, n_cam, n_pts, cmap, H0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        H_new, X_new, res = sfmba.apply_bundle_adjustment(H0, registered, X0, obs, K, tol=1e-10, verbose=2)
    assert "Iteration" in buf.getvalue() and "termination condition is satisfied" in buf.getvalue()
    assert (res.status, res.nfev, res.njev) == (o.status, o.nfev, o.njev)
    assert abs(res.cost - o.cost) <= 1e-9 * o.cost
    assert np.array_equal(H_new[2], H0[2])                        # the unregistered node keeps its pose
    for k in range(len(H0)):
        if registered[k]:
            R = H_new[k][:3, :3]
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and np.array_equal(H_new[k][3], [0, 0, 0, 1])
    # No camera is held fixed (sfm.py:264), so poses and cloud are determined up to a similarity transform, and the
    # component of a step along those 7 directions is rounding noise divided by the damping term: it differs between
    # two factorisations of the same matrix.  What is determined is compared: the reprojection of every observation.
    xg = orc.pack_problem(H_new, registered, X_new, obs)[0]
    xr = orc.pack_problem(H_ref, registered, X_ref, obs)[0]
    rg = orc.compute_residuals(xg, n_cam, n_pts, ci, pi, uv, K)
    rr = orc.compute_residuals(xr, n_cam, n_pts, ci, pi, uv, K)
    assert np.abs(rg - rr).max() < 1e-5 and np.abs(rg - res.fun).max() < 1e-8
    assert np.abs(H_new[0] - H_ref[0]).max() < 0.1 and np.abs(X_new - X_ref).max() < 0.5        # same basin


def test_verbose_table_matches_scipy_format():
    """verbose=2 (forwarded at sfm.py:266): header and rows are scipy's print_header_nonlinear /
    print_iteration_nonlinear (SCIPY/optimize/_lsq/common.py:545-563) character for character, reach Python's
    sys.stdout in order with the verbose>=1 summary (least_squares.py:966-970), and row 0 carries the values of
    the recorded scipy run on the same problem."""
    import sfmba
    from scipy.optimize._lsq import common as sc_common
    rec = json.load(open(os.path.join(GOLDEN, "scipy_cfg2_run.json")))
    pb = sfmba.make_config("cfg2")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args, verbose=2)
    lines = buf.getvalue().splitlines()
    hdr = io.StringIO()
    with contextlib.redirect_stdout(hdr):
        sc_common.print_header_nonlinear()
    assert lines[0] == hdr.getvalue().rstrip("\n")
    rows = lines[1:-2]
    assert len(rows) == res.iterations + 1 and all(len(r) == 90 for r in rows)
    assert lines[-2] == res.message
    assert lines[-1] == (f"Function evaluations {res.nfev}, initial cost {res.cost0:.4e}, final cost "
                         f"{res.cost:.4e}, first-order optimality {res.optimality:.2e}.")
    for k, row in enumerate(rows):
        cols = [row[15 * j:15 * (j + 1)].strip() for j in range(6)]
        it, nfev, cost, opt = int(cols[0]), int(cols[1]), float(cols[2]), float(cols[5])
        red = float(cols[3]) if cols[3] else None
        step = float(cols[4]) if cols[4] else None
        assert it == k and (k == 0) == (red is None) == (step is None)
        again = io.StringIO()
        with contextlib.redirect_stdout(again):
            sc_common.print_iteration_nonlinear(it, nfev, cost, red, step, opt)
        assert again.getvalue().rstrip("\n") == row               # same widths, centring and number formats
    # iteration 0 is evaluated at x0 by both: same cost and first-order optimality as scipy recorded
    t0 = rec["table"][0]
    c0 = [rows[0][15 * j:15 * (j + 1)].strip() for j in range(6)]
    assert (int(c0[0]), int(c0[1])) == (int(t0[0]), int(t0[1]))
    assert abs(float(c0[2]) - t0[2]) <= 1e-3 * t0[2] and abs(float(c0[5]) - t0[3]) <= 1e-2 * t0[3]
    assert float(rows[-1][30:45]) <= rec["cost"] * (1 + 1e-4)


def test_solve_from_worker_thread_while_main_thread_holds_a_backend():
    """The reference's GUI runs BA in a threading.Thread (/root/reference/app.py:80-85,109); include/sfmba.h:
    one handle per thread, any thread may call.  The main thread keeps its own Backend (with a problem set)
    while a worker solves; both results equal the single-threaded ones."""
    import sfmba
    pb_main = sfmba.make_problem(5, 40, 300, seed=4)
    pb_work = sfmba.make_problem(6, 80, 500, seed=12)
    ref_main = sfmba.least_squares(sfmba.compute_residuals, pb_main.x0, x_scale="jac", ftol=1e-10, method="trf",
                                   args=pb_main.args)
    ref_work = sfmba.least_squares(sfmba.compute_residuals, pb_work.x0, x_scale="jac", ftol=1e-10, method="trf",
                                   args=pb_work.args)
    main_be = sfmba.get_backend(0)
    main_be.set_problem(*pb_main.args)
    out = {}

    def worker():
        try:
            out["be_id"] = id(sfmba.get_backend(0))
            out["res"] = sfmba.least_squares(sfmba.compute_residuals, pb_work.x0, x_scale="jac", ftol=1e-10,
                                             method="trf", args=pb_work.args)
        except Exception as exc:                                   # noqa: BLE001
            out["exc"] = exc

    t = threading.Thread(target=worker)
    t.start()
    r_main = main_be.residuals(pb_main.x0)                        # the main thread's handle is used meanwhile
    t.join(timeout=120)
    assert not t.is_alive() and "exc" not in out, out.get("exc")
    assert out["be_id"] != id(main_be)                            # a handle of its own
    assert np.array_equal(out["res"].x, ref_work.x) and out["res"].cost == ref_work.cost
    again = sfmba.least_squares(sfmba.compute_residuals, pb_main.x0, x_scale="jac", ftol=1e-10, method="trf",
                                args=pb_main.args)
    assert np.array_equal(again.x, ref_main.x)
    assert np.abs(r_main - sfmba.compute_residuals(pb_main.x0, *pb_main.args)).max() == 0.0


def test_load_calibration_data(tmp_path):
    """sfmba.load_calibration_data <-> /root/reference/sfm_lite/utils.py:24-35: 3x3 whitespace-separated text,
    shape assert included."""
    import sfmba
    f = tmp_path / "K.txt"
    f.write_text("2905.88 0 1416\n0 2905.88 1064\n0 0 1\n")
    K = sfmba.load_calibration_data(str(f))
    assert K.dtype == np.float64 and np.array_equal(K, sfmba.K_SCEAUX)
    pb = sfmba.make_problem(3, 8, 20, seed=0)
    r = sfmba.compute_residuals(pb.x0, pb.n_cameras, pb.n_points, pb.camera_indices, pb.point_indices, pb.points_2d, K)
    assert np.array_equal(r, sfmba.compute_residuals(pb.x0, *pb.args))
    bad = tmp_path / "bad.txt"
    bad.write_text("1 2 3\n4 5 6\n")
    with pytest.raises(AssertionError):
        sfmba.load_calibration_data(str(bad))


def test_int64_pixels_entry_equals_float_entry():
    """sfmba_set_problem_i64 (the reference's own int64 pixel arrays, graph.py:112-113) and sfmba_set_problem
    (float64 pixels) are the same problem: bitwise equal residuals, Jacobian, blocks and solve."""
    import sfmba
    pb = sfmba.make_problem(7, 90, 700, seed=8)
    assert pb.points_2d.dtype == np.int64
    a, b = sfmba.Backend(0), sfmba.Backend(0)
    try:
        a.set_problem(pb.n_cameras, pb.n_points, pb.camera_indices, pb.point_indices, pb.points_2d, pb.K)
        b.set_problem(pb.n_cameras, pb.n_points, pb.camera_indices, pb.point_indices,
                      pb.points_2d.astype(np.float64), pb.K)
        for u, v in zip(a.residual_jacobian(pb.x0), b.residual_jacobian(pb.x0)):
            assert np.array_equal(u, v)
        for u, v in zip(a.normal_blocks(pb.x0), b.normal_blocks(pb.x0)):
            assert np.array_equal(u, v)
        opt = a.default_options()
        opt.ftol = 1e-10
        xa, ra, fa, ga = a.solve(pb.x0, opt)
        xb, rb, fb, gb = b.solve(pb.x0, opt)
        assert np.array_equal(xa, xb) and ra.cost == rb.cost and np.array_equal(fa, fb) and np.array_equal(ga, gb)
    finally:
        a.close()
        b.close()


def _stage_problem(pb, cams_registered, n_points):
    """The least_squares arguments SFM._apply_bundle_adjustment would build (sfm.py:248-262) when only
    `cams_registered` (node ids) are registered and the cloud holds the first `n_points` points of `pb`: observations
    of unregistered cameras are skipped (graph.py:186-191), registered cameras are compacted in node order
    (sfm.py:251-256), points without any remaining observation are dropped from the cloud."""
    C = pb.n_cameras
    reg = sorted(cams_registered)
    cmap = -np.ones(C, dtype=np.int64)
    cmap[reg] = np.arange(len(reg))
    keep = (pb.point_indices < n_points) & (cmap[pb.camera_indices] >= 0)
    pts_seen = np.unique(pb.point_indices[keep])
    pmap = -np.ones(pb.n_points, dtype=np.int64)
    pmap[pts_seen] = np.arange(len(pts_seen))
    ci, pi, uv = cmap[pb.camera_indices[keep]], pmap[pb.point_indices[keep]], pb.points_2d[keep]
    cams = pb.x0[:6 * C].reshape(C, 6)[reg]
    pts = pb.x0[6 * C:].reshape(-1, 3)[pts_seen]
    return np.concatenate([cams.ravel(), pts.ravel()]), (len(reg), len(pts_seen), ci, pi, uv, pb.K)


def test_growing_reconstruction_reuses_the_previous_problem():
    """SURVEY.md section 8f-3.  The reference runs BA after every fused edge on a growing problem
    (/root/reference/sfm_lite/sfm.py:59-71).  While cameras are still being registered the compact camera indices
    of existing observations shift (sfm.py:251-256) and little can be re-used; once all are registered an edge
    only appends points and observations, and the whole previous problem is re-used (only the tail crosses PCIe).
    One handle is taken through the whole sequence; every stage must equal, bit for bit, what a fresh handle
    computes."""
    import sfmba
    pb = sfmba.make_problem(8, 600, 5000, seed=17)
    order = [0, 5, 2, 7, 1, 6, 3, 4]                               # node ids in the order they get registered
    stages = [(order[:k], 300) for k in range(2, 9)] + [(order, n) for n in (350, 420, 421, 500, 600)]
    inc = sfmba.Backend(0)
    opt = inc.default_options()
    opt.ftol = 1e-10
    prev_n = None
    appended = 0
    try:
        for k, (cams, npts) in enumerate(stages):
            x0, args = _stage_problem(pb, cams, npts)
            inc.set_problem(*args)
            reused, uploaded = inc.problem_reuse()
            n_obs = len(args[2])
            if len(cams) == 8 and k > 7:                           # all cameras registered before this stage: append
                assert reused == prev_n and uploaded <= (n_obs - prev_n) + 256
                appended += 1
            fresh = sfmba.Backend(0)
            try:
                fresh.set_problem(*args)
                assert fresh.problem_reuse()[0] == 0
                for u, v in zip(inc.residual_jacobian(x0), fresh.residual_jacobian(x0)):
                    assert np.array_equal(u, v)
                for u, v in zip(inc.normal_blocks(x0), fresh.normal_blocks(x0)):
                    assert np.array_equal(u, v)
                xa, ra, fa, ga = inc.solve(x0, opt)
                xb, rb, fb, gb = fresh.solve(x0, opt)
                assert np.array_equal(xa, xb) and ra.cost == rb.cost and (ra.nfev, ra.status) == (rb.nfev, rb.status)
                assert np.array_equal(fa, fb) and np.array_equal(ga, gb)
                assert ra.status > 0
            finally:
                fresh.close()
            prev_n = n_obs
        assert appended == 4
        # the same arrays again: everything is re-used; a changed pixel in the middle: re-use up to it
        x0, args = _stage_problem(pb, order, 600)
        inc.set_problem(*args)
        assert inc.problem_reuse() == (prev_n, (prev_n + 255) // 256 * 256 - prev_n)
        uv2 = args[4].copy()
        uv2[1234, 0] += 1
        inc.set_problem(args[0], args[1], args[2], args[3], uv2, args[5])
        assert inc.problem_reuse()[0] == 1234
        fresh = sfmba.Backend(0)
        try:
            fresh.set_problem(args[0], args[1], args[2], args[3], uv2, args[5])
            assert np.array_equal(inc.residuals(x0), fresh.residuals(x0))
            # fp32 storage after fp64 on the same handle: nothing of the fp64 arrays is re-used
            inc.set_precision(32)
            inc.set_problem(*args)
            assert inc.problem_reuse()[0] == 0
            fresh.set_precision(32)
            fresh.set_problem(*args)
            assert np.array_equal(inc.residuals(x0), fresh.residuals(x0))
        finally:
            fresh.close()
    finally:
        inc.close()


def test_result_fun_and_grad_stay_on_the_device_until_read(orc):
    """least_squares returns scipy's OptimizeResult with `fun` (16 MB at 1M observations) and `grad` still in the
    handle's buffers: the reference reads result.x only (sfm.py:271,281).  They must behave like scipy's arrays
    whenever they ARE read -- at once, after the next call on the same handle has overwritten the buffers, through
    every dict access path -- and cost no download when the result is dropped unread."""
    import gc
    import pickle
    import sfmba
    be = sfmba.get_backend(0)
    pa = sfmba.make_problem(5, 60, 300, seed=3)
    pb = sfmba.make_problem(7, 90, 500, seed=4)

    def run(p):
        return sfmba.least_squares(sfmba.compute_residuals, p.x0, x_scale="jac", ftol=1e-10, method="trf", args=p.args)

    n0 = getattr(be, "n_lazy_downloads", 0)
    ra = run(pa)
    assert getattr(be, "n_lazy_downloads", 0) == n0                  # nothing downloaded yet
    fa = ra.fun                                                      # read at once
    assert getattr(be, "n_lazy_downloads", 0) == n0 + 1
    assert np.abs(fa - orc.compute_residuals(ra.x, *pa.args)).max() < 1e-8
    assert abs(ra.optimality - np.abs(ra.grad).max()) <= 1e-12 * max(1.0, ra.optimality)
    # kept alive across the next call: downloaded right before the buffers are overwritten
    rb = run(pb)
    rc = run(pa)
    assert getattr(be, "n_lazy_downloads", 0) == n0 + 2              # rb was flushed by the call that made rc
    assert np.abs(rb["fun"] - orc.compute_residuals(rb.x, *pb.args)).max() < 1e-8
    assert rb.fun.shape == (2 * pb.n_obs,) and rc.fun.shape == (2 * pa.n_obs,)
    assert np.array_equal(rc.fun, fa) and np.array_equal(rc.x, ra.x)  # same input, same bits
    # dropped unread: no download
    n1 = getattr(be, "n_lazy_downloads", 0)
    rd = run(pb)
    xd = rd.x.copy()
    del rd
    gc.collect()
    re_ = run(pb)
    assert getattr(be, "n_lazy_downloads", 0) == n1
    # the other access paths
    d = dict(re_.items())
    assert d["fun"].shape == (2 * pb.n_obs,) and "grad" in re_.keys() and re_.get("fun") is d["fun"]
    assert np.array_equal(re_.x, xd)
    rf = pickle.loads(pickle.dumps(run(pa)))
    assert np.array_equal(rf.fun, fa) and rf.status == ra.status
    assert "fun:" in repr(run(pa))


def test_in_out_and_separate_array_forms_of_the_solve_agree():
    """include/sfmba.h: sfmba_solve(h, x_inout, ...) is sfmba_solve_from(h, x0, x_out, ...) with both pointers equal; the
    start array of the separate form is left untouched (scipy's least_squares does not modify its x0 either)."""
    import ctypes as C
    import sfmba
    from sfmba import _capi
    pb = sfmba.make_problem(30, 400, 3000, seed=4)
    be = sfmba.Backend(0)
    try:
        be.set_problem(*pb.args)
        opt = be.default_options()
        opt.ftol = 1e-10
        x0 = pb.x0.copy()
        x_sep, res_sep, _, _ = be.solve(x0, opt, want_fun=False, want_grad=False)          # separate arrays
        assert np.array_equal(x0, pb.x0)
        x_io = pb.x0.copy()
        res_io = _capi.Result()
        assert be._lib.sfmba_solve(be._h, _capi.ptr(x_io), C.byref(opt), C.byref(res_io)) == 0      # in/out
        assert np.array_equal(x_io, x_sep)
        assert (res_io.status, res_io.nfev, res_io.njev, res_io.cost) == (res_sep.status, res_sep.nfev, res_sep.njev, res_sep.cost)
        null = _capi.Result()
        assert be._lib.sfmba_solve_from(be._h, _capi.ptr(x0), None, C.byref(opt), C.byref(null)) == -1      # x_out is NULL
    finally:
        be.close()


def test_camera_major_order_sorted_on_the_device_equals_the_hosts():
    """set_problem builds the camera-major order with a stable counting sort ON THE DEVICE (k_cam_hist / k_cam_offsets /
    k_cam_scatter; the reference hands a new problem to every call, sfm.py:59-71) instead of sorting on the host and
    uploading the permutation (debug option cm_device = 0).  Same order entry for entry, so every per-camera sum keeps
    its summation order: normal blocks, the Schur product and whole solves are equal to the BIT -- with duplicated
    (camera, point) pairs, cameras nobody observes, cameras held still, fp32 storage, the XCD-aware chunk table of
    pass B (built by k_xcd_chunks from the device-side order) and more observations than one slice batch."""
    import sfmba
    cases = [(sfmba.make_problem(7, 60, 400, seed=3), (), 64, -1), (sfmba.make_problem(300, 4000, 30000, seed=3), (), 64, -1),
             (sfmba.drop_observations(sfmba.make_problem(40, 400, 5000, seed=3), cameras=(5, 39)), (0, 17), 64, 1),
             (sfmba.make_problem(1300, 4000, 70000, seed=21), (), 32, 1), (sfmba.make_problem(3, 8, 20, seed=0), (1,), 64, -1)]
    for pb, fixed, bits, xcd in cases:
        outs = []
        for dev in (0, 1):                                        # (1: forced; the default takes the device path from 64k observations)
            be = sfmba.Backend(0)
            try:
                be.debug_option("cm_device", dev)
                be.debug_option("xcd_chunks", xcd)
                be.debug_option("dense", 0)
                be.set_precision(bits)
                be.set_fixed_cameras(fixed)
                be.set_problem(*pb.args)
                U, V, gc, gp = be.normal_blocks(pb.x0)
                C, P = pb.n_cameras, pb.n_points
                rng = np.random.default_rng(2)
                y = be.schur_matvec(pb.x0, np.full(6 * C, 10.0), np.full(3 * P, 10.0), rng.normal(size=6 * C))
                opt = be.default_options()
                opt.ftol = 1e-10
                x, res, fun, grad = be.solve(pb.x0, opt)
                outs.append((U, gc, y, x, fun, grad, res.cost, res.nfev, res.pcg_iterations))
            finally:
                be.close()
        a, b = outs
        for u, v in zip(a[:6], b[:6]):
            assert np.array_equal(u, v)
        assert a[6:] == b[6:]


def test_packed_upload_of_the_observation_arrays_changes_nothing():
    """set_problem sends the observation arrays over PCIe packed when they allow it -- camera indices as uint16, integer
    pixels as int16 pairs, no point indices (they follow from the run offsets): 6 bytes per observation instead of 24 --
    and expands them on the device (k_unpack_obs, k_expand_pt_idx).  Same residuals, blocks and solves to the BIT as the
    plain upload (debug option packed_upload = 0); pixels that are no int16 integers (float pixels, a pixel of 40000) take
    the plain path by themselves; growing problems keep their prefix in either form."""
    import sfmba
    base = sfmba.make_problem(40, 400, 5000, seed=3)
    tiny = sfmba.make_problem(3, 8, 20, seed=0)
    far = base.points_2d.copy()
    far[17, 0] = 40000
    cases = [(base.args, 64, base.x0), (base.args, 32, base.x0),
             (base.args[:4] + (base.points_2d.astype(np.float64) + 0.25, base.K), 64, base.x0),
             (base.args[:4] + (far, base.K), 64, base.x0), (tiny.args, 64, tiny.x0)]
    for args, bits, x0 in cases:
        outs = []
        for packed in (0, 1):                                     # (1: forced; by default problems under 64k observations upload plain)
            be = sfmba.Backend(0)
            try:
                be.debug_option("packed_upload", packed)
                be.debug_option("dense", 0)
                be.set_precision(bits)
                # a shorter problem first: the second call re-uses its prefix on the device
                n_head = int(np.searchsorted(args[3], args[1] // 2, side="left"))
                be.set_problem(args[0], args[1], args[2][:n_head], args[3][:n_head], args[4][:n_head], args[5])
                be.set_problem(*args)
                assert be.problem_reuse()[0] >= n_head
                r = be.residuals(x0)
                U, V, gc, gp = be.normal_blocks(x0)
                opt = be.default_options()
                opt.ftol = 1e-10
                x, res, fun, grad = be.solve(x0, opt)
                be.set_problem(*args)                                 # once more, unchanged: nothing uploaded
                x2, res2, _, _ = be.solve(x0, opt)
                assert np.array_equal(x, x2)
                outs.append((r, U, V, x, fun, res.cost, res.nfev))
            finally:
                be.close()
        a, b = outs
        for u, v in zip(a[:5], b[:5]):
            assert np.array_equal(u, v)
        assert a[5:] == b[5:]
