"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle and the fixtures
captured from the reference.  Tolerances are stated where used; the arithmetic is fp64 everywhere.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import sfmba
    b = sfmba.Backend(0)
    yield b
    b.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import ba_oracle
    return ba_oracle


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))


def _upper(M):
    """(B,k,k) symmetric -> (B, k(k+1)/2) row-major upper triangle."""
    k = M.shape[1]
    iu = np.triu_indices(k)
    return M[:, iu[0], iu[1]]


# ---- A1/A2: residuals ----------------------------------------------------------------------------

def test_residuals_match_reference_fixtures(be):
    g = np.load(os.path.join(GOLDEN, "residual_cases.npz"))
    for k in range(int(g["n_cases"])):
        pre = f"c{k:02d}_"
        C, P, N = (int(v) for v in g[pre + "dims"])
        be.set_problem(C, P, g[pre + "ci"], g[pre + "pi"], g[pre + "uv"], g[pre + "K"])
        r = be.residuals(g[pre + "x"])
        ref = g[pre + "r"]
        # 1e-11 of the pixel magnitude: fp64 with fma contraction vs numpy's unfused ops
        tol = 1e-11 * max(1.0, float(np.abs(ref).max()), 3000.0)
        assert r.shape == ref.shape
        assert np.abs(r - ref).max() <= tol, str(g[pre + "tag"])


def test_residuals_any_observation_order(be, orc):
    from sfmba import make_problem
    pb = make_problem(5, 40, 300, seed=4)
    perm = np.random.default_rng(0).permutation(300)
    ci, pi, uv = pb.camera_indices[perm], pb.point_indices[perm], pb.points_2d[perm]
    be.set_problem(5, 40, ci, pi, uv, pb.K)
    r = be.residuals(pb.x0)
    ref = orc.compute_residuals(pb.x0, 5, 40, ci, pi, uv, pb.K)
    assert np.abs(r - ref).max() < 1e-8
    r2, Jc, Jp = be.residual_jacobian(pb.x0)
    _, Jc_o, Jp_o = orc.jacobian_blocks(pb.x0, 5, 40, ci, pi, uv, pb.K)
    assert np.abs(r2 - ref).max() < 1e-8
    assert _rel(Jc, Jc_o) < 1e-11 and _rel(Jp, Jp_o) < 1e-11


# ---- A6: Jacobian ----------------------------------------------------------------------------------

def test_jacobian_vs_oracle_and_reference_fd(be, orc):
    from sfmba import make_problem
    g = np.load(os.path.join(GOLDEN, "jacobian_fd_cases.npz"))
    for k in range(int(g["n_cases"])):
        pre = f"j{k}_"
        C, P, N = (int(v) for v in g[pre + "dims"])
        pb = make_problem(C, P, N, seed=int(g[pre + "seed"]))
        be.set_problem(*pb.args)
        r, Jc, Jp = be.residual_jacobian(pb.x0)
        r_o, Jc_o, Jp_o = orc.jacobian_blocks(pb.x0, *pb.args)
        assert _rel(Jc, Jc_o) < 1e-12 and _rel(Jp, Jp_o) < 1e-12         # analytic vs analytic
        J = orc.jacobian_csr(Jc, Jp, C, P, pb.camera_indices, pb.point_indices).toarray()
        assert _rel(J, g[pre + "J3"]) < 1e-8                             # vs 3-point FD of the reference
        assert _rel(J, g[pre + "J2"]) < 1e-5                             # vs the 2-point FD scipy uses


def test_jacobian_rotation_magnitudes(be, orc):
    g = np.load(os.path.join(GOLDEN, "residual_cases.npz"))
    tags = [str(g[f"c{k:02d}_tag"]) for k in range(int(g["n_cases"]))]
    pre = f"c{tags.index('theta_sweep'):02d}_"
    C, P, N = (int(v) for v in g[pre + "dims"])
    args = (C, P, g[pre + "ci"], g[pre + "pi"], g[pre + "uv"], g[pre + "K"])
    be.set_problem(*args)
    r, Jc, Jp = be.residual_jacobian(g[pre + "x"])
    _, Jc_o, Jp_o = orc.jacobian_blocks(g[pre + "x"], *args)
    assert _rel(Jc, Jc_o) < 1e-11 and _rel(Jp, Jp_o) < 1e-11


# ---- K2/K3: normal-equation blocks; K4/K5: implicit Schur product --------------------------------------

def _blocks_case(be, orc, pb):
    be.set_problem(*pb.args)
    U, V, gc, gp = be.normal_blocks(pb.x0)
    r, Jc, Jp = orc.jacobian_blocks(pb.x0, *pb.args)
    nb = orc.normal_blocks(r, Jc, Jp, pb.n_cameras, pb.n_points, pb.camera_indices, pb.point_indices)
    assert _rel(U, _upper(nb.U)) < 1e-11
    assert _rel(V, _upper(nb.V)) < 1e-11
    assert _rel(gc, nb.gc) < 1e-10
    assert _rel(gp, nb.gp) < 1e-10
    return nb


def _matvec_case(be, orc, pb, nb, seed=0):
    rng = np.random.default_rng(seed)
    C, P = pb.n_cameras, pb.n_points
    dc = 1e-3 * np.einsum("cii->ci", nb.U) + 1e-6
    dp = 1e-3 * np.einsum("pii->pi", nb.V) + 1e-6
    v = rng.normal(size=6 * C)
    y = be.schur_matvec(pb.x0, dc, dp, v)
    ci, pi = pb.camera_indices, pb.point_indices
    Vd = nb.V.copy()
    Vd[:, np.arange(3), np.arange(3)] += dp
    Vinv = np.linalg.inv(Vd)
    vc = v.reshape(C, 6)
    yy = np.zeros((P, 3))
    np.add.at(yy, pi, np.einsum("nij,ni->nj", nb.W, vc[ci]))
    z = np.einsum("pij,pj->pi", Vinv, yy)
    ref = np.einsum("cij,cj->ci", nb.U, vc) + dc * vc
    np.add.at(ref, ci, -np.einsum("nij,nj->ni", nb.W, z[pi]))
    assert _rel(y, ref.ravel()) < 1e-9
    return y


def test_normal_blocks_and_schur_matvec_small(be, orc):
    from sfmba import make_problem
    for (C, P, N, seed) in [(3, 8, 20, 0), (4, 30, 120, 11), (11, 300, 2000, 2)]:
        pb = make_problem(C, P, N, seed=seed)
        nb = _blocks_case(be, orc, pb)
        _matvec_case(be, orc, pb, nb)


def test_long_tracks_and_ragged_runs(be, orc):
    """Points with more than 64 observations (the wave loops over the run), single-observation points
    and a run that crosses every 64-lane step boundary."""
    from sfmba import make_problem
    pb = make_problem(7, 12, 1500, seed=6)
    pi = pb.point_indices.copy()
    # ragged: point 0 gets 1 observation, point 1 gets ~700, the rest share what is left
    pi[:] = np.sort(np.concatenate([[0], np.full(700, 1), np.full(65, 2), np.full(64, 3), np.full(63, 4),
                                    np.random.default_rng(1).integers(5, 12, 1500 - 893)]))
    pb2 = type(pb)(pb.n_cameras, pb.n_points, pb.camera_indices, pi, pb.points_2d, pb.K, pb.x0, pb.x_true)
    nb = _blocks_case(be, orc, pb2)
    _matvec_case(be, orc, pb2, nb)


def test_random_track_length_distributions(be, orc):
    """Random run structures: geometric and heavy-tailed track lengths, runs of exactly 64 / 65 / 128 /
    256 / 300 observations (step cuts, the > 64 long-run path, the 255 clip of the run-offset byte),
    points without observations, a camera seen once."""
    from sfmba import make_problem
    rng = np.random.default_rng(42)
    for trial in range(5):
        P = 150
        if trial == 0:
            lens = rng.geometric(0.15, P)
        elif trial == 1:
            lens = np.minimum(1 + (rng.pareto(1.2, P) * 3).astype(int), 400)
        elif trial == 2:
            lens = rng.integers(0, 4, P)                       # many empty points
            lens[:6] = [64, 65, 128, 256, 300, 1]
        elif trial == 3:
            lens = np.full(P, 1)
            lens[::7] = 63
        else:
            lens = rng.integers(1, 130, P)
        lens = np.asarray(lens, dtype=np.int64)
        lens[-1] = max(lens[-1], 1)
        N = int(lens.sum())
        C = 9
        base = make_problem(C, P, max(N, P), seed=100 + trial)
        pi = np.repeat(np.arange(P, dtype=np.int64), lens)
        ci = rng.integers(0, C - 1, N).astype(np.int64)
        ci[0] = C - 1                                          # camera C-1 is seen exactly once
        uv = base.points_2d[:N]
        pb = type(base)(C, P, ci, pi, uv, base.K, base.x0, base.x_true)
        nb = _blocks_case(be, orc, pb)
        _matvec_case(be, orc, pb, nb, seed=trial)
        r, Jc, Jp = be.residual_jacobian(pb.x0)
        r_o, Jc_o, Jp_o = orc.jacobian_blocks(pb.x0, *pb.args)
        assert np.abs(r - r_o.ravel()).max() <= 1e-11 * max(3000.0, np.abs(r_o).max())
        assert _rel(Jc, Jc_o) < 1e-11 and _rel(Jp, Jp_o) < 1e-11


def test_many_cameras_global_table_variants(be, orc):
    """More cameras than fit the LDS: 1300 (camera table of K1 read from L2), 1800 / 2600 / 21000 (pass A with its
    table in global memory)."""
    from sfmba import make_problem
    for C in (1300, 1800, 2600, 21000):
        pb = make_problem(C, 500, 6000, seed=C)
        be.set_problem(*pb.args)
        r, Jc, Jp = be.residual_jacobian(pb.x0)
        r_o, Jc_o, Jp_o = orc.jacobian_blocks(pb.x0, *pb.args)
        assert np.abs(r - r_o.ravel()).max() < 1e-8
        assert _rel(Jc, Jc_o) < 1e-11 and _rel(Jp, Jp_o) < 1e-11
        nb = _blocks_case(be, orc, pb)
        _matvec_case(be, orc, pb, nb)


@pytest.fixture
def dbg():
    """Sets sfmba_debug_option values on the thread's Backend (the one sfmba.least_squares uses) and on the
    module's `be`, and resets them afterwards."""
    import sfmba
    touched = []

    def set_(backends, name, value):
        for b in backends:
            b.debug_option(name, value)
            touched.append((b, name))
    yield set_
    defaults = dict(pcg_fused=-1, tab_lds=-1, vec_lds=-1, cam_chunk=0, pcg_guess_bias=0, sweep_rc=-1, dense=-1, precond=-1, pcg_local=-1,
                    pcg_split=-1, rhsrec=-1, cost_rider=-1, xcd_chunks=-1, pcg_mixed=-1, pcg_mixed_b=-1, jfree=-1, xcd_cam=-1)
    for b, name in touched:
        b.debug_option(name, defaults[name])


def test_operand_placements_and_camera_chunking(be, orc, dbg):
    """The two forms of pass A (recomputing from the LDS table R|T|a'|u_T; reading the stored Jacobian, with the
    camera vector in LDS or gathered from L2), the placement of the camera table, and the chunking of the
    camera-major passes (one chunk per camera, written directly; several chunks, combined by k_cam_combine) give
    the same blocks, the same product and the same solve."""
    import sfmba
    pb = sfmba.make_problem(40, 400, 5000, seed=3)
    tls = sfmba.get_backend(0)
    ref = None
    # (rhsrec: the rhs + preconditioner pass gathering one 128-byte record per observation instead of the point record
    # and the inverse block -- the form problems of >= 250k points run; tab_lds = 0: camera rows through LDS-DMA slabs)
    # xcd_chunks: every camera's list cut at the eight point-range boundaries (one chunk per XCD; >= 250k points) -- pass B,
    # K3 and the rhs pass then run one wave per chunk; xcd_cam = 0: pass B alone (K3 and the rhs pass one workgroup per camera)
    for tab_lds, vec_lds, chunk, rc, rr, xc, xk in ((-1, -1, 0, -1, -1, -1, -1), (-1, -1, 0, 0, -1, -1, -1), (0, -1, 0, 0, 1, -1, -1),
                                                    (-1, 0, 0, 0, -1, -1, -1), (-1, -1, 64, -1, 1, -1, -1), (0, 0, 50, 0, -1, -1, -1),
                                                    (-1, -1, 0, -1, 1, 1, -1), (0, -1, 0, -1, -1, 1, -1), (-1, -1, 0, -1, -1, 1, 0)):
        for name, v in (("tab_lds", tab_lds), ("vec_lds", vec_lds), ("cam_chunk", chunk), ("sweep_rc", rc), ("rhsrec", rr),
                        ("xcd_chunks", xc), ("xcd_cam", xk)):
            dbg((be, tls), name, v)
        nb = _blocks_case(be, orc, pb)
        y = _matvec_case(be, orc, pb, nb)
        res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args)
        if ref is None:
            ref = (y, res)
        assert _rel(y, ref[0]) < 1e-11
        assert (res.status, res.nfev) == (ref[1].status, ref[1].nfev)
        assert abs(res.cost - ref[1].cost) <= 1e-10 * ref[1].cost


def test_mixed_precision_schur_product(be, orc, dbg):
    """BASELINE configs[4]: "mixed-precision PCG with fp64 accumulation".  Debug option pcg_mixed = 1 (the default with
    fp32 storage) runs the implicit Schur product of the PCG on fp32 OPERANDS -- camera rows R | T - o | a' | u_T, point
    records X - o | z -- with every arithmetic operation and sum in fp64; gradient, blocks, right-hand side, preconditioner
    and the PCG's own vectors stay fp64.  Checked on the LDS-table form (<= 1100 cameras, fused and two-kernel PCG), on
    the table-in-global-memory form (LDS-DMA row slabs), with pass B on fp32 records and on its fp64 ones, with the
    XCD-aware chunk table (one wave per chunk, camera count not a multiple of four, cameras with empty ranges), and on a
    scene far from the coordinate origin (coordinates are rounded relative to an origin near the points):
    product within 1e-6 of the oracle's (measured 1e-7), symmetric to 1e-6; whole solves take the iterations of the
    exact product -- status, nfev, njev, PCG iterations per outer iteration -- and end at the same cost."""
    import sfmba
    tls = sfmba.get_backend(0)
    dbg((be, tls), "dense", 0)
    for C, P, N, shift in ((37, 400, 5000, 0.0), (300, 4000, 30000, 0.0), (1300, 4000, 40000, 0.0), (37, 400, 5000, 4.0e6)):
        pb = sfmba.make_problem(C, P, N, seed=3)
        if shift:                                                # the same scene 4000 km from the origin
            x0 = pb.x0.copy()
            x0[:6 * C].reshape(C, 6)[:, 3:] += shift
            x0[6 * C:] += shift
            pb = sfmba.BAProblem(C, P, pb.camera_indices, pb.point_indices, pb.points_2d, pb.K, x0, pb.x_true)
        dbg((be, tls), "pcg_mixed", 0)
        nb = _blocks_case(be, orc, pb)
        y_ref = _matvec_case(be, orc, pb, nb)
        exact = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf", args=pb.args)
        hist = tls.pcg_history()
        for mixed_b, xcd, fused in ((-1, -1, -1), (0, -1, -1), (-1, 1, -1), (0, 1, -1), (-1, -1, 0)):
            for name, v in (("pcg_mixed", 1), ("pcg_mixed_b", mixed_b), ("xcd_chunks", xcd), ("pcg_fused", fused)):
                dbg((be, tls), name, v)
            be.set_problem(*pb.args)
            dc = 1e-3 * np.einsum("cii->ci", nb.U).ravel() + 1e-6
            dp = 1e-3 * np.einsum("pii->pi", nb.V).ravel() + 1e-6
            v = np.random.default_rng(0).normal(size=6 * C)          # (the vector of _matvec_case)
            w = np.random.default_rng(5).normal(size=6 * C)
            Sv, Sw = be.schur_matvec(pb.x0, dc, dp, v), be.schur_matvec(pb.x0, dc, dp, w)
            tag = (C, shift, mixed_b, xcd, fused)
            assert 1e-13 < _rel(Sv, y_ref) < 1e-6, tag
            assert abs(w @ Sv - v @ Sw) <= 1e-6 * abs(w @ Sv), tag
            res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf", args=pb.args)
            assert (res.status, res.nfev, res.njev) == (exact.status, exact.nfev, exact.njev), tag
            assert tls.pcg_history() == hist, tag
            # (the shifted scene stops on xtol after its first steps -- |x| is huge -- i.e. NOT at a minimum, where the
            # 1e-7 perturbation of the PCG iterates shows in the ninth digit of the cost; converged runs agree to 1e-12)
            assert abs(res.cost - exact.cost) <= (1e-8 if shift else 1e-9) * exact.cost, tag
            assert np.abs(res.x - exact.x).max() <= 1e-6 * np.abs(exact.x).max(), tag
        for name in ("pcg_mixed", "pcg_mixed_b", "xcd_chunks", "pcg_fused"):
            dbg((be, tls), name, -1)


def test_j_free_iteration_equals_the_stored_jacobian_iteration(dbg):
    """Debug option jfree = 1 (a measurement mode): K1 forms the blocks for the point sums but does
    not write them, and the two consumers of the stored Jacobian -- k_jdot (J D^2 g) and k_backsub (back-substitution,
    J p, Gram sums) -- recompute every observation's blocks from the LDS camera table and its point, by the function K1
    itself uses.  With fp64 storage that is the same arithmetic on the same inputs: the whole solve is equal to the BIT;
    with fp32 storage the recomputed blocks are the exact ones instead of their fp32 roundings: same iterations, cost to
    1e-9."""
    import sfmba
    tls = sfmba.get_backend(0)
    dbg((tls,), "dense", 0)
    for pb in (sfmba.make_problem(300, 4000, 30000, seed=3), sfmba.make_config("cfg2"),
               sfmba.make_problem(6, 80, 500, seed=5, x0_noise=0.2), sfmba.make_problem(900, 3000, 40000, seed=8)):
        for bits in (64, 32):
            runs = []
            for jf in (-1, 1):
                dbg((tls,), "jfree", jf)
                runs.append(sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                                args=pb.args, storage_bits=bits))
                runs[-1].fun                                   # (download before the next solve reuses the buffers)
            a, b = runs
            assert (a.status, a.nfev, a.njev, a.pcg_iterations) == (b.status, b.nfev, b.njev, b.pcg_iterations)
            if bits == 64:
                assert a.cost == b.cost and np.array_equal(a.x, b.x) and np.array_equal(a.fun, b.fun)
            else:
                # (x: the fp32 roundings of the stored blocks move the iterates along the weakly determined directions)
                assert abs(a.cost - b.cost) <= 1e-9 * a.cost and np.abs(a.fun - b.fun).max() <= 1e-3
                assert np.abs(a.x - b.x).max() <= 1e-3 * np.abs(a.x).max()
    dbg((tls,), "jfree", -1)


def test_results_are_bitwise_reproducible(be):
    """No kernel uses atomics: per-point sums are reduced inside a wave, per-camera sums inside a workgroup
    over the camera-major order, chunks and workgroup partials in index order.  Two runs from the same input
    must agree in every bit -- blocks, product and the whole solve."""
    import sfmba
    for pb in (sfmba.make_config("cfg2"), sfmba.make_problem(300, 4000, 30000, seed=3)):
        be.set_problem(*pb.args)
        a, b = be.normal_blocks(pb.x0), be.normal_blocks(pb.x0)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))
        runs = [sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                    args=pb.args) for _ in range(3)]
        for r in runs[1:]:
            assert np.array_equal(r.x, runs[0].x) and r.cost == runs[0].cost
            assert (r.status, r.nfev, r.njev, r.pcg_iterations) == (runs[0].status, runs[0].nfev, runs[0].njev,
                                                                    runs[0].pcg_iterations)
            assert np.array_equal(r.fun, runs[0].fun) and np.array_equal(r.grad, runs[0].grad)


# ---- A5-A9: the solver -------------------------------------------------------------------------------

def _fun_bound(case, dense):
    """Twice the largest residual difference the ORACLE shows against the same scipy result (tools/gen_golden.py --params)."""
    rows = json.load(open(os.path.join(GOLDEN, "param_bounds.json")))["rows"]
    return 2.0 * next(r["fun_max"] for r in rows if r["case"] == case and r["settings"] == ("dense" if dense else "implicit"))


def _oracle_kwargs(dense):
    """The reduced camera system is solved by PCG to 1e-2 with the Schur-diagonal block preconditioner: with the
    implicit product (two launches per iteration), or, when 6 C <= 128 (`dense`), with S formed and the iterations
    inside one workgroup.  Same algorithm, so one restatement in the oracle."""
    if dense:         # in LDS: a decade tighter, and the preconditioner blocks are those of the formed matrix
        return dict(linear="pcg", pcg_tol=1e-3, precond="schur_exact")
    return dict(linear="pcg", pcg_tol=1e-2, pcg_tol_max=0.1, precond="schur")


def _ran_dense(be, calls):
    """Which path ran: the in-LDS PCG needs ~15 launches per outer iteration whatever its iteration count, the
    implicit one two more per PCG iteration."""
    l0 = be.counters()[0]
    out = calls()
    launches = be.counters()[0] - l0
    return out, launches < 17 * (out.nfev + 1) and out.pcg_iterations > 0


def test_solve_matches_scipy_on_tiny_problems(orc, dbg):
    """Stated tolerance (BASELINE.json north_star): final reprojection RMSE within 1e-6 px of the scipy
    reference on identical inputs.  scipy stops in a slow tail (optimality ~1e-1), so our cost may only
    be lower; the residual vectors agree to twice what the oracle measured against the same scipy run (2e-3 ... 9e-3 px;
    parameters: tests/test_gpu_params.py)."""
    import sfmba
    g = np.load(os.path.join(GOLDEN, "lsq_tiny_cases.npz"))
    for k, dense in [(k, d) for k in range(int(g["n_cases"])) for d in (True, False)]:
        dbg((sfmba.get_backend(0),), "dense", -1 if dense else 0)
        pre = f"l{k}_"
        C, P, N = (int(v) for v in g[pre + "dims"])
        pb = sfmba.make_problem(C, P, N, seed=int(g[pre + "seed"]))
        status, nfev, njev, cost, rmse, opt = g[pre + "summary"]
        res, ran_dense = _ran_dense(sfmba.get_backend(0), lambda: sfmba.least_squares(
            sfmba.compute_residuals, pb.x0, jac_sparsity=None, verbose=0, x_scale="jac", ftol=1e-10, method="trf",
            args=pb.args))
        assert ran_dense == dense
        assert res.success and res.status in (1, 2, 3, 4)
        my_rmse = float(np.sqrt(np.mean(res.fun ** 2)))
        assert abs(my_rmse - rmse) < 1e-6
        assert abs(res.rmse - my_rmse) < 1e-12
        assert res.cost <= cost * (1 + 1e-9)
        assert np.abs(res.fun - g[pre + "fun"]).max() <= _fun_bound(f"tiny{k}_{C}_{P}_{N}", dense)
        # and against the oracle's restatement of the same algorithm
        o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, **_oracle_kwargs(dense))
        assert abs(res.cost - o.cost) <= 1e-8 * o.cost
        assert (res.status, res.nfev, res.njev) == (o.status, o.nfev, o.njev)
        # result.fun / result.grad are consistent with result.x
        r = orc.compute_residuals(res.x, *pb.args)
        assert np.abs(r - res.fun).max() < 1e-8
        assert abs(res.optimality - np.abs(res.grad).max()) <= 1e-12 * max(1.0, res.optimality)


def test_solve_cfg2_matches_recorded_scipy_run(orc):
    import sfmba
    path = os.path.join(GOLDEN, "scipy_cfg2_run.json")
    if not os.path.exists(path):
        pytest.skip("scipy cfg2 capture not generated")
    rec = json.load(open(path))
    pb = sfmba.make_problem(11, 3000, 10000, seed=0)
    S = sfmba.create_sparsity_matrix(11, 3000, 10000, pb.camera_indices, pb.point_indices)
    res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, jac_sparsity=S, verbose=0, x_scale="jac",
                              ftol=1e-10, method="trf", args=pb.args)
    assert abs(res.rmse0 - rec["rmse0"]) < 1e-9
    assert abs(res.rmse - rec["rmse"]) < 1e-6
    assert res.cost <= rec["cost"] * (1 + 1e-9)
    assert res.nfev <= rec["nfev"]


@pytest.mark.parametrize("dense", [True, False])
def test_rejected_steps_and_nfev_limit_follow_the_oracle(orc, dbg, dense):
    """Far starts (x0 noise 0.2) make the trust region reject steps: the retry path (2-D model re-solved
    with a smaller radius, speculative normal blocks discarded) and the max_nfev exit must follow the
    oracle's restatement of trf_no_bounds.

    Both forms of the PCG (implicit product / S in LDS) are held to the same standard."""
    import sfmba
    dbg((sfmba.get_backend(0),), "dense", -1 if dense else 0)
    saw_rejection = False
    for seed in (1, 2, 5):
        pb = sfmba.make_problem(6, 80, 500, seed=seed, x0_noise=0.2)
        o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, **_oracle_kwargs(dense))
        res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args)
        assert res.status == o.status
        assert (res.nfev, res.njev) == (o.nfev, o.njev)
        assert abs(res.cost - o.cost) <= 1e-8 * o.cost
        saw_rejection |= res.nfev > res.njev
    assert saw_rejection
    pb = sfmba.make_problem(6, 80, 500, seed=5, x0_noise=0.2)
    for max_nfev in range(2, 14):
        o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, max_nfev=max_nfev, **_oracle_kwargs(dense))
        res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args, max_nfev=max_nfev)
        assert res.status == o.status == 0 and res.nfev == o.nfev == max_nfev
        # unconverged iterates of a far start amplify last-bit differences (the in-LDS iterations use v_rcp-based
        # reciprocals and other sum orders than the oracle): two more digits of slack there
        assert abs(res.cost - o.cost) <= (1e-6 if dense else 1e-8) * o.cost
        assert np.abs(res.x - o.x).max() <= (1e-4 if dense else 1e-6) * np.abs(o.x).max()
        r = orc.compute_residuals(res.x, *pb.args)
        assert np.abs(r - res.fun).max() < 1e-8            # result.fun belongs to result.x


def test_speculative_pcg_miss_changes_nothing(dbg):
    """The PCG iterations of an outer iteration are enqueued from a guess, with the first trial step
    decided on the device behind them.  When the guess is too small the device cancels the trial and
    the host finishes the PCG by polling: the outcome must be the one of a run whose guesses sufficed.
    (debug option pcg_guess_bias shifts the guess; -5 makes every speculative batch fall short.)"""
    import sfmba
    tls = sfmba.get_backend(0)
    dbg((tls,), "dense", 0)                    # these sizes would take the dense path, which has no PCG to speculate on
    for pb in (sfmba.make_config("cfg2"), sfmba.make_problem(6, 80, 500, seed=5, x0_noise=0.2)):
        runs = []
        for bias in (0, -5, 4):
            dbg((tls,), "pcg_guess_bias", bias)
            runs.append(sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10,
                                            method="trf", args=pb.args))
        a = runs[0]
        for b in runs[1:]:
            assert (a.status, a.nfev, a.njev, a.pcg_iterations) == (b.status, b.nfev, b.njev, b.pcg_iterations)
            assert a.cost == b.cost and np.array_equal(a.x, b.x)           # same arithmetic, whatever the launch pattern
            r = sfmba.compute_residuals(b.x, *pb.args)
            assert np.abs(r - b.fun).max() < 1e-8


def test_replayed_pcg_record_that_turns_out_wrong(dbg):
    """Back-to-back solves of the same problem from the same start replay the previous solve's PCG record count for
    count: no spare launch, and the pass B behind the launch the record calls the last one is not even enqueued.  The
    record is keyed on the problem and the start, not on the options: the same call with a tighter forcing term needs
    MORE iterations than recorded.  The device then cancels the trial, the host enqueues the owed pass B and polls to the
    end -- same result as on a handle without a record; a record that is too long costs empty launches only."""
    import sfmba
    pb = sfmba.make_problem(60, 900, 9000, seed=12)
    hist = {}
    for tol in (1e-2, 1e-5):
        fresh = sfmba.Backend(0)
        fresh.debug_option("dense", 0)
        fresh.set_problem(*pb.args)
        opt = fresh.default_options()
        opt.ftol = 1e-10
        opt.pcg_tol = tol
        opt.pcg_tol_max = tol
        x, res, _, _ = fresh.solve(pb.x0, opt)
        hist[tol] = (x, res.nfev, res.cost, fresh.pcg_history())
        fresh.close()
    assert sum(hist[1e-5][3]) > sum(hist[1e-2][3]) + 5                   # the two records really differ
    be = sfmba.Backend(0)
    be.debug_option("dense", 0)
    be.set_problem(*pb.args)
    opt = be.default_options()
    opt.ftol = 1e-10
    for skip in (-1, 0):                                                 # (0: the last pass B is enqueued as before)
        be.debug_option("pcg_skip_last", skip)
        for tol in (1e-2, 1e-2, 1e-5, 1e-5, 1e-2):                       # exact replay, too short a record, too long a one
            opt.pcg_tol = tol
            opt.pcg_tol_max = tol
            x, res, _, _ = be.solve(pb.x0, opt)
            assert np.array_equal(x, hist[tol][0]) and (res.nfev, res.cost) == hist[tol][1:3]
            assert be.pcg_history() == hist[tol][3]
    be.close()


def test_fused_pcg_launch_equals_sweep_plus_update(dbg):
    """With v in LDS and at most 1024 cameras, the PCG update of the previous product runs in the prologue of
    pass A (one launch less per iteration).  It must walk through the same iterates as the separate
    k_pcg_update (debug option pcg_fused = 0, read at set_problem)."""
    import sfmba
    tls = sfmba.get_backend(0)
    dbg((tls,), "dense", 0)
    for pb in (sfmba.make_config("cfg2"), sfmba.make_problem(300, 4000, 30000, seed=3),
               sfmba.make_problem(6, 80, 500, seed=5, x0_noise=0.2), sfmba.make_problem(1024, 3000, 20000, seed=8)):
        for rc in (-1, 0):
            dbg((tls,), "sweep_rc", rc)
            runs = []
            # fused with the per-camera bookkeeping in pass B (default on one rank), fused with the whole update in
            # pass A's prologue, two kernels, and the stepped form of the sharded solves (k_pcg_begin / k_pcg_split: the
            # whole update once per iteration in one workgroup; here without ranks to reduce over) with the fused and
            # the plain pass A
            for fused, local, step in ((-1, -1, -1), (-1, 0, -1), (0, -1, -1), (-1, -1, 1), (0, -1, 1)):
                dbg((tls,), "pcg_fused", fused)
                dbg((tls,), "pcg_local", local)
                dbg((tls,), "pcg_split", step)
                runs.append(sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10,
                                                method="trf", args=pb.args))
            dbg((tls,), "pcg_split", -1)
            a = runs[0]
            for b in runs[1:]:
                # the last trial step of a converged run changes the cost in its 13th digit: whether it counts as
                # accepted (njev, ftol/xtol status 2/3/4) is rounding noise between the forms; the path is not
                assert (a.nfev, a.pcg_iterations) == (b.nfev, b.pcg_iterations)
                assert abs(a.njev - b.njev) <= 1 and {a.status, b.status} <= {2, 3, 4}
                assert abs(a.cost - b.cost) <= 1e-12 * a.cost
                # the forms sum their dot products in different (fixed) orders, and the local form takes gamma from a
                # one-step recurrence: steps differ in the last bits, x along the weakly determined gauge directions
                assert np.abs(a.x - b.x).max() <= 1e-6 * np.abs(a.x).max()
                assert np.abs(a.fun - b.fun).max() <= 1e-6


def test_full_solves_with_many_cameras_vs_oracle(orc, dbg):
    """Camera counts past the fused PCG launch and the LDS-resident tables, end to end: 1300 and 1800 cameras (camera
    table of K1 read from L2; pass A recomputing from a table in global memory, k_rc_table; the per-camera PCG
    bookkeeping in pass B with k_pcg_update_local).  The older forms stay reachable through debug options and must
    walk through the same iterations: the two-kernel update (pcg_local = 0) and pass A reading the stored Jacobian
    (sweep_rc = 0)."""
    import sfmba
    tls = sfmba.get_backend(0)
    for C, P, N in ((1300, 4000, 40000), (1800, 3000, 30000)):
        pb = sfmba.make_problem(C, P, N, seed=21)
        o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, **_oracle_kwargs(False))
        res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args)
        assert res.status == o.status and (res.nfev, res.njev) == (o.nfev, o.njev)
        assert abs(res.cost - o.cost) <= 1e-9 * o.cost
        r = sfmba.compute_residuals(res.x, *pb.args)
        assert np.abs(r - res.fun).max() < 1e-8
        if C == 1300:
            for name in ("pcg_local", "sweep_rc", "pcg_split"):
                dbg((tls,), name, 1 if name == "pcg_split" else 0)
                alt = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                          args=pb.args)
                dbg((tls,), name, -1)
                assert (alt.nfev, alt.pcg_iterations) == (res.nfev, res.pcg_iterations), name
                assert abs(alt.cost - res.cost) <= 1e-12 * res.cost, name
                assert np.abs(alt.x - res.x).max() <= 1e-6 * np.abs(res.x).max(), name


def test_cfg3_full_loop_vs_oracle(orc):
    """BASELINE config 3 (200 cameras / 20k points / 200k observations): the full Schur-LM loop on the
    GPU against the oracle's."""
    import sfmba
    pb = sfmba.make_config("cfg3")
    o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, **_oracle_kwargs(False))
    res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args)
    assert res.status == o.status and (res.nfev, res.njev) == (o.nfev, o.njev)
    assert abs(res.cost - o.cost) <= 1e-9 * o.cost
    assert abs(res.rmse - float(np.sqrt(np.mean(o.fun ** 2)))) < 1e-9


def test_unobserved_camera_and_point(orc):
    import sfmba
    g = np.load(os.path.join(GOLDEN, "lsq_tiny_cases.npz"))
    pb = sfmba.drop_observations(sfmba.make_problem(5, 40, 200, seed=9), cameras=(3,), points=(7,))
    status, nfev, njev, cost, rmse, opt = g["gaps_summary"]
    res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args, return_jac=True)
    assert res.success and abs(res.rmse - rmse) < 1e-6 and res.cost <= cost * (1 + 1e-9)
    for sl in (slice(18, 24), slice(30 + 21, 30 + 24)):          # unobserved parameters do not move
        assert np.array_equal(res.x[sl], pb.x0[sl])
    # result.jac: scipy's CSR layout, analytic values; J^T f reproduces result.grad
    r, Jc, Jp = orc.jacobian_blocks(res.x, *pb.args)
    J = orc.jacobian_csr(Jc, Jp, 5, 40, pb.camera_indices, pb.point_indices)
    assert res.jac.shape == J.shape and abs(res.jac - J).max() <= 1e-11 * abs(J).max()
    assert np.abs(res.jac.T @ res.fun - res.grad).max() <= 1e-9 * max(1.0, np.abs(res.grad).max())


@pytest.mark.parametrize("dense", [True, False])
def test_ring_scene_large_rotations(orc, dbg, dense):
    """Rotation vectors of every magnitude up to pi inside the solver (not only in K1): against scipy's
    recorded result and the oracle."""
    import sfmba
    from sfmba.synthetic import make_ring_problem
    g = np.load(os.path.join(GOLDEN, "lsq_tiny_cases.npz"))
    pb = make_ring_problem(12, 150, 900, seed=1)
    status, nfev, njev, cost, rmse, opt = g["ring_summary"]
    dbg((sfmba.get_backend(0),), "dense", -1 if dense else 0)
    res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args)
    o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, **_oracle_kwargs(dense))
    assert res.success and abs(res.rmse - rmse) < 1e-6 and res.cost <= cost * (1 + 1e-9)
    assert (res.status, res.nfev, res.njev) == (o.status, o.nfev, o.njev)
    assert abs(res.cost - o.cost) <= 1e-9 * o.cost
    assert np.abs(res.fun - g["ring_fun"]).max() <= _fun_bound("ring_12_150_900", dense)


def test_error_behaviour(be):
    import sfmba
    pb = sfmba.make_problem(3, 8, 20, seed=0)
    bad = pb.camera_indices.copy()
    bad[3] = 3
    with pytest.raises(ValueError):
        be.set_problem(3, 8, bad, pb.point_indices, pb.points_2d, pb.K)
    x = pb.x0.copy()
    x[0] = np.nan
    with pytest.raises(ValueError, match="not finite"):
        sfmba.least_squares(sfmba.compute_residuals, x, x_scale="jac", method="trf", args=pb.args)
    with pytest.raises(ValueError, match="wrong shape"):
        sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", method="trf", args=pb.args,
                            jac_sparsity=np.zeros((3, 3)))


# ---- size-independent properties at full size (cfg4: 1000 cams / 100k points / 1M observations) ---------

def test_properties_at_full_size(be):
    import sfmba
    pb = sfmba.make_config("cfg4")
    be.set_problem(*pb.args)
    rng = np.random.default_rng(0)
    # (1) noise-free projection of the truth reproduces the integer pixels to within truncation + noise
    r_true = be.residuals(pb.x_true).reshape(-1, 2)
    assert np.abs(r_true).max() < 1.0 + 6 * 0.5
    # (2) the gradient J^T r agrees with a directional finite difference of the cost
    U, V, gc, gp = be.normal_blocks(pb.x0)
    gvec = np.concatenate([gc.ravel(), gp.ravel()])
    d = rng.normal(size=gvec.shape)
    d /= np.linalg.norm(d)
    eps = 1e-4                                # cost ~1e9: fd round-off ~1e-16*1e9/eps
    cp = 0.5 * np.sum(be.residuals(pb.x0 + eps * d) ** 2)
    cm = 0.5 * np.sum(be.residuals(pb.x0 - eps * d) ** 2)
    fd = (cp - cm) / (2 * eps)
    assert abs(fd - gvec @ d) <= 1e-5 * abs(fd)
    # (3) the implicit Schur complement is linear and symmetric
    dc = 1e-3 * U[:, [0, 6, 11, 15, 18, 20]] + 1e-6
    dp = 1e-3 * V[:, [0, 3, 5]] + 1e-6
    v, w = rng.normal(size=6000), rng.normal(size=6000)
    Sv, Sw = be.schur_matvec(pb.x0, dc, dp, v), be.schur_matvec(pb.x0, dc, dp, w)
    Svw = be.schur_matvec(pb.x0, dc, dp, 2.0 * v - 3.0 * w)
    assert _rel(Svw, 2.0 * Sv - 3.0 * Sw) < 1e-10
    assert abs(v @ Sw - w @ Sv) <= 1e-10 * abs(v @ Sw)
    assert v @ Sv > 0
    # (4) a full solve converges, monotonically, to the noise floor of the generator
    res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args)
    assert res.success and res.cost < res.cost0
    assert 0.3 < res.rmse < 0.7            # sqrt(0.5^2 + truncation variance) ~ 0.58 px minus fitted dof
    r_fin = be.residuals(res.x)
    assert abs(0.5 * np.sum(r_fin ** 2) - res.cost) <= 1e-9 * res.cost


def test_reproj_error_rt_convention_and_problem_files(tmp_path):
    """cv2_lite.reproj_error / calc_reproj_error mirror (the pipeline's [R|t] convention) and the .npz
    problem file round trip."""
    import sfmba
    rng = np.random.default_rng(3)
    g = np.load(os.path.join(GOLDEN, "pack_cases.npz"))
    R, t = g["R"][5], np.array([0.3, -0.2, 9.0])
    X = rng.normal(size=(50, 3))
    uv = rng.normal(size=(50, 2)) * 100
    K = sfmba.K_SCEAUX
    ref = (K @ (R @ X.T + t.reshape(3, 1))).T
    ref = ref[:, :2] / ref[:, 2:3] - uv                       # solve_pnp.py:10-14
    err = sfmba.reproj_error(X, uv, K, R, t)
    assert np.abs(err - ref).max() < 1e-9
    assert abs(sfmba.calc_reproj_error(X, uv, K, R, t) - np.linalg.norm(ref, axis=1).mean()) < 1e-9
    pb = sfmba.make_problem(4, 30, 120, seed=11)
    f = tmp_path / "pb.npz"
    sfmba.save_problem(f, pb.x0, *pb.args)
    x0, args = sfmba.load_problem(f)
    a = sfmba.least_squares(sfmba.compute_residuals, x0, x_scale="jac", ftol=1e-10, method="trf", args=args)
    b = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf", args=pb.args)
    assert abs(a.cost - b.cost) <= 1e-10 * b.cost


# ---- fp32-storage mode (BASELINE config 5): floats in HBM, fp64 arithmetic and accumulation --------------

def test_fp32_storage_mode(orc):
    """Stated tolerances: stored residuals / Jacobian entries carry one fp32 rounding (<= 2^-24 relative
    to the largest magnitude, checked at 1e-6); the solve reaches scipy's RMSE on the SceauxCastle-scale
    problem to 1e-6 px and the fp64 path's cost to 1e-9 relative."""
    import sfmba
    be = sfmba.Backend(0)
    try:
        be.set_precision(32)
        pb = sfmba.make_problem(11, 300, 2000, seed=2)
        be.set_problem(*pb.args)
        r, Jc, Jp = be.residual_jacobian(pb.x0)
        r_o, Jc_o, Jp_o = orc.jacobian_blocks(pb.x0, *pb.args)
        assert np.abs(r - r_o.ravel()).max() <= 1e-6 * np.abs(r_o).max()
        assert _rel(Jc, Jc_o) < 1e-6 and _rel(Jp, Jp_o) < 1e-6
        nb = orc.normal_blocks(r_o, Jc_o, Jp_o, pb.n_cameras, pb.n_points, pb.camera_indices, pb.point_indices)
        U, V, gc, gp = be.normal_blocks(pb.x0)
        assert _rel(U, _upper(nb.U)) < 1e-6 and _rel(V, _upper(nb.V)) < 1e-6
        assert _rel(gc, nb.gc) < 1e-5 and _rel(gp, nb.gp) < 1e-5
    finally:
        be.close()
    rec = json.load(open(os.path.join(GOLDEN, "scipy_cfg2_run.json")))
    pb = sfmba.make_problem(11, 3000, 10000, seed=0)
    a = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                            args=pb.args, storage_bits=32)
    b = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                            args=pb.args, storage_bits=64)
    assert a.success and abs(a.rmse - rec["rmse"]) < 1e-6
    assert abs(a.cost - b.cost) <= 1e-9 * b.cost
    assert np.abs(a.fun - b.fun).max() < 1e-4


# ---- the N>1 code path on one GPU: world_size-1 RCCL through the same Exchange / callback plumbing -------

def test_exchange_path_world1_nccl(dbg):
    """Runs a solve with the all-reduce callback active (torch.distributed backend nccl = RCCL, world
    size 1, exchange arena = a torch CUDA tensor, kernels and collectives on one torch stream) and
    checks that it reproduces the single-process result exactly."""
    import socket
    import torch
    import torch.distributed as td
    import sfmba
    from sfmba import dist as sdist
    pb = sfmba.make_problem(11, 3000, 10000, seed=0)
    dbg((sfmba.get_backend(0),), "dense", 0)       # with a transport registered the solver takes the PCG path
    dbg((sfmba.get_backend(0),), "pcg_split", 1)   # ... in the form sharded solves run (tail behind the reduction): bitwise comparable
    ref = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    td.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                          device_id=torch.device("cuda", 0))
    try:
        be = sfmba.Backend(0)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            be.set_stream(stream.cuda_stream)
            be.set_problem(*pb.args)
            ex = sdist.Exchange(be, n_obs_local=pb.n_obs)
            assert ex.n_obs_total == pb.n_obs
            opt = be.default_options()
            opt.ftol = 1e-10
            x, res, fun, grad = be.solve(pb.x0, opt)
            torch.cuda.synchronize()
        assert ex.n_calls > 20                      # the callback really was on the path
        assert res.status == ref.status and int(res.nfev) == ref.nfev
        assert res.cost == ref.cost
        assert np.array_equal(x, ref.x)              # a world-1 all-reduce is the identity: bitwise the same solve
        be.close()
        # the native path: RCCL called from C++ on the solver's stream
        be2 = sfmba.Backend(0)
        be2.set_problem(*pb.args)
        nc = sdist.NativeComm(be2, n_obs_local=pb.n_obs)
        assert nc.world == 1 and nc.n_obs_total == pb.n_obs
        x2, res2, _, _ = be2.solve(pb.x0, opt)
        assert res2.status == ref.status and int(res2.nfev) == ref.nfev
        assert res2.cost == ref.cost and np.array_equal(x2, ref.x)
        # a new problem tears the transport down (include/sfmba.h) -- and says so: the next compute call fails until a
        # transport is attached again or single-rank use is acknowledged; it does not quietly solve the shard alone
        be2.set_problem(*pb.args)
        with pytest.raises(ValueError, match="transport"):
            be2.solve(pb.x0, opt)
        with pytest.raises(ValueError, match="transport"):
            be2.residuals(pb.x0)
        nc = sdist.NativeComm(be2, n_obs_local=pb.n_obs)            # set up again: works
        x3 = be2.solve(pb.x0, opt)[0]
        assert np.array_equal(x3, ref.x)
        be2.set_problem(*pb.args)
        be2.set_exchange(0, 0, None, 0)                              # acknowledged: single-rank
        be2.debug_option("dense", 0)                                 # (the forms `ref` was computed with)
        be2.debug_option("pcg_split", 1)
        be2.set_problem(*pb.args)
        assert np.array_equal(be2.solve(pb.x0, opt)[0], ref.x)
        be2.comm_destroy()
        be2.close()
    finally:
        td.destroy_process_group()


# ---- a REAL two-rank run of the C++ solver: both ranks share the one GPU, collectives over gloo ------------

def _two_rank_worker(rank, world, port, q, direct=False, dims=(11, 3000, 10000, 0, 0.01), debug=()):
    import torch
    import torch.distributed as td
    import sfmba
    from sfmba import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _two_rank_body(rank, world, q, direct, dims, debug, torch, td, sfmba, sdist)
    except Exception as e:                                       # noqa: BLE001 -- the parent fails at once, with the text
        q.put(dict(error=f"rank {rank}: {type(e).__name__}: {e}"))
        raise
    finally:
        td.destroy_process_group()


def _two_rank_body(rank, world, q, direct, dims, debug, torch, td, sfmba, sdist):
    pb = sfmba.make_problem(dims[0], dims[1], dims[2], seed=dims[3], x0_noise=dims[4])
    shards = sdist.partition_points(pb.point_indices, pb.n_points, world)
    loc = sdist.shard_problem(pb, shards[rank])
    be = sfmba.Backend(0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        be.set_stream(stream.cuda_stream)
        for name, value, *only in debug:                                  # (name, value[, the one rank it is set on])
            if not only or only[0] == rank:
                be.debug_option(name, value)
        be.set_problem(*loc.args)
        ex = sdist.Exchange(be, n_obs_local=loc.n_obs, device="cuda")     # gloo all-reduces CUDA tensors
        link = sdist.DirectLink(be) if direct else None                   # peers mapped through hipIpc
        opt = be.default_options()
        opt.ftol = 1e-10
        n0 = be.counters()[0]
        x, res, fun, grad = be.solve(loc.x0, opt)
        launches = be.counters()[0] - n0
        torch.cuda.synchronize()
        x2 = be.solve(loc.x0, opt)[0] if direct else x                    # staging buffers are reusable
    td.barrier()
    direct_calls = be.p2p_calls()
    if link is not None:
        link_active = link.active
        link.close()
    else:
        link_active = False
    xs = [None] * world
    td.all_gather_object(xs, x)
    if rank == 0:
        q.put(dict(x=sdist.merge_solutions(xs, shards, pb.n_cameras, pb.n_points),
                   cams_equal=all(np.array_equal(xi[:6 * dims[0]], xs[0][:6 * dims[0]]) for xi in xs),
                   status=int(res.status), nfev=int(res.nfev), cost=float(res.cost), rmse=float(res.rmse),
                   calls=ex.n_calls, direct_calls=direct_calls, link_active=link_active, launches=int(launches),
                   again=float(np.abs(x2 - x).max())))
    be.close()


def _run_ranks(world, direct, dims=(11, 3000, 10000, 0, 0.01), debug=()):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q, direct, dims, debug)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=20 if "error" in out else 300)
        if p.is_alive():
            p.terminate()
    assert "error" not in out, out["error"]
    assert all(p.exitcode == 0 for p in procs)
    return out


def test_direct_allreduce_over_peer_mapped_memory(dbg):
    """The latency path of the collectives (k_p2p_allreduce): ranks map each other's staging buffers through
    hipIpc and all-reduce with one kernel per collective.  Here the ranks are 2 and 3 processes on the one
    GPU (on a node they are one per GPU over xGMI; the code path is the same).  Every collective of the
    solves must have gone through the direct path (the gloo callback stays registered but idle), cameras
    bitwise replicated, result equal to the single-process solve."""
    import sfmba
    pb = sfmba.make_problem(11, 3000, 10000, seed=0)
    dbg((sfmba.get_backend(0),), "dense", 0)       # sharded solves take the PCG path: so does the reference run
    ref = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args)
    for world in (2, 3):
        out = _run_ranks(world, direct=True)
        assert out["link_active"], "peers could not be mapped or the self-test failed"
        assert out["calls"] == 0 and out["direct_calls"] > 40
        assert out["cams_equal"] and out["again"] == 0.0                  # same input, same bits
        # whether the last, 13th-digit step counts as an ftol, xtol or joint stop (2 / 3 / 4) is rounding noise
        # between differently ordered sums; the path (nfev, cost, x) is not
        assert out["nfev"] == ref.nfev and (out["status"] == ref.status or {out["status"], ref.status} <= {2, 3, 4})
        assert abs(out["cost"] - ref.cost) <= 1e-10 * ref.cost
        assert np.abs(out["x"] - ref.x).max() <= 1e-6 * np.abs(ref.x).max()
    # other shapes: an odd camera count (scalar slots no longer 16-byte aligned), a far start with rejected steps
    # (retries, speculative trials), a hundred cameras, and sizes at which ranks that SHARE a device fall back to the
    # collective launches by themselves (300 cameras; 1300: also more than the LDS table of pass A holds) -- waiting camera
    # workgroups of one rank on every CU would keep the other rank's pass A from ever starting
    for world, dims in ((3, (7, 500, 4000, 3, 0.01)), (2, (6, 80, 500, 5, 0.2)), (2, (100, 1500, 12000, 4, 0.01)),
                        (3, (60, 900, 8000, 6, 0.01)), (2, (300, 2000, 16000, 9, 0.01)), (2, (1300, 4000, 40000, 21, 0.01))):
        pb = sfmba.make_problem(dims[0], dims[1], dims[2], seed=dims[3], x0_noise=dims[4])
        ref = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args)
        out = _run_ranks(world, direct=True, dims=dims)
        assert out["link_active"] and out["calls"] == 0 and out["cams_equal"]
        # whether the last, 13th-digit step counts as an ftol, xtol or joint stop (2 / 3 / 4) is rounding noise
        # between differently ordered sums; the path (nfev, cost, x) is not
        assert out["nfev"] == ref.nfev and (out["status"] == ref.status or {out["status"], ref.status} <= {2, 3, 4})
        assert abs(out["cost"] - ref.cost) <= 1e-9 * ref.cost
        if dims[0] in (100, 300):
            # the exchange really ran inside the producing kernels at 100 cameras (fewer launches than with the
            # collectives as launches of their own, same result) and did not at 300
            off = _run_ranks(world, direct=True, dims=dims, debug=(("pcg_inline", 0),))
            assert off["nfev"] == out["nfev"] and abs(off["cost"] - out["cost"]) <= 1e-9 * out["cost"]
            assert (out["launches"] < off["launches"]) == (dims[0] == 100), (out["launches"], off["launches"])
    # the form problems of more than 1100 cameras run -- pass A's table in global memory, the PCG update a kernel of its
    # own -- WITH the per-camera exchange inside pass B / K3 / the rhs pass; forced at 100 cameras through the test hooks (the
    # camera count at which two ranks on one device still take the in-kernel exchange), with fp64 and with fp32 operands
    pb = sfmba.make_problem(100, 1500, 12000, seed=4)
    ref = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf", args=pb.args)
    for extra in ((), (("pcg_mixed", 1),)):
        out = _run_ranks(2, direct=True, dims=(100, 1500, 12000, 4, 0.01), debug=(("sweep_rc", 2), ("pcg_fused", 0)) + extra)
        assert out["link_active"] and out["calls"] == 0 and out["cams_equal"] and out["again"] == 0.0
        assert out["nfev"] == ref.nfev and (out["status"] == ref.status or {out["status"], ref.status} <= {2, 3, 4})
        assert abs(out["cost"] - ref.cost) <= 1e-9 * ref.cost
    # Whether a camera's list is cut into several chunks is a property of a rank's SHARD, and the in-kernel exchange needs
    # single-chunk cameras: the ranks settle it at attach.  Rank 1 alone is given short chunks here -- every rank then
    # keeps the collective launches (ranks that decided each for itself would wait for each other in different kernels).
    out = _run_ranks(2, direct=True, dims=(100, 1500, 12000, 4, 0.01), debug=(("cam_chunk", 16, 1),))
    assert out["link_active"] and out["calls"] == 0 and out["cams_equal"] and out["again"] == 0.0
    assert out["nfev"] == ref.nfev and abs(out["cost"] - ref.cost) <= 1e-9 * ref.cost


def test_two_rank_solve_on_one_gpu_gloo(dbg):
    """The production solver, observation-sharded over TWO processes that share the single GPU, with the
    callback transport over gloo: replicated cameras bitwise identical on both ranks, merged solution and
    cost equal to the single-process solve (and scipy's RMSE to 1e-6)."""
    import sfmba
    rec = json.load(open(os.path.join(GOLDEN, "scipy_cfg2_run.json")))
    pb = sfmba.make_problem(11, 3000, 10000, seed=0)
    dbg((sfmba.get_backend(0),), "dense", 0)       # with a transport registered the solver takes the PCG path
    dbg((sfmba.get_backend(0),), "pcg_split", 1)   # ... in the form sharded solves run
    ref = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                              args=pb.args)
    out = _run_ranks(2, direct=False)
    assert out["cams_equal"] and out["calls"] > 20
    # (2 / 3 / 4: which of ftol and xtol the last, 13th-digit step satisfies is rounding noise between sum orders)
    assert out["nfev"] == ref.nfev and (out["status"] == ref.status or {out["status"], ref.status} <= {2, 3, 4})
    assert abs(out["cost"] - ref.cost) <= 1e-10 * ref.cost
    assert abs(out["rmse"] - rec["rmse"]) < 1e-6
    assert np.abs(out["x"] - ref.x).max() <= 1e-6 * np.abs(ref.x).max()


# ---- few cameras: S formed, the PCG inside one workgroup with S in LDS ---------------------------------------------

def test_dense_reduced_camera_matrix_and_in_lds_pcg(be, orc):
    """6 C <= 128: the reduced camera matrix is formed block pair by block pair and the PCG runs inside one workgroup
    with S in LDS.  S against the oracle's explicit Schur complement, the solve (run to the end through the test
    entry) against numpy's, for 2 ... 21 cameras, duplicated (camera, point) pairs, an unobserved camera."""
    import sfmba
    rng = np.random.default_rng(7)
    for C, P, N, seed in ((2, 12, 40, 1), (3, 8, 20, 0), (5, 40, 200, 9), (11, 300, 2000, 2), (16, 200, 1500, 4),
                          (21, 150, 1200, 5)):
        pb = sfmba.make_problem(C, P, N, seed=seed)
        if C == 5:
            pb = sfmba.drop_observations(pb, cameras=(3,), points=(7,))
        ci = pb.camera_indices.copy()
        if C == 11:
            ci[5] = ci[4] if pb.point_indices[5] == pb.point_indices[4] else ci[5]      # duplicated pair
        args = (pb.n_cameras, pb.n_points, ci, pb.point_indices, pb.points_2d, pb.K)
        be.set_problem(*args)
        r, Jc, Jp = orc.jacobian_blocks(pb.x0, *args)
        nb = orc.normal_blocks(r, Jc, Jp, C, pb.n_points, ci, pb.point_indices)
        dc = 1e-3 * np.einsum("cii->ci", nb.U) + 1e-6
        dp = 1e-3 * np.einsum("pii->pi", nb.V) + 1e-6
        rhs = rng.normal(size=6 * C)
        S, y = be.dense_schur(pb.x0, dc, dp, rhs)
        Vd = nb.V.copy()
        Vd[:, np.arange(3), np.arange(3)] += dp
        Vinv = np.linalg.inv(Vd)
        Sref = np.zeros((C, 6, C, 6))
        Sref[np.arange(C), :, np.arange(C), :] = nb.U
        Sref = Sref.reshape(6 * C, 6 * C) + np.diag(dc.ravel())
        Wfull = np.zeros((6 * C, 3 * pb.n_points))
        for i in range(len(ci)):
            Wfull[6 * ci[i]:6 * ci[i] + 6, 3 * pb.point_indices[i]:3 * pb.point_indices[i] + 3] += nb.W[i]
        Vbig = np.zeros((3 * pb.n_points, 3 * pb.n_points))
        for p in range(pb.n_points):
            Vbig[3 * p:3 * p + 3, 3 * p:3 * p + 3] = Vinv[p]
        Sref = Sref - Wfull @ Vbig @ Wfull.T
        assert np.abs(S - S.T).max() <= 1e-13 * np.abs(S).max()     # diagonal blocks: (u,v) and (v,u) are summed separately
        assert np.abs(S - Sref).max() <= 1e-10 * np.abs(Sref).max()
        yref = np.linalg.solve(Sref, rhs)
        assert np.abs(y - yref).max() <= 1e-8 * np.abs(yref).max(), (C, np.abs(y - yref).max(), np.abs(yref).max())
        assert np.abs(S @ y - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max()) * np.linalg.cond(Sref) ** 0.5


def _late_rank_worker(rank, world, port, q):
    import time
    import torch
    import torch.distributed as td
    import sfmba
    from sfmba import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pb = sfmba.make_problem(30, 600, 5000, seed=2)
        shards = sdist.partition_points(pb.point_indices, pb.n_points, world)
        loc = sdist.shard_problem(pb, shards[rank])
        be = sfmba.Backend(0)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            be.set_stream(stream.cuda_stream)
            be.set_problem(*loc.args)
            ex = sdist.Exchange(be, n_obs_local=loc.n_obs, device="cuda")
            link = sdist.DirectLink(be)
            assert link.active
            opt = be.default_options()
            opt.ftol = 1e-10
            x_ok, res_ok, _, _ = be.solve(loc.x0, opt)             # a healthy solve first
            td.barrier()
            be.debug_option("p2p_timeout_ms", 300)                 # both time-outs of the direct all-reduce
            if rank == 1:
                be.debug_option("p2p_delay_ms", 1500)              # this rank enters its solve 1.5 s late
            t0 = time.perf_counter()
            err = None
            try:
                be.solve(loc.x0, opt)
            except sfmba.BackendError as exc:
                err = str(exc)
            dt = time.perf_counter() - t0
            be.debug_option("p2p_timeout_ms", 0)
            be.debug_option("p2p_delay_ms", 0)
            td.barrier()
            # the direct link is gone after the failure; the callback transport registered before it still serves
            x_again, res_again, _, _ = be.solve(loc.x0, opt)
            torch.cuda.synchronize()
        calls_after = ex.n_calls
        q.put(dict(rank=rank, err=err, seconds=dt, cost_ok=float(res_ok.cost), cost_again=float(res_again.cost),
                   # (over the direct link the product is reduced camera by camera inside pass B and the PCG bookkeeping
                   # sums its dot products in that kernel's order; behind the callback it runs in k_pcg_tail: same
                   # arithmetic, other summation order -- equal to rounding, not to the bit)
                   same=bool(np.abs(x_ok[:180] - x_again[:180]).max() <= 1e-9 * np.abs(x_ok[:180]).max()), callback_calls=calls_after))
        td.barrier()
        be.close()
    finally:
        td.destroy_process_group()


def test_direct_allreduce_peer_late_beyond_timeout_fails_promptly_and_cleanly():
    """a rank that enters a collective after its peers gave up must not make the others optimise on
    stale sums until max_nfev.  With the time-out of the direct all-reduce at 0.3 s and rank 1 entering its solve
    1.5 s late, BOTH ranks get error -5 (rank 0 from the hand-off that follows the timed-out collective, rank 1 once
    it finds its peer gone), within seconds, without hanging or aborting; the handle then falls back to the
    transport registered before the direct link and solves again with the same result."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_late_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=300) for _ in range(2)), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for o in outs:
        assert o["err"] is not None and "-5" in o["err"] and "timed out" in o["err"], o
        assert o["seconds"] < 10.0, o
        assert o["callback_calls"] > 10                      # the fallback transport carried the last solve
        assert abs(o["cost_again"] - o["cost_ok"]) <= 1e-12 * o["cost_ok"] and o["same"]
