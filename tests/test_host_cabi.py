"""CPU-side checks of the boundary: libsfmba.so loads without a GPU and exports every symbol
include/sfmba.h declares; host-only logic (2-D trust-region solve, argument validation, sparsity
builder) behaves like the scipy / reference code it replaces.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

LIB = os.path.join(ROOT, "sfm-python_amd", "sfmba", "libsfmba.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as ge
        ge.build()
    return ctypes.CDLL(LIB)


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "sfmba.h")).read()
    declared = set(re.findall(r"\b(sfmba_[a-z0-9_]+)\s*\(", header))
    declared -= {"sfmba_allreduce_fn"}
    assert len(declared) >= 15
    from sfmba import _capi
    assert declared == set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    from sfmba import _capi
    assert ctypes.sizeof(_capi.Options) == 80
    assert ctypes.sizeof(_capi.Result) == 128


def test_one_hip_runtime_per_process_when_torch_is_installed():
    """libsfmba.so and a PyTorch-ROCm wheel both want `libamdhip64.so.7`; the wheel ships its own copy.  Loaded in the
    wrong order (the system's copy first), `torch.cuda` later finds no GPU in the process.  `_capi.load()` therefore loads
    torch's copy first when torch is installed -- here: a fresh interpreter that never imports torch ends up with exactly
    one libamdhip64 mapped, the wheel's; SFMBA_HIP_RUNTIME=system keeps the system's."""
    import ast
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.find_spec("torch")
    if spec is None or not os.path.exists(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")):
        pytest.skip("no PyTorch-ROCm wheel with its own HIP runtime in this environment")
    code = ("import sys; sys.path[:0] = [%r]; from sfmba import _capi; _capi.load(); "
            "print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})); "
            "print('torch' in sys.modules)") % os.path.join(ROOT, "sfm-python_amd")
    for env_extra, want_torch_copy in (({}, True), ({"SFMBA_HIP_RUNTIME": "system"}, False)):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                             env={**os.environ, **env_extra})
        assert out.returncode == 0, out.stderr
        libs, torch_imported = out.stdout.strip().splitlines()[-2:]
        libs = ast.literal_eval(libs)
        assert torch_imported == "False"                       # located, not imported
        assert len(libs) == 1, libs
        assert (os.sep + "torch" + os.sep in libs[0]) == want_torch_copy, libs


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sfmba
    with pytest.raises(sfmba.BackendError, match="no CPU fallback"):
        sfmba.Backend(0)
    pb = sfmba.make_problem(3, 8, 20, seed=0)
    with pytest.raises(sfmba.BackendError):
        sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", method="trf", args=pb.args)


def test_trust_region_2d_matches_scipy(lib):
    from scipy.optimize._lsq.common import solve_trust_region_2d
    lib.sfmba_tr2d_solve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    rng = np.random.default_rng(0)
    for trial in range(400):
        A = rng.normal(size=(3, 2)) * 10 ** rng.uniform(-3, 3)
        B = A.T @ A                                   # PSD, as J_S^T J_S always is
        if trial % 7 == 0:
            B = np.outer(A[0], A[0])                  # singular
        if trial % 11 == 0:
            B = B - 1.5 * np.trace(B) * np.eye(2)     # indefinite (never produced by the solver)
        g = rng.normal(size=2) * 10 ** rng.uniform(-3, 3)
        Delta = 10 ** rng.uniform(-4, 4)
        p_ref, newton_ref = solve_trust_region_2d(B, g, Delta)
        B3 = (ctypes.c_double * 3)(B[0, 0], B[0, 1], B[1, 1])
        g2 = (ctypes.c_double * 2)(*g)
        p2 = (ctypes.c_double * 2)()
        newton = lib.sfmba_tr2d_solve(B3, g2, Delta, p2)
        p = np.array(p2[:])
        val = lambda q: 0.5 * q @ B @ q + g @ q      # noqa: E731
        assert np.linalg.norm(p) <= Delta * (1 + 1e-12)
        # same model value (the minimiser can be non-unique in degenerate cases)
        assert val(p) <= val(p_ref) + 1e-9 * max(1.0, abs(val(p_ref)))
        if newton_ref:
            assert newton == 1 and np.allclose(p, p_ref, rtol=1e-9, atol=1e-12 * Delta)


def test_least_squares_argument_validation():
    import sfmba
    pb = sfmba.make_problem(3, 8, 20, seed=0)
    f = sfmba.compute_residuals
    with pytest.raises(ValueError, match="trf"):
        sfmba.least_squares(f, pb.x0, method="lm", x_scale="jac", args=pb.args)
    with pytest.raises(ValueError, match="x_scale"):
        sfmba.least_squares(f, pb.x0, method="trf", x_scale=1.0, args=pb.args)
    with pytest.raises(ValueError, match="verbose"):
        sfmba.least_squares(f, pb.x0, method="trf", x_scale="jac", verbose=3, args=pb.args)
    with pytest.raises(ValueError, match="x0"):
        sfmba.least_squares(f, pb.x0[:-1], method="trf", x_scale="jac", args=pb.args)
    with pytest.raises(ValueError, match="wrong shape"):
        sfmba.least_squares(f, pb.x0, method="trf", x_scale="jac", args=pb.args,
                            jac_sparsity=np.zeros((2, 2)))
    with pytest.raises(ValueError, match="bounds"):
        sfmba.least_squares(f, pb.x0, method="trf", x_scale="jac", args=pb.args, bounds=(0, 1))


def test_fun_is_checked_against_the_model():
    """A foreign `fun` is compared with the model on a few observations before anything is optimised."""
    import sfmba
    from sfmba import api
    from oracle import ba_oracle as orc
    pb = sfmba.make_problem(3, 8, 20, seed=0)
    api._check_fun(orc.compute_residuals, pb.x0, *pb.args)                  # the right model passes
    api._check_fun(orc.compute_residuals_loop, pb.x0, *pb.args)
    with pytest.raises(ValueError, match="does not compute"):
        api._check_fun(lambda x, *a: orc.compute_residuals(x, *a) * 2.0, pb.x0, *pb.args)
    with pytest.raises(ValueError, match="could not be evaluated"):
        api._check_fun(lambda x: x, pb.x0, *pb.args)
    with pytest.raises(ValueError, match="does not compute"):            # checked before the GPU is touched
        sfmba.least_squares(lambda x, *a: np.zeros(12), pb.x0, x_scale="jac", method="trf", args=pb.args)


def test_create_sparsity_matrix_matches_reference_capture():
    import sfmba
    g = np.load(os.path.join(GOLDEN, "sparsity_cases.npz"))
    C, P, N = (int(v) for v in g["dims"])
    for tag in ("free", "fixed"):
        M = sfmba.create_sparsity_matrix(C, P, N, g["ci"], g["pi"],
                                         fixed_camera_indices=tuple(g[tag + "_fixed"]))
        assert M.format == "lil" and M.dtype == int and M.shape == (2 * N, 6 * C + 3 * P)
        M = M.tocsr()
        M.sort_indices()
        assert np.array_equal(M.indptr, g[tag + "_indptr"])
        assert np.array_equal(M.indices, g[tag + "_indices"])
        assert np.all(M.data == 1)


def test_sparsity_matrix_is_built_only_when_looked_into():
    """The pattern object is a lil_matrix whose rows appear at first use (the solver reads its shape only):
    equal to the eagerly built one, unaffected by later changes of the caller's index arrays, usable by
    scipy itself, picklable."""
    import copy
    import pickle
    import scipy.sparse as sp
    from scipy.optimize import least_squares as scipy_least_squares
    import sfmba
    from oracle import ba_oracle as orc
    pb = sfmba.make_problem(4, 30, 120, seed=11)
    ci, pi = pb.camera_indices.copy(), pb.point_indices.copy()
    eager = sfmba.create_sparsity_matrix(4, 30, 120, ci, pi, lazy=False)
    assert type(eager) is sp.lil_matrix
    lazy = sfmba.create_sparsity_matrix(4, 30, 120, ci, pi)
    assert isinstance(lazy, sp.lil_matrix) and sp.issparse(lazy) and lazy.format == "lil"
    assert lazy.shape == eager.shape and lazy.dtype == eager.dtype
    assert "rows" not in lazy.__dict__                     # nothing built so far
    ci[:] = 0                                              # the caller's arrays change afterwards
    assert (lazy.tocsr() != eager.tocsr()).nnz == 0 and "rows" in lazy.__dict__
    for clone in (pickle.loads(pickle.dumps(sfmba.create_sparsity_matrix(4, 30, 120, pb.camera_indices, pi))),
                  copy.deepcopy(sfmba.create_sparsity_matrix(4, 30, 120, pb.camera_indices, pi))):
        assert type(clone) is sp.lil_matrix and (clone.tocsr() != eager.tocsr()).nnz == 0
    # scipy's own driver takes it as jac_sparsity (this is the reference's call with the CPU residual)
    S = sfmba.create_sparsity_matrix(4, 30, 120, pb.camera_indices, pb.point_indices)
    a = scipy_least_squares(orc.compute_residuals, pb.x0, jac_sparsity=S, x_scale="jac", ftol=1e-10,
                            method="trf", args=pb.args, max_nfev=5)
    b = scipy_least_squares(orc.compute_residuals, pb.x0, jac_sparsity=eager, x_scale="jac", ftol=1e-10,
                            method="trf", args=pb.args, max_nfev=5)
    assert a.nfev == b.nfev and np.array_equal(a.x, b.x)


def test_pack_unpack_mirror_matches_scipy_rotations():
    import sfmba
    g = np.load(os.path.join(GOLDEN, "pack_cases.npz"))
    H = g["H"][4:10]
    registered = [True, False, True, True, False, True]
    X3d = np.arange(15, dtype=np.float64).reshape(5, 3)
    obs = [(0, 0, (10, 20)), (0, 2, (11, 21)), (1, 3, (5, 6)), (3, 5, (7, 8)), (4, 0, (1, 2))]
    x0, nc, npnt, ci, pi, uv, cmap = sfmba.pack_cameras_points(H, registered, X3d, obs)
    assert (nc, npnt) == (4, 5) and cmap == {0: 0, 2: 1, 3: 2, 5: 3}
    for cam_id, n in cmap.items():
        assert np.allclose(x0[6 * n:6 * n + 3], g["rotvec_from_matrix"][4 + cam_id], atol=1e-12)
        assert np.array_equal(x0[6 * n + 3:6 * n + 6], H[cam_id][:3, 3])
    H2, X2 = sfmba.unpack_cameras_points(x0, nc, npnt, cmap, H)
    assert all(np.allclose(a, b, atol=1e-12) for a, b in zip(H2, H))
    assert np.array_equal(X2, X3d)


def test_pack_unpack_rt_convention_round_trip():
    """pose_convention='rt' (SURVEY.md section 8f-1): H = [R | t] with x_cam = R X + t (sfm.py:135,212) is handed
    to the model as the camera centre -R^T t, so that the BA residual IS the pipeline's reprojection error
    (cv2_lite/solve_pnp.py:9-15); 'reference' keeps the reference's verbatim copy (sfm.py:252)."""
    import sfmba
    from oracle import ba_oracle as orc
    g = np.load(os.path.join(GOLDEN, "pack_cases.npz"))
    H = g["H"][4:8]
    registered = [True, True, False, True]
    X3d = np.random.default_rng(0).normal(size=(6, 3)) + np.array([0.0, 0.0, 8.0])
    obs = [(p, c, (3 * p + c, 7 * p - c)) for p in range(6) for c in (0, 1, 3)]
    x0, nc, npnt, ci, pi, uv, cmap = sfmba.pack_cameras_points(H, registered, X3d, obs, pose_convention="rt")
    for cam_id, n in cmap.items():
        R, t = H[cam_id][:3, :3], H[cam_id][:3, 3]
        assert np.allclose(x0[6 * n + 3:6 * n + 6], -R.T @ t, atol=1e-13)
    r = orc.compute_residuals(x0, nc, npnt, ci, pi, uv, sfmba.K_SCEAUX).reshape(-1, 2)
    for k, (p, c, px) in enumerate(obs):                       # K (R X + t) projected minus the pixel
        q = sfmba.K_SCEAUX @ (H[c][:3, :3] @ X3d[p] + H[c][:3, 3])
        assert np.allclose(r[k], q[:2] / q[2] - np.asarray(px, dtype=float), atol=1e-9)
    H2, X2 = sfmba.unpack_cameras_points(x0, nc, npnt, cmap, H, pose_convention="rt")
    assert all(np.allclose(a, b, atol=1e-12) for a, b in zip(H2, H)) and np.array_equal(X2, X3d)
    with pytest.raises(ValueError):
        sfmba.pack_cameras_points(H, registered, X3d, obs, pose_convention="centre")


def test_bal_reader_round_trip_and_model_mapping(tmp_path):
    """the published BAL text format.  A file generated in the test (no such file ships
    with the reference): camera-major observation order, BAL's own conventions (P = R X + t, p = -P / P.z, pixel =
    f p), read back into the reference's args tuple (sfm.py:268); the BA residual of the mapped problem must be
    BAL's reprojection error, and write_bal -> read_bal must reproduce the problem exactly."""
    import sfmba
    from oracle import ba_oracle as orc
    rng = np.random.default_rng(11)
    C, P = 4, 25
    f0 = 850.0
    rot = rng.normal(0, 0.2, (C, 3))
    t = rng.normal(0, 0.3, (C, 3)) + np.array([0.0, 0.0, -6.0])       # BAL cameras look down -z
    X = rng.normal(0, 1.0, (P, 3))
    obs = [(c, p) for c in range(C) for p in range(P) if rng.random() < 0.7]     # camera-major, as BAL files are
    lines = [f"{C} {P} {len(obs)}"]
    bal_px = []
    for c, p in obs:
        Pc = orc.rodrigues(rot[c]) @ X[p] + t[c]
        px = -f0 * Pc[:2] / Pc[2] + rng.normal(0, 0.3, 2)
        bal_px.append(px)
        lines.append(f"{c} {p} {float(px[0])!r} {float(px[1])!r}")
    for c in range(C):
        lines += [repr(float(v)) for v in (*rot[c], *t[c], f0, 0.0, 0.0)]
    for p in range(P):
        lines += [repr(float(v)) for v in X[p]]
    path = tmp_path / "problem.txt"
    path.write_text("\n".join(lines) + "\n")
    x0, args, info = sfmba.read_bal(path)
    nC, nP, ci, pi, uv, K = args
    assert (nC, nP, len(ci)) == (C, P, len(obs)) and info["exact"] and info["focal_used"] == f0
    assert np.all(np.diff(pi) >= 0)                                   # point-major
    r = orc.compute_residuals(x0, *args).reshape(-1, 2)
    want = np.empty_like(r)
    for k, j in enumerate(info["file_order"]):                        # BAL's own reprojection error, per observation
        c, p = obs[j]
        Pc = orc.rodrigues(rot[c]) @ X[p] + t[c]
        want[k] = -f0 * Pc[:2] / Pc[2] - bal_px[j]
    assert np.abs(r - want).max() < 1e-9
    out = tmp_path / "again.txt.gz"
    sfmba.write_bal(out, x0, *args)
    x1, args1, info1 = sfmba.read_bal(out)
    assert np.array_equal(x1[6 * C:], x0[6 * C:]) and np.abs(x1[:6 * C] - x0[:6 * C]).max() < 1e-12
    assert all(np.array_equal(a, b) for a, b in zip(args1[2:5], args[2:5])) and np.array_equal(args1[5], K)
    with pytest.raises(ValueError):
        (tmp_path / "bad.txt").write_text("3 4 5\n0 0 1.0 2.0\n")
        sfmba.read_bal(tmp_path / "bad.txt")
    with pytest.raises(ValueError):
        sfmba.write_bal(tmp_path / "k.txt", x0, nC, nP, ci, pi, uv, sfmba.K_SCEAUX)      # principal point: not BAL


def _arr(v):
    assert isinstance(v, np.ndarray), type(v)
    return v


def _upd(o):
    d = {}
    d.update(o)
    return d


def test_lazy_result_behaves_like_scipys_without_a_gpu():
    """The host logic of `result.fun` / `result.grad` staying on the device (sfmba.api.LazyResult, Backend._flush_pending),
    against a stand-in for the back end: downloaded on first access through any dict path, exactly once; downloaded
    before the handle's next operation when the result is still alive; never when it was dropped."""
    import gc
    import pickle
    import weakref
    from sfmba import api

    class FakeBackend:
        def __init__(self):
            self.fetches = 0
            self._pending = None
            self.generation = 0

        def _register_pending(self, obj):
            self._pending = weakref.ref(obj)

        def _flush_pending(self):                       # what Backend does before every operation on the handle
            ref, self._pending = self._pending, None
            if ref is not None and ref() is not None:
                ref()._materialize()

        def fetch_fun_grad(self, want_fun=True, want_grad=True):
            self.fetches += 1
            return np.full(4, float(self.generation)), np.full(3, -float(self.generation))

        def next_operation(self):
            self._flush_pending()
            self.generation += 1

    class Res:
        status, cost, optimality, nfev, njev, iterations, pcg_iterations = 2, 1.5, 1e-9, 6, 5, 5, 18
        rmse = rmse0 = cost0 = seconds_total = seconds_device = resjac_avg_us = 0.0
        resjac_launches = 0

    be = FakeBackend()
    x = np.arange(3.0)
    r = api._make_result(x, Res(), be, verbose=0)
    assert be.fetches == 0 and r.status == 2 and r.success and r.x is x and set(("fun", "grad")) <= set(r.keys())
    assert np.array_equal(r.fun, np.zeros(4)) and be.fetches == 1          # first access downloads ...
    assert np.array_equal(r["grad"], -np.zeros(3)) and r.get("fun") is r.fun and be.fetches == 1       # ... once
    be.next_operation()
    assert be.fetches == 1
    kept = api._make_result(x, Res(), be, verbose=0)                         # alive across the next operation:
    be.next_operation()                                                      # downloaded right before it
    assert be.fetches == 2 and np.array_equal(kept.fun, np.full(4, 1.0))     # (generation 1 = its own solve)
    dropped = api._make_result(x, Res(), be, verbose=0)
    del dropped
    gc.collect()
    be.next_operation()
    assert be.fetches == 2                                                   # dropped unread: no download
    for access in (lambda o: dict(o.items())["fun"], lambda o: list(o.values()), lambda o: repr(o), lambda o: o.copy()["fun"],
                   lambda o: pickle.loads(pickle.dumps(o)).fun, lambda o: o == {},
                   # CPython's dict fast paths (PyDict_Merge and friends) must not hand out the placeholder None
                   lambda o: _arr(dict(o)["fun"]), lambda o: _arr({**o}["grad"]), lambda o: _arr(_upd(o)["fun"]),
                   lambda o: _arr((o | {})["fun"]), lambda o: _arr(({} | o)["grad"]), lambda o: _arr(o.pop("fun")),
                   lambda o: _arr(o.setdefault("fun", 7)), lambda o: _arr(dict(o.popitem() for _ in range(len(o)))["fun"])):
        o = api._make_result(x, Res(), be, verbose=0)
        n = be.fetches
        access(o)
        assert be.fetches == n + 1 and (isinstance(o.get("fun", np.zeros(1)), np.ndarray))
        assert be._pending is None                                           # nothing left to flush
    o = api._make_result(x, Res(), be, verbose=0)
    with pytest.raises(AttributeError):
        o.no_such_field


def test_held_cameras_are_read_from_the_sparsity_pattern():
    """api.fixed_cameras_of: the pattern of create_sparsity_matrix(..., fixed_camera_indices) (bundle_adjustment.py:13-14)
    says which cameras scipy would never move; this module's lazy pattern knows, any other matrix is read row by row,
    and a pattern that is not the bundle-adjustment block pattern raises instead of being ignored."""
    import sfmba
    from sfmba.api import fixed_cameras_of
    from sfmba.synthetic import make_problem
    g = np.load(os.path.join(GOLDEN, "sparsity_cases.npz"))
    C, P, N = (int(v) for v in g["dims"])
    ci, pi = g["ci"], g["pi"]
    import scipy.sparse as sp
    for tag in ("free", "fixed"):
        fixed = tuple(int(c) for c in g[tag + "_fixed"])
        ref = sp.csr_matrix((g[tag + "_data"], g[tag + "_indices"], g[tag + "_indptr"]), shape=tuple(g[tag + "_shape"]))   # the reference's own
        lazy = sfmba.create_sparsity_matrix(C, P, N, ci, pi, fixed_camera_indices=fixed)
        assert fixed_cameras_of(lazy, C, P, ci) == fixed and "_sfmba_builder" in lazy.__dict__        # without building it
        for S in (ref, ref.tolil(), ref.tocsc(), ref.toarray(), sfmba.create_sparsity_matrix(C, P, N, ci, pi, fixed, lazy=False)):
            assert fixed_cameras_of(S, C, P, ci) == fixed
    ref = sp.csr_matrix((g["fixed_data"], g["fixed_indices"], g["fixed_indptr"]), shape=tuple(g["fixed_shape"])).tolil()
    k = int(np.flatnonzero(ci == 1)[0])                           # camera 1 is free: drop its columns in ONE observation
    ref[2 * k, 6:12] = 0
    ref[2 * k + 1, 6:12] = 0
    with pytest.raises(ValueError, match="some of its observations"):
        fixed_cameras_of(ref.tocsr(), C, P, ci)
    bad = sp.csr_matrix((g["free_data"], g["free_indices"], g["free_indptr"]), shape=tuple(g["free_shape"])).tolil()
    bad[0, int(6 * ci[0])] = 0
    for S in (bad, bad.tocsr()):
        with pytest.raises(ValueError, match="not the bundle-adjustment pattern"):
            fixed_cameras_of(S, C, P, ci)
    explicit = sp.csr_matrix((g["free_data"], g["free_indices"], g["free_indptr"]), shape=tuple(g["free_shape"])).astype(float)
    explicit.data[:] = 1.0
    explicit[0, 6 * C + 3 * P - 1] = 0.0                           # an explicit zero is no structure for scipy either
    assert fixed_cameras_of(explicit, C, P, ci) == ()
