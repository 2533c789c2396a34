"""Host-side mirror of the reference's bundle-adjustment interface, backed by the HIP path.

Same names, argument meaning and error behaviour as the code it replaces:

* ``least_squares``          <- ``scipy.optimize.least_squares`` as called at
                                /root/reference/sfm_lite/sfm.py:266-268
* ``compute_residuals``      <- /root/reference/sfm_lite/bundle_adjustment.py:35-42
* ``project_points``         <- /root/reference/sfm_lite/bundle_adjustment.py:20-32
* ``create_sparsity_matrix`` <- /root/reference/sfm_lite/bundle_adjustment.py:6-17
* ``pack_cameras_points`` / ``unpack_cameras_points`` <- sfm.py:248-262 / 271-281

Everything numeric runs on the GPU through libsfmba.so; there is no CPU fallback.
"""
from __future__ import annotations

import threading

import numpy as np

from . import _capi
from .backend import Backend

# scipy's termination messages (SCIPY/optimize/_lsq/least_squares.py:18-25)
TERMINATION_MESSAGES = {
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}

try:                                     # scipy is a dependency of the reference, so normally present
    from scipy.optimize import OptimizeResult
except Exception:                        # pragma: no cover
    class OptimizeResult(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

_local = threading.local()


def get_backend(device: int = 0) -> Backend:
    """One Backend per (thread, device): the reference may call BA from a worker thread
    (/root/reference/app.py:80-85) and a handle is not thread-safe."""
    cache = getattr(_local, "cache", None)
    if cache is None:
        cache = _local.cache = {}
    be = cache.get(device)
    if be is None:
        be = cache[device] = Backend(device)
    return be


def _split_args(args):
    if len(args) != 6:
        raise ValueError("args must be (n_cameras, n_points, camera_indices, point_indices, points_2d, K) "
                         "as in sfm_lite/sfm.py:268")
    return args


def _build_sparsity(n_cameras, n_points, n_obs, ci, pi, fixed_camera_indices):
    from scipy.sparse import csr_matrix
    fixed = np.isin(ci, np.asarray(list(fixed_camera_indices), dtype=np.int64))
    cols_c = ci[:, None] * 6 + np.arange(6)[None, :]
    cols_p = n_cameras * 6 + pi[:, None] * 3 + np.arange(3)[None, :]
    rows_c = np.repeat(np.arange(n_obs)[~fixed] * 2, 6)
    rows_p = np.repeat(np.arange(n_obs) * 2, 3)
    rows = np.concatenate([rows_c, rows_c + 1, rows_p, rows_p + 1])
    cc = cols_c[~fixed].ravel()
    cp = cols_p.ravel()
    cols = np.concatenate([cc, cc, cp, cp])
    M = csr_matrix((np.ones(len(rows), dtype=int), (rows, cols)),
                   shape=(n_obs * 2, n_cameras * 6 + n_points * 3), dtype=int)
    M.data[:] = 1                                   # duplicated (cam, point) pairs still give 1
    return M.tolil()


def _lazy_lil_class():
    """lil_matrix whose row lists are built at first use.  The reference builds this matrix before every BA
    call (sfm.py:264) only to hand it to ``least_squares``; this back end reads nothing but its shape, and
    even the vectorised construction of 18 non-zeros per observation costs ~1.7 s at 1M observations -- five
    hundred times the solve.  Anything that looks inside (``.rows``, ``.data``, conversions, arithmetic,
    scipy's own ``least_squares``) triggers the construction and sees an ordinary ``lil_matrix``."""
    global _LazyLil
    if _LazyLil is None:
        from scipy.sparse import lil_matrix

        class LazyLil(lil_matrix):
            def __init__(self, shape, builder):           # lil_matrix.__init__ is NOT called: no row lists yet
                self._shape = (int(shape[0]), int(shape[1]))
                self.dtype = np.dtype(int)
                self.maxprint = 50
                self._sfmba_builder = builder
                self._sfmba_fixed = ()

            def __getattr__(self, name):                  # only reached for attributes that do not exist yet
                if name in ("rows", "data") and "_sfmba_builder" in self.__dict__:
                    built = self.__dict__.pop("_sfmba_builder")()
                    self.__dict__["rows"], self.__dict__["data"] = built.rows, built.data
                    return self.__dict__[name]
                raise AttributeError(name)

            def __reduce__(self):                         # pickling / copy.deepcopy: as a plain lil_matrix
                from scipy.sparse import lil_matrix as _lil
                return (_lil, (self.tocsr(),))

        _LazyLil = LazyLil
    return _LazyLil


_LazyLil = None


def create_sparsity_matrix(n_cameras, n_points, n_obs, camera_indices, point3d_indices,
                           fixed_camera_indices=(), lazy=True):
    """Same 0/1 ``lil_matrix`` (2 n_obs, 6 n_cameras + 3 n_points), dtype int, as the reference builds
    with a Python loop (bundle_adjustment.py:9-15); built here from index arithmetic, and (``lazy``) only
    when something looks inside it: the solver reads its shape alone -- the block structure is implied by
    the index arrays."""
    ci = np.array(camera_indices, dtype=np.int64)           # copies: the pattern must not change if the
    pi = np.array(point3d_indices, dtype=np.int64)          # caller's arrays do before it is built
    assert len(ci) == len(pi)
    fixed = tuple(fixed_camera_indices)

    def build():
        return _build_sparsity(n_cameras, n_points, n_obs, ci, pi, fixed)

    if not lazy:
        return build()
    out = _lazy_lil_class()((n_obs * 2, n_cameras * 6 + n_points * 3), build)
    out._sfmba_fixed = tuple(sorted(set(int(c) for c in fixed if 0 <= int(c) < n_cameras)))
    return out


def fixed_cameras_of(jac_sparsity, n_cameras, n_points, camera_indices):
    """The cameras a ``jac_sparsity`` pattern holds still: ``create_sparsity_matrix(..., fixed_camera_indices)``
    leaves their six columns empty (bundle_adjustment.py:13-14), scipy's sparse finite differences then never fill
    them, and the parameters stay where they are.  A pattern built by this module knows its list; any other sparse
    matrix / array is read row by row: rows 2i, 2i+1 must hold the 3 point columns plus either the 6 camera columns
    (free) or nothing (held still), the same for every observation of a camera -- anything else is not the
    bundle-adjustment pattern this back end implements and raises ``ValueError`` instead of being ignored."""
    known = getattr(jac_sparsity, "__dict__", {}).get("_sfmba_fixed")
    if known is not None:
        return tuple(known)
    from scipy.sparse import issparse, csr_matrix
    ci = np.asarray(camera_indices, dtype=np.int64)
    n_obs = len(ci)
    bad = ValueError("`jac_sparsity` is not the bundle-adjustment pattern of create_sparsity_matrix (3 point columns "
                     "per row, plus the 6 camera columns unless the camera is held still)")
    if issparse(jac_sparsity) and jac_sparsity.format == "lil":
        rows = jac_sparsity.rows
        lens = np.fromiter(map(len, rows), dtype=np.int64, count=2 * n_obs)
        first = None
        if lens.sum() == 18 * n_obs and lens.min(initial=9) == 9:
            return ()
        first = np.fromiter((r[0] if r else -1 for r in rows), dtype=np.int64, count=2 * n_obs)
    else:
        S = jac_sparsity.tocsr() if issparse(jac_sparsity) else csr_matrix(np.atleast_2d(np.asarray(jac_sparsity)))
        if S.nnz and not np.all(S.data):                      # explicit zeros are not part of the structure for scipy
            S = S.copy()
            S.eliminate_zeros()
        S.sort_indices()
        lens = np.diff(S.indptr).astype(np.int64)
        if S.nnz == 18 * n_obs and lens.min(initial=9) == 9:
            return ()
        first = np.full(2 * n_obs, -1, dtype=np.int64)
        nz = lens > 0
        first[nz] = S.indices[S.indptr[:-1][nz]]
    lens = lens.reshape(n_obs, 2)
    first = first.reshape(n_obs, 2)
    if np.any(lens[:, 0] != lens[:, 1]) or np.any((lens[:, 0] != 3) & (lens[:, 0] != 9)):
        raise bad
    held = lens[:, 0] == 3
    if np.any(first[held] < 6 * n_cameras) or np.any(first[~held] != 6 * ci[~held][:, None]):
        raise bad
    fixed = np.zeros(n_cameras, dtype=bool)
    fixed[ci[held]] = True
    if np.any(fixed[ci[~held]]):
        raise ValueError("`jac_sparsity` holds a camera still in some of its observations and not in others")
    return tuple(int(c) for c in np.flatnonzero(fixed))


def compute_residuals(x, n_cameras, n_points, camera_indices, point_indices, points_2d, K, device=0):
    """(2N,) interleaved residuals pi(K R(w)(X - T)) - uv, evaluated by the HIP kernel."""
    be = get_backend(device)
    be.set_precision(64)
    be.set_fixed_cameras(())
    be.set_problem(n_cameras, n_points, camera_indices, point_indices, points_2d, K)
    return be.residuals(x)


def project_points(points, camera_params, K, device=0):
    """(N,3), (N,6) -> (N,2) projections, one camera row per point as in bundle_adjustment.py:20-32."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    camera_params = np.ascontiguousarray(camera_params, dtype=np.float64)
    n = len(points)
    x = np.concatenate([camera_params.ravel(), points.ravel()])
    idx = np.arange(n, dtype=np.int64)
    r = compute_residuals(x, n, n, idx, idx, np.zeros((n, 2)), K, device=device)
    return r.reshape(n, 2)


def least_squares(fun, x0, jac="2-point", bounds=(-np.inf, np.inf), method="trf", ftol=1e-8, xtol=1e-8,
                  gtol=1e-8, x_scale=1.0, loss="linear", f_scale=1.0, diff_step=None, tr_solver=None,
                  tr_options=None, jac_sparsity=None, max_nfev=None, verbose=0, args=(), kwargs=None,
                  device=0, max_iter=None, pcg_tol=None, profile=False, return_jac=False, storage_bits=64,
                  check_fun=True, backend=None):
    """Drop-in for the reference's ``least_squares(compute_residuals, x0, jac_sparsity=..., verbose=...,
    x_scale='jac', ftol=tol, method='trf', args=(...))`` (sfm.py:266-268).

    ``fun`` must be the bundle-adjustment residual (the reference's ``compute_residuals`` or this
    module's): the optimisation does not call it -- the same model runs as a HIP kernel with an analytic
    Jacobian.  With ``check_fun=True`` (default) a foreign ``fun`` is evaluated once on the first few
    observations and compared with the model this back end implements; a mismatch raises ``ValueError``
    instead of silently optimising a different function.
    ``jac_sparsity`` is shape-checked and read for cameras it holds still (empty camera columns, the
    ``fixed_camera_indices`` of ``create_sparsity_matrix``): their parameters do not move, as with scipy; a pattern
    that is not the bundle-adjustment block pattern raises ``ValueError``.  ``jac``, ``diff_step``, ``tr_solver``,
    ``tr_options`` are accepted and ignored (the Jacobian is analytic, the trust-region step comes
    from the Schur-complement PCG).  Unsupported: bounds, robust losses, methods other than 'trf',
    x_scale other than 'jac'.  ``return_jac=True`` fills ``result.jac`` with the analytic Jacobian at
    ``result.x`` in scipy's CSR layout (2N x (6C+3P), 9 entries per row); by default it is ``None``
    because the reference only reads ``result.x`` (sfm.py:271,281).  ``storage_bits=32`` keeps the
    per-observation streams (pixels, residuals, Jacobian) in fp32 with fp64 arithmetic and accumulation.
    ``backend``: a :class:`Backend` to run on instead of the calling thread's own (``device`` is then ignored).
    """
    if method != "trf":
        raise ValueError("sfmba.least_squares implements method='trf' only (the reference's choice).")
    if not (isinstance(x_scale, str) and x_scale == "jac"):
        raise ValueError("sfmba.least_squares implements x_scale='jac' only (the reference's choice).")
    if loss != "linear":
        raise ValueError("only loss='linear' is supported")
    lb, ub = bounds
    if np.any(np.isfinite(np.atleast_1d(lb))) or np.any(np.isfinite(np.atleast_1d(ub))):
        raise ValueError("bounds are not supported (the reference passes none)")
    if verbose not in (0, 1, 2):
        raise ValueError("`verbose` must be in [0, 1, 2].")
    if max_nfev is not None and max_nfev <= 0:
        raise ValueError("`max_nfev` must be None or positive integer.")
    if kwargs:
        raise ValueError("kwargs are not supported; pass the BA arguments through args")
    n_cameras, n_points, camera_indices, point_indices, points_2d, K = _split_args(tuple(args))
    x0 = np.atleast_1d(np.asarray(x0, dtype=np.float64))
    if x0.ndim > 1:
        raise ValueError("`x0` must have at most 1 dimension.")
    n = 6 * int(n_cameras) + 3 * int(n_points)
    if x0.shape[0] != n:
        raise ValueError(f"`x0` has {x0.shape[0]} elements, expected 6*n_cameras + 3*n_points = {n}")
    n_obs = len(camera_indices)
    if jac_sparsity is not None and tuple(jac_sparsity.shape) != (2 * n_obs, n):
        raise ValueError("`jac_sparsity` has wrong shape.")          # least_squares.py:160-161

    if check_fun and fun is not compute_residuals:
        _check_fun(fun, x0, int(n_cameras), int(n_points), camera_indices, point_indices, points_2d, K)

    fixed = () if jac_sparsity is None else fixed_cameras_of(jac_sparsity, int(n_cameras), int(n_points), camera_indices)
    be = backend if backend is not None else get_backend(device)
    be.set_precision(storage_bits)
    be.set_fixed_cameras(fixed)
    be.set_problem(n_cameras, n_points, camera_indices, point_indices, points_2d, K)
    opt = be.default_options()
    opt.ftol = 0.0 if ftol is None else float(ftol)
    opt.xtol = 0.0 if xtol is None else float(xtol)
    opt.gtol = 0.0 if gtol is None else float(gtol)
    opt.max_nfev = 0 if max_nfev is None else int(max_nfev)
    opt.verbose = int(verbose)
    opt.max_iter = 0 if max_iter is None else int(max_iter)
    if pcg_tol is not None:
        opt.pcg_tol = float(pcg_tol)
    opt.profile = 1 if profile else 0
    x, res, _, _ = be.solve(x0, opt, want_fun=False, want_grad=False)
    out = _make_result(x, res, be, verbose)
    if return_jac:
        out.jac = _jacobian_csr(be, x, int(n_cameras), int(n_points), camera_indices, point_indices, fixed)
    return out


def _check_fun(fun, x0, n_cameras, n_points, camera_indices, point_indices, points_2d, K, k=6):
    """Argument validation, not a compute path: does ``fun`` implement pi(K R(w)(X - T)) - uv?  Evaluated
    on the first k observations only (the reference's function costs ~15 us per observation)."""
    ci = np.asarray(camera_indices)[:k]
    pi = np.asarray(point_indices)[:k]
    uv = np.asarray(points_2d)[:k]
    if len(ci) == 0 or not callable(fun):
        raise ValueError("`fun` must be callable and the problem must have observations")
    try:
        got = np.asarray(fun(x0, n_cameras, n_points, ci, pi, uv, K), dtype=np.float64).ravel()
    except Exception as exc:                                   # noqa: BLE001
        raise ValueError(f"`fun` could not be evaluated with the bundle-adjustment arguments: {exc}") from exc
    Kf = np.asarray(K, dtype=np.float64)
    cams = x0[:6 * n_cameras].reshape(n_cameras, 6)
    pts = x0[6 * n_cameras:].reshape(n_points, 3)
    want = np.empty((len(ci), 2))
    for j, (c, p) in enumerate(zip(ci, pi)):
        q = Kf @ (_matrix_from_rotvec(cams[c, :3]) @ (pts[p] - cams[c, 3:]))
        want[j] = q[:2] / q[2] - uv[j]
    want = want.ravel()
    if got.shape != want.shape or not np.allclose(got, want, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(want).max()),
                                                  equal_nan=True):
        raise ValueError("`fun` does not compute the bundle-adjustment residual pi(K R(w)(X - T)) - uv that "
                         "sfmba.least_squares optimises (sfm_lite/bundle_adjustment.py:35-42); pass "
                         "check_fun=False to skip this check")


def _jacobian_csr(be, x, n_cameras, n_points, camera_indices, point_indices, fixed=()):
    from scipy.sparse import csr_matrix
    _, Jc, Jp = be.residual_jacobian(x)
    ci = np.asarray(camera_indices, dtype=np.int64)
    if len(fixed):                                   # columns of cameras held still are zero, as in scipy's result.jac
        Jc[np.isin(ci, np.asarray(fixed, dtype=np.int64))] = 0.0
    pi = np.asarray(point_indices, dtype=np.int64)
    n_obs = len(ci)
    cols = np.concatenate([ci[:, None] * 6 + np.arange(6)[None, :],
                           n_cameras * 6 + pi[:, None] * 3 + np.arange(3)[None, :]], axis=1)
    cols = np.repeat(cols[:, None, :], 2, axis=1).reshape(-1)
    vals = np.concatenate([Jc, Jp], axis=2).reshape(-1)
    indptr = np.arange(0, 18 * n_obs + 1, 9)
    return csr_matrix((vals, cols, indptr), shape=(2 * n_obs, 6 * n_cameras + 3 * n_points))


class LazyResult(OptimizeResult):
    """scipy's OptimizeResult whose ``fun`` (16 MB at a million observations) and ``grad`` stay on the device until
    somebody looks at them.  The reference reads ``result.x`` only (sfm.py:271,281) and drops the object: then they
    are never downloaded (1.2 of the 5.1 ms of a call at 1M observations).  If the object is still alive when the
    back end's next operation is about to overwrite its buffers, they are downloaded at that moment
    (Backend._flush_pending), so a result that is kept stays valid like scipy's."""
    _LAZY = ("fun", "grad")

    def _materialize(self):
        be = self.__dict__.pop("_backend", None)
        if be is not None:
            if be._pending is not None and be._pending() is self:
                be._pending = None
            fun, grad = be.fetch_fun_grad(True, True)
            be.n_lazy_downloads = getattr(be, "n_lazy_downloads", 0) + 1
            dict.__setitem__(self, "fun", fun)
            dict.__setitem__(self, "grad", grad)

    def _touch(self, key):
        if key in self._LAZY and "_backend" in self.__dict__:
            self._materialize()

    def __getitem__(self, key):
        self._touch(key)
        return dict.__getitem__(self, key)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def get(self, key, default=None):
        self._touch(key)
        return dict.get(self, key, default)

    def _all(self):
        if "_backend" in self.__dict__:
            self._materialize()
        return self

    # keys() and __iter__ as Python-level delegations: CPython's dict(res), {**res}, d.update(res) and res | d then
    # take the generic keys() + __getitem__ route instead of copying the underlying table (which holds None for the
    # lazy fields) behind this class's back
    def keys(self):
        return dict.keys(self)

    def __iter__(self):
        return dict.__iter__(self)

    def items(self):
        return dict.items(self._all())

    def pop(self, key, *default):
        self._touch(key)
        return dict.pop(self, key, *default)

    def popitem(self):
        return dict.popitem(self._all())

    def setdefault(self, key, default=None):
        self._touch(key)
        return dict.setdefault(self, key, default)

    def __or__(self, other):
        return OptimizeResult(dict.__or__(dict(self._all()), other))

    def __ror__(self, other):
        return OptimizeResult(dict.__or__(dict(other), dict(self._all())))

    def __ior__(self, other):
        dict.update(self._all(), other)
        return self

    def values(self):
        return dict.values(self._all())

    def copy(self):
        return OptimizeResult(dict.copy(self._all()))

    def __repr__(self):
        self._all()
        return OptimizeResult.__repr__(self)

    def __eq__(self, other):
        return dict.__eq__(self._all(), other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __reduce__(self):
        return (OptimizeResult, (dict(self._all()),))


def _make_result(x, res, be, verbose):
    status = int(res.status)
    out = LazyResult(
        x=x, cost=res.cost, fun=None, jac=None, grad=None, optimality=res.optimality,
        active_mask=np.zeros_like(x), nfev=int(res.nfev), njev=int(res.njev), status=status,
        message=TERMINATION_MESSAGES[status], success=status > 0,
        # extras (not in scipy's result)
        iterations=int(res.iterations), pcg_iterations=int(res.pcg_iterations), rmse=res.rmse,
        rmse0=res.rmse0, cost0=res.cost0, seconds=res.seconds_total, seconds_device=res.seconds_device,
        resjac_avg_us=res.resjac_avg_us, resjac_launches=int(res.resjac_launches))
    out.__dict__["_backend"] = be                  # (instance attribute, not a key of the result)
    be._register_pending(out)
    if verbose >= 1:                                                  # least_squares.py:966-970
        print(out.message)
        print(f"Function evaluations {out.nfev}, initial cost {res.cost0:.4e}, final cost "
              f"{out.cost:.4e}, first-order optimality {out.optimality:.2e}.")
    return out


# ---- pack / unpack of the caller (sfm.py:248-262, 271-281) -----------------------------------------

def _rotvec_from_matrix(R):
    R = np.asarray(R, dtype=np.float64)
    q = np.empty(4)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q[:] = (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    nv = np.linalg.norm(q[:3])
    if nv < 1e-12:
        return 2.0 * q[:3]
    return q[:3] * (2.0 * np.arctan2(nv, q[3]) / nv)


def _matrix_from_rotvec(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-4:
        a, b = 1.0 - th * th / 6.0, 0.5 - th * th / 24.0
    else:
        a, b = np.sin(th) / th, 0.5 * (np.sin(0.5 * th) / (0.5 * th)) ** 2
    return np.eye(3) + a * Wx + b * (Wx @ Wx)


def pack_cameras_points(H_list, registered, X3d, observations, pose_convention="reference"):
    """What sfm.py:248-262 builds from the graph: ``observations`` iterates
    (point_idx, cam_id, (x, y)) in ``Graph.pt3ds_pt2ds`` order; ``H_list[k]`` / ``registered[k]``
    describe node k.

    ``pose_convention="reference"`` (default) copies ``T = H[:3, 3]`` verbatim, as the reference does
    (sfm.py:252,255) although its residual treats T as the camera CENTRE (bundle_adjustment.py:27,
    SURVEY.md section 3.4).  ``"rt"`` is the consistent reading of ``H = [R | t]`` with ``x_cam = R X + t``
    (sfm.py:135,212): the centre ``-R^T t`` is handed to the model, and :func:`unpack_cameras_points` maps
    it back."""
    if pose_convention not in ("reference", "rt"):
        raise ValueError("pose_convention must be 'reference' or 'rt'")
    data = list(observations)
    pt_indices = np.array([d[0] for d in data])
    cam_ids = [d[1] for d in data]
    pt2ds = np.array([d[2] for d in data])
    reg = [k for k, f in enumerate(registered) if f]
    camera_map = {k: i for i, k in enumerate(reg)}
    params = []
    for k in reg:
        H = np.asarray(H_list[k], dtype=np.float64)
        T = H[:3, 3].flatten() if pose_convention == "reference" else -H[:3, :3].T @ H[:3, 3]
        params.append(np.hstack([_rotvec_from_matrix(H[:3, :3]), T]))
    camera_indices = np.array([camera_map[c] for c in cam_ids])
    X3d = np.asarray(X3d, dtype=np.float64)
    x0 = np.hstack([np.hstack(params).ravel(), X3d.ravel()])
    return x0, len(reg), len(X3d), camera_indices, pt_indices, pt2ds, camera_map


def unpack_cameras_points(x, n_cam, n_points, camera_map, H_list, pose_convention="reference"):
    """sfm.py:271-281: new 4x4 poses for the registered cameras and the (n_points, 3) cloud."""
    cams = np.asarray(x[:n_cam * 6]).reshape((n_cam, 6))
    H_out = [np.array(H, dtype=np.float64, copy=True) for H in H_list]
    for cam_id, k in camera_map.items():
        H = np.eye(4)
        H[:3, :3] = _matrix_from_rotvec(cams[k, :3])
        H[:3, 3] = cams[k, 3:] if pose_convention == "reference" else -H[:3, :3] @ cams[k, 3:]
        H_out[cam_id] = H
    return H_out, np.asarray(x[n_cam * 6:]).reshape((n_points, 3))


def apply_bundle_adjustment(H_list, registered, X3d, observations, K, tol=1e-10, verbose=2, device=0,
                            pose_convention="reference"):
    """The whole of ``SFM._apply_bundle_adjustment`` (sfm.py:243-281) on plain arrays: pack, the sparsity
    pattern of sfm.py:264 (built lazily: the solver reads its shape only), the ``least_squares`` call of
    sfm.py:266-268, unpack."""
    x0, n_cam, n_points, ci, pi, uv, cmap = pack_cameras_points(H_list, registered, X3d, observations,
                                                                pose_convention)
    jac_sparsity = create_sparsity_matrix(n_cam, n_points, len(ci), ci, pi)
    res = least_squares(compute_residuals, x0, jac_sparsity=jac_sparsity, verbose=verbose, x_scale="jac",
                        ftol=tol, method="trf", args=(n_cam, n_points, ci, pi, uv, K), device=device)
    return unpack_cameras_points(res.x, n_cam, n_points, cmap, H_list, pose_convention) + (res,)
