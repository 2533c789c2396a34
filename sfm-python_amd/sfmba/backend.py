"""Handle wrapper around the C-ABI: owns one sfmba_handle (one GPU, one problem at a time)."""
from __future__ import annotations

import ctypes as C

import weakref

import numpy as np

from . import _capi


class BackendError(RuntimeError):
    pass


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"{name} has shape {a.shape}, expected {shape}")
    return a


class Backend:
    """One MI355X.  Not thread-safe; use one Backend per thread (include/sfmba.h, Threading)."""

    def __init__(self, device: int = 0):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        rc = self._lib.sfmba_create(C.byref(self._h), int(device))
        if rc != 0:
            self._h = None
            raise BackendError(
                f"sfmba_create(device={device}) failed with code {rc}: no usable MI355X/HIP device. "
                "The bundle-adjustment path has no CPU fallback.")
        self.device = int(device)
        self.n_cameras = self.n_points = self.n_obs = 0
        self._keep = []          # arrays / callbacks the library borrows beyond one call
        # the verbose = 2 iteration table goes through Python's sys.stdout, like scipy's print() calls
        # (looked up at call time, so contextlib.redirect_stdout and notebook capture see it)
        self._print_cb = _capi.PRINT_FN(lambda ctx, line: print(line.decode("utf-8", "replace")))
        self._lib.sfmba_set_print(self._h, self._print_cb, None)

    def close(self):
        if getattr(self, "_h", None):
            try:
                self._flush_pending()
            except Exception:                                  # noqa: BLE001 (a dying handle must still be released)
                pass
            self._lib.sfmba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc == 0:
            return
        msg = self._lib.sfmba_last_error(self._h).decode("utf-8", "replace")
        if rc in (-1, -2):
            raise ValueError(msg)            # scipy raises ValueError for both (least_squares.py:844)
        if rc == -4:
            raise MemoryError(msg)
        raise BackendError(f"sfmba error {rc}: {msg}")

    # ---- results left on the device -----------------------------------------------------------------
    # The reference reads only ``result.x`` (sfm.py:271,281), and ``result.fun`` is 16 MB at a million observations:
    # `solve(..., want_fun=False, want_grad=False)` leaves fun and grad in the handle's buffers and registers the result object (weakly).  They are
    # downloaded when somebody looks at them, or -- if the object is still alive -- right before this handle's next
    # operation overwrites the buffers; a result that was dropped unread costs no download at all.
    def _flush_pending(self):
        ref, self._pending = getattr(self, "_pending", None), None
        if ref is not None:
            obj = ref()
            if obj is not None:
                obj._materialize()

    def _register_pending(self, obj):
        self._pending = weakref.ref(obj)

    def fetch_fun_grad(self, want_fun=True, want_grad=True):
        """fun (2 n_obs) and / or grad (n) of the last solve, while the handle still holds them."""
        fun = np.empty(2 * self.n_obs) if want_fun else None
        grad = np.empty(self.n_params) if want_grad else None
        if want_fun or want_grad:
            self._check(self._lib.sfmba_get_fun_grad(self._h, _capi.ptr(fun) if want_fun else None,
                                                     _capi.ptr(grad) if want_grad else None))
        return fun, grad

    def set_precision(self, storage_bits: int):
        """64 (default) or 32: storage of uv / r / Jacobian; arithmetic stays fp64.  Applies from the
        next set_problem."""
        self._flush_pending()
        self._check(self._lib.sfmba_set_precision(self._h, int(storage_bits)))

    def set_fixed_cameras(self, camera_indices=()):
        """Cameras whose six parameters stay as they are (``create_sparsity_matrix(..., fixed_camera_indices)``).
        Applies from the next set_problem until replaced."""
        idx = np.ascontiguousarray(sorted(set(int(c) for c in camera_indices)), dtype=np.int64)
        if tuple(idx) == getattr(self, "_fixed", ()):
            return
        self._flush_pending()
        self._check(self._lib.sfmba_set_fixed_cameras(self._h, _capi.ptr(idx) if len(idx) else None, len(idx)))
        self._fixed = tuple(int(c) for c in idx)

    def debug_option(self, name: str, value: int):
        """Test / diagnostic hook (include/sfmba.h: sfmba_debug_option)."""
        self._check(self._lib.sfmba_debug_option(self._h, name.encode(), int(value)))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.sfmba_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def set_problem(self, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
        self._flush_pending()
        ci = np.ascontiguousarray(camera_indices, dtype=np.int64).ravel()
        pi = np.ascontiguousarray(point_indices, dtype=np.int64).ravel()
        if ci.shape != pi.shape:
            raise ValueError("camera_indices and point_indices differ in length")
        n_obs = ci.shape[0]
        Kc = _f64(K, (3, 3), "K")
        p2 = np.asarray(points_2d)
        if p2.dtype == np.int64 and p2.shape == (n_obs, 2) and p2.flags.c_contiguous:
            # the reference's own pixel arrays (graph.py:112-113): handed over as they are
            self._check(self._lib.sfmba_set_problem_i64(self._h, int(n_cameras), int(n_points), n_obs,
                                                        _capi.ptr(ci), _capi.ptr(pi), _capi.ptr(p2),
                                                        _capi.ptr(Kc)))
        else:
            uv = _f64(points_2d, (n_obs, 2), "points_2d")    # promoted as bundle_adjustment.py:41
            self._check(self._lib.sfmba_set_problem(self._h, int(n_cameras), int(n_points), n_obs,
                                                    _capi.ptr(ci), _capi.ptr(pi), _capi.ptr(uv),
                                                    _capi.ptr(Kc)))
        self.n_cameras, self.n_points, self.n_obs = int(n_cameras), int(n_points), n_obs
        self._keep = []

    @property
    def n_params(self):
        return 6 * self.n_cameras + 3 * self.n_points

    def exchange_doubles(self) -> int:
        return int(self._lib.sfmba_exchange_doubles(self.n_cameras))

    def set_exchange(self, arena_ptr: int, arena_doubles: int, callback, n_obs_total: int):
        """callback(dev_ptr:int, count:int, op:int) -> None; op 0 sum, 1 max."""
        if callback is None:
            self._check(self._lib.sfmba_set_exchange(self._h, None, 0, _capi.ALLREDUCE_FN(), None, 0))
            self._keep = []
            return

        def _tramp(ctx, dev_ptr, count, op):
            try:
                callback(int(dev_ptr), int(count), int(op))
                return 0
            except Exception as exc:                         # noqa: BLE001 -- reported through rc
                import traceback
                traceback.print_exc()
                self._cb_error = exc
                return 1

        cfn = _capi.ALLREDUCE_FN(_tramp)
        self._keep = [cfn, callback]
        self._check(self._lib.sfmba_set_exchange(self._h, C.c_void_p(int(arena_ptr)),
                                                 int(arena_doubles), cfn, None, int(n_obs_total)))

    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id; call on rank 0 and broadcast (sfmba.dist.NativeComm does)."""
        buf = C.create_string_buffer(128)
        rc = _capi.load().sfmba_comm_get_unique_id(buf)
        if rc != 0:
            raise BackendError("sfmba_comm_get_unique_id failed: librccl.so.1 not loadable")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int, n_obs_total: int):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self._lib.sfmba_comm_init(self._h, buf, int(rank), int(world), int(n_obs_total)))

    def comm_destroy(self):
        self._check(self._lib.sfmba_comm_destroy(self._h))

    # direct all-reduce over peer-mapped memory (include/sfmba.h: sfmba_p2p_*)
    def p2p_export(self, world: int) -> bytes:
        buf = C.create_string_buffer(64)
        self._check(self._lib.sfmba_p2p_export(self._h, int(world), buf))
        return buf.raw

    def p2p_attach(self, handles: bytes, rank: int, world: int) -> int:
        """Returns the C-ABI code (0 attached, -5 mapping or self-test failed) instead of raising: the
        caller has to agree on the outcome with the other ranks first."""
        if len(handles) != 64 * world:
            raise ValueError("handles must hold world x 64 bytes")
        buf = C.create_string_buffer(handles, len(handles))
        return int(self._lib.sfmba_p2p_attach(self._h, buf, int(rank), int(world)))

    def p2p_detach(self):
        self._check(self._lib.sfmba_p2p_detach(self._h))

    def problem_reuse(self):
        """(observations re-used from the previous problem of this handle, observations uploaded) of the last
        set_problem (include/sfmba.h: incremental re-use)."""
        a, b = C.c_int64(), C.c_int64()
        self._lib.sfmba_problem_reuse(self._h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def counters(self):
        """(kernel launches, collectives) enqueued by this handle so far."""
        a, b = C.c_int64(), C.c_int64()
        self._lib.sfmba_get_counters(self._h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def pcg_history(self):
        """PCG iterations of every outer iteration of the last solve on this handle."""
        n = int(self._lib.sfmba_get_pcg_history(self._h, None, 0))        # (returns the count, writes min(count, cap))
        buf = (C.c_int32 * max(n, 1))()
        n = min(n, int(self._lib.sfmba_get_pcg_history(self._h, buf, n)))
        return [int(buf[k]) for k in range(n)]

    def p2p_calls(self) -> int:
        return int(self._lib.sfmba_p2p_calls(self._h))

    # ------------------------------------------------------------------------------------------
    def residuals(self, x):
        self._flush_pending()
        x = _f64(x, (self.n_params,), "x")
        out = np.empty(2 * self.n_obs)
        self._check(self._lib.sfmba_residuals(self._h, _capi.ptr(x), _capi.ptr(out)))
        return out

    def residual_jacobian(self, x):
        self._flush_pending()
        x = _f64(x, (self.n_params,), "x")
        r = np.empty(2 * self.n_obs)
        Jc = np.empty((self.n_obs, 2, 6))
        Jp = np.empty((self.n_obs, 2, 3))
        self._check(self._lib.sfmba_residual_jacobian(self._h, _capi.ptr(x), _capi.ptr(r),
                                                      _capi.ptr(Jc), _capi.ptr(Jp)))
        return r, Jc, Jp

    def normal_blocks(self, x):
        self._flush_pending()
        x = _f64(x, (self.n_params,), "x")
        U = np.empty((self.n_cameras, 21))
        V = np.empty((self.n_points, 6))
        gc = np.empty((self.n_cameras, 6))
        gp = np.empty((self.n_points, 3))
        self._check(self._lib.sfmba_normal_blocks(self._h, _capi.ptr(x), _capi.ptr(U), _capi.ptr(V),
                                                  _capi.ptr(gc), _capi.ptr(gp)))
        return U, V, gc, gp

    def schur_matvec(self, x, dc, dp, v):
        self._flush_pending()
        x = _f64(x, (self.n_params,), "x")
        dc = _f64(dc).reshape(-1)
        dp = _f64(dp).reshape(-1)
        v = _f64(v).reshape(-1)
        y = np.empty(6 * self.n_cameras)
        self._check(self._lib.sfmba_schur_matvec(self._h, _capi.ptr(x), _capi.ptr(dc), _capi.ptr(dp),
                                                 _capi.ptr(v), _capi.ptr(y)))
        return y

    def dense_schur(self, x, dc, dp, rhs):
        """(S, y): the formed reduced camera matrix and the solution of S y = rhs by the in-LDS PCG run to the end."""
        self._flush_pending()
        x = _f64(x, (self.n_params,), "x")
        dc, dp, rhs = _f64(dc).reshape(-1), _f64(dp).reshape(-1), _f64(rhs).reshape(-1)
        n = 6 * self.n_cameras
        S = np.empty((n, n))
        y = np.empty(n)
        self._check(self._lib.sfmba_dense_schur(self._h, _capi.ptr(x), _capi.ptr(dc), _capi.ptr(dp), _capi.ptr(rhs),
                                                _capi.ptr(S), _capi.ptr(y)))
        return S, y

    def time_kernel(self, x, which: int, reps: int) -> float:
        self._flush_pending()
        x = _f64(x, (self.n_params,), "x")
        us = C.c_double()
        self._check(self._lib.sfmba_time_kernel(self._h, _capi.ptr(x), int(which), int(reps),
                                                C.byref(us)))
        return us.value

    def default_options(self) -> _capi.Options:
        o = _capi.Options()
        self._lib.sfmba_default_options(C.byref(o))
        return o

    def solve(self, x0, options: _capi.Options | None = None, want_fun=True, want_grad=True):
        """-> (x, result struct, fun, grad).  fun / grad not wanted stay on the device: `fetch_fun_grad` gets them
        until the next operation on this handle."""
        self._flush_pending()
        x_start = _f64(x0, (self.n_params,), "x0")
        x = np.empty_like(x_start)               # start and result in separate arrays: no copy of x0 on the way in
        opt = options if options is not None else self.default_options()
        res = _capi.Result()
        self._check(self._lib.sfmba_solve_from(self._h, _capi.ptr(x_start), _capi.ptr(x), C.byref(opt), C.byref(res)))
        fun, grad = self.fetch_fun_grad(want_fun, want_grad)
        return x, res, fun, grad
