"""ctypes binding of libsfmba.so (include/sfmba.h).  Thin by design: plain pointers and sizes.

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFMBA_LIB") or os.path.join(_HERE, "libsfmba.so")   # SFMBA_LIB: diagnostic builds

# every symbol include/sfmba.h declares (tests check the library exports each of them)
SYMBOLS = (
    "sfmba_create", "sfmba_destroy", "sfmba_last_error", "sfmba_default_options", "sfmba_set_stream",
    "sfmba_set_problem", "sfmba_set_problem_i64", "sfmba_exchange_doubles", "sfmba_set_exchange", "sfmba_residuals",
    "sfmba_residual_jacobian", "sfmba_solve", "sfmba_solve_from", "sfmba_get_fun_grad", "sfmba_time_kernel",
    "sfmba_normal_blocks", "sfmba_schur_matvec", "sfmba_comm_get_unique_id", "sfmba_comm_init",
    "sfmba_comm_destroy", "sfmba_set_precision", "sfmba_p2p_export", "sfmba_p2p_attach", "sfmba_p2p_detach",
    "sfmba_p2p_calls", "sfmba_tr2d_solve", "sfmba_debug_option", "sfmba_set_print", "sfmba_get_counters", "sfmba_problem_reuse", "sfmba_dense_schur",
    "sfmba_get_pcg_history", "sfmba_set_fixed_cameras",
)


class Options(C.Structure):
    _fields_ = [("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double),
                ("max_nfev", C.c_int64), ("verbose", C.c_int32), ("max_iter", C.c_int32),
                ("pcg_tol", C.c_double), ("pcg_max_iter", C.c_int32), ("pcg_check_every", C.c_int32),
                ("reg_min", C.c_double), ("profile", C.c_int32), ("reserved", C.c_int32),
                ("pcg_tol_max", C.c_double)]


class Result(C.Structure):
    _fields_ = [("cost", C.c_double), ("cost0", C.c_double), ("optimality", C.c_double),
                ("rmse", C.c_double), ("rmse0", C.c_double), ("nfev", C.c_int64), ("njev", C.c_int64),
                ("iterations", C.c_int64), ("pcg_iterations", C.c_int64), ("status", C.c_int32),
                ("reserved", C.c_int32), ("seconds_total", C.c_double), ("seconds_device", C.c_double),
                ("resjac_avg_us", C.c_double), ("resjac_launches", C.c_int64),
                ("last_step_norm", C.c_double), ("last_reg", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)
PRINT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  libsfmba.so needs `libamdhip64.so.7`; a PyTorch-ROCm wheel ships its own copy of that
    library (and of the HSA runtime) under torch/lib and loads it by path.  Whichever is loaded first wins the SONAME, and
    when the system's copy came first, `torch.cuda` later finds "No HIP GPUs" in this process (seen on this image: ROCm 7.2
    under /opt/rocm, 7.0 inside the wheel).  The package works with torch at its side (streams handed over by
    `set_stream`, `sfmba.dist`), so when torch is INSTALLED -- imported or not -- its copy is loaded first, and
    libsfmba.so binds to it exactly as it does when `import torch` came first.  SFMBA_HIP_RUNTIME=system skips this."""
    if os.environ.get("SFMBA_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                             # torch's libraries are in the process already
    try:
        spec = importlib.util.find_spec("torch")           # (locates the package without importing it)
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libsfmba.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP back end first (python __graft_entry__.py, or "
            "make -C sfm-python_amd).  sfmba has no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    lib.sfmba_create.argtypes = [C.POINTER(P), C.c_int]
    lib.sfmba_create.restype = C.c_int
    lib.sfmba_destroy.argtypes = [P]
    lib.sfmba_destroy.restype = None
    lib.sfmba_last_error.argtypes = [P]
    lib.sfmba_last_error.restype = C.c_char_p
    lib.sfmba_default_options.argtypes = [C.POINTER(Options)]
    lib.sfmba_default_options.restype = None
    lib.sfmba_set_stream.argtypes = [P, P]
    lib.sfmba_set_problem.argtypes = [P, C.c_int64, C.c_int64, C.c_int64, P, P, P, P]
    lib.sfmba_set_problem_i64.argtypes = [P, C.c_int64, C.c_int64, C.c_int64, P, P, P, P]
    lib.sfmba_exchange_doubles.argtypes = [C.c_int64]
    lib.sfmba_exchange_doubles.restype = C.c_int64
    lib.sfmba_set_exchange.argtypes = [P, P, C.c_int64, ALLREDUCE_FN, P, C.c_int64]
    lib.sfmba_residuals.argtypes = [P, P, P]
    lib.sfmba_residual_jacobian.argtypes = [P, P, P, P, P]
    lib.sfmba_solve.argtypes = [P, P, C.POINTER(Options), C.POINTER(Result)]
    lib.sfmba_solve_from.argtypes = [P, P, P, C.POINTER(Options), C.POINTER(Result)]
    lib.sfmba_get_fun_grad.argtypes = [P, P, P]
    lib.sfmba_time_kernel.argtypes = [P, P, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.sfmba_normal_blocks.argtypes = [P, P, P, P, P, P]
    lib.sfmba_schur_matvec.argtypes = [P, P, P, P, P, P]
    lib.sfmba_dense_schur.argtypes = [P, P, P, P, P, P, P]
    lib.sfmba_dense_schur.restype = C.c_int
    lib.sfmba_tr2d_solve.argtypes = [P, P, C.c_double, P]
    lib.sfmba_comm_get_unique_id.argtypes = [P]
    lib.sfmba_comm_init.argtypes = [P, P, C.c_int32, C.c_int32, C.c_int64]
    lib.sfmba_comm_destroy.argtypes = [P]
    lib.sfmba_set_precision.argtypes = [P, C.c_int32]
    lib.sfmba_p2p_export.argtypes = [P, C.c_int32, P]
    lib.sfmba_p2p_attach.argtypes = [P, P, C.c_int32, C.c_int32]
    lib.sfmba_p2p_detach.argtypes = [P]
    lib.sfmba_debug_option.argtypes = [P, C.c_char_p, C.c_int64]
    lib.sfmba_problem_reuse.argtypes = [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.sfmba_problem_reuse.restype = C.c_int
    lib.sfmba_get_counters.argtypes = [P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.sfmba_get_counters.restype = C.c_int
    lib.sfmba_get_pcg_history.argtypes = [P, P, C.c_int32]
    lib.sfmba_get_pcg_history.restype = C.c_int32
    lib.sfmba_set_print.argtypes = [P, PRINT_FN, P]
    lib.sfmba_set_print.restype = C.c_int
    lib.sfmba_set_fixed_cameras.argtypes = [P, P, C.c_int64]
    lib.sfmba_set_fixed_cameras.restype = C.c_int
    lib.sfmba_p2p_calls.argtypes = [P]
    lib.sfmba_p2p_calls.restype = C.c_int64
    for name in ("sfmba_set_stream", "sfmba_set_problem", "sfmba_set_problem_i64", "sfmba_set_exchange", "sfmba_residuals",
                 "sfmba_residual_jacobian", "sfmba_solve", "sfmba_solve_from", "sfmba_get_fun_grad", "sfmba_time_kernel",
                 "sfmba_normal_blocks", "sfmba_schur_matvec", "sfmba_tr2d_solve", "sfmba_comm_get_unique_id",
                 "sfmba_comm_init", "sfmba_comm_destroy", "sfmba_set_precision", "sfmba_p2p_export",
                 "sfmba_p2p_attach", "sfmba_p2p_detach", "sfmba_debug_option"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)
