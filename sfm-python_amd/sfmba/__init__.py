"""sfmba -- MI355X-native bundle adjustment behind the reference's least_squares call.

Public surface mirrors /root/reference/sfm_lite/bundle_adjustment.py and the scipy call of
/root/reference/sfm_lite/sfm.py:266-268.  Requires libsfmba.so (HIP, gfx950); no CPU fallback.
"""
from .api import (TERMINATION_MESSAGES, apply_bundle_adjustment, compute_residuals,
                  create_sparsity_matrix, get_backend, least_squares, pack_cameras_points,
                  project_points, unpack_cameras_points)
from .backend import Backend, BackendError
from .bal import read_bal, write_bal
from .extras import (calc_reproj_error, load_calibration_data, load_problem, reproj_error,
                     save_problem)
from .synthetic import (BAProblem, K_SCEAUX, drop_observations, growing_reconstruction, make_config,
                        make_problem, make_ring_problem)

__all__ = ["TERMINATION_MESSAGES", "apply_bundle_adjustment", "compute_residuals",
           "create_sparsity_matrix", "get_backend", "least_squares", "pack_cameras_points",
           "project_points", "unpack_cameras_points", "Backend", "BackendError", "BAProblem",
           "K_SCEAUX", "drop_observations", "growing_reconstruction", "make_config", "make_problem", "make_ring_problem",
           "calc_reproj_error", "load_calibration_data",
           "load_problem", "reproj_error", "save_problem", "read_bal", "write_bal"]
