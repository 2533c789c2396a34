"""Helpers either side of the hot path: reprojection-error reporting in the pipeline's
``[R | t]`` convention, calibration / problem files.  Numeric work still goes through the HIP path."""
from __future__ import annotations

import numpy as np

from . import api


def reproj_error(point3ds, point2ds, K, R, tvec, device=0):
    """``(K (R X + t))`` projected minus ``point2ds`` -> (N,2), as
    /root/reference/cv2_lite/solve_pnp.py:9-15.  Evaluated by the BA residual kernel: ``R X + t`` equals
    the kernel's ``R (X - T)`` with the camera centre ``T = -R^T t``."""
    point3ds = np.ascontiguousarray(point3ds, dtype=np.float64)
    point2ds = np.ascontiguousarray(point2ds, dtype=np.float64)
    R = np.asarray(R, dtype=np.float64)
    tvec = np.asarray(tvec, dtype=np.float64).reshape(3)
    n = len(point3ds)
    if point3ds.shape != (n, 3) or point2ds.shape != (n, 2) or np.shape(K) != (3, 3):
        raise ValueError("expected point3ds (N,3), point2ds (N,2), K (3,3)")   # check_inputs.py:7-48
    cam = np.hstack([api._rotvec_from_matrix(R), -R.T @ tvec])
    x = np.concatenate([cam, point3ds.ravel()])
    r = api.compute_residuals(x, 1, n, np.zeros(n, dtype=np.int64), np.arange(n, dtype=np.int64), point2ds, K,
                              device=device)
    return r.reshape(n, 2)


def calc_reproj_error(points3d, points2d, K, R, tvec, device=0):
    """Mean L2 reprojection error, /root/reference/sfm_lite/sfm.py:38-41."""
    return float(np.linalg.norm(reproj_error(points3d, points2d, K, R, tvec, device=device), axis=1).mean())


def load_calibration_data(txt_path):
    """3x3 whitespace-separated text -> ndarray, /root/reference/sfm_lite/utils.py:24-35."""
    K = np.array([[float(v) for v in line.split()] for line in open(txt_path) if line.strip()])
    assert K.shape == (3, 3), K.shape
    return K


def save_problem(path, x0, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """The arguments of the reference's least_squares call as one .npz (no pickles)."""
    np.savez_compressed(path, x0=np.asarray(x0, dtype=np.float64), dims=np.array([n_cameras, n_points], dtype=np.int64),
                        camera_indices=np.asarray(camera_indices, dtype=np.int64),
                        point_indices=np.asarray(point_indices, dtype=np.int64), points_2d=np.asarray(points_2d),
                        K=np.asarray(K, dtype=np.float64))


def load_problem(path):
    """-> (x0, args) with args the tuple of sfm.py:268."""
    g = np.load(path, allow_pickle=False)
    C, P = (int(v) for v in g["dims"])
    return g["x0"], (C, P, g["camera_indices"], g["point_indices"], g["points_2d"], g["K"])
