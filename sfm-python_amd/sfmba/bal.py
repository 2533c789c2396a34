"""Reader / writer for the text format of the public "Bundle Adjustment in the Large" problems (SURVEY.md
section 8f-2: reproducible inputs beyond the synthetic generator).  No such file ships with the reference or this
image; the format is the published one:

    <num_cameras> <num_points> <num_observations>
    <camera_index> <point_index> <x> <y>                    (num_observations lines)
    <9 values per camera, one per line: Rodrigues vector (3), translation t (3), f, k1, k2>
    <3 values per point, one per line>

BAL's camera model is  P = R X + t,  p = -P / P.z,  pixel = f * (1 + k1 |p|^2 + k2 |p|^4) * p  with per-camera
intrinsics.  The model of this back end is the reference's (/root/reference/sfm_lite/bundle_adjustment.py:20-42):
one shared 3x3 K, no distortion, the translation slot holding the camera CENTRE.  The mapping used here is exact
for cameras without distortion that share their focal length:

    T = -R^T t (centre),   K = diag(-f, -f, 1)   (the minus signs carry BAL's "-P / P.z")

``read_bal`` reports what it had to approximate (spread of the focal lengths, largest |k1|, |k2|) instead of
hiding it; observations come back point-major, the order ``Graph.pt3ds_pt2ds`` produces
(/root/reference/sfm_lite/graph.py:186-191).
"""
from __future__ import annotations

import bz2
import gzip

import numpy as np

from .api import _matrix_from_rotvec


def _open(path, mode):
    p = str(path)
    if p.endswith(".bz2"):
        return bz2.open(p, mode + "t")
    if p.endswith(".gz"):
        return gzip.open(p, mode + "t")
    return open(p, mode)


def read_bal(path, focal="median"):
    """-> (x0, args, info): ``x0`` (6C + 3P,) and ``args`` = (n_cameras, n_points, camera_indices, point_indices,
    points_2d, K), the ``args=`` tuple of /root/reference/sfm_lite/sfm.py:268, ready for ``sfmba.least_squares``.
    ``focal``: "median" (of the per-camera focal lengths) or a number.  ``info`` holds the file's own per-camera
    intrinsics and how far they are from the shared-K, distortion-free model."""
    with _open(path, "r") as f:
        tokens = f.read().split()
    if len(tokens) < 3:
        raise ValueError("not a BAL file: header missing")
    C, P, N = (int(t) for t in tokens[:3])
    need = 3 + 4 * N + 9 * C + 3 * P
    if C <= 0 or P <= 0 or N <= 0 or len(tokens) < need:
        raise ValueError(f"not a BAL file: header says {C} cameras / {P} points / {N} observations = {need} values, "
                         f"found {len(tokens)}")
    obs = np.array(tokens[3:3 + 4 * N], dtype=np.float64).reshape(N, 4)
    ci = obs[:, 0].astype(np.int64)
    pi = obs[:, 1].astype(np.int64)
    if np.any(ci != obs[:, 0]) or np.any(pi != obs[:, 1]) or ci.min() < 0 or ci.max() >= C or pi.min() < 0 or pi.max() >= P:
        raise ValueError("BAL file: observation indices out of range")
    uv = obs[:, 2:4].copy()
    cams = np.array(tokens[3 + 4 * N:3 + 4 * N + 9 * C], dtype=np.float64).reshape(C, 9)
    pts = np.array(tokens[3 + 4 * N + 9 * C:need], dtype=np.float64).reshape(P, 3)
    order = np.argsort(pi, kind="stable")                       # point-major, stable within a point
    ci, pi, uv = ci[order], pi[order], uv[order]
    fs = cams[:, 6]
    f0 = float(np.median(fs)) if focal == "median" else float(focal)
    centres = np.stack([-_matrix_from_rotvec(c[:3]).T @ c[3:6] for c in cams])
    x0 = np.concatenate([np.hstack([cams[:, :3], centres]).ravel(), pts.ravel()])
    K = np.array([[-f0, 0.0, 0.0], [0.0, -f0, 0.0], [0.0, 0.0, 1.0]])
    info = dict(focal_used=f0, focal_min=float(fs.min()), focal_max=float(fs.max()),
                max_abs_k1=float(np.abs(cams[:, 7]).max()), max_abs_k2=float(np.abs(cams[:, 8]).max()),
                focal=fs.copy(), k1=cams[:, 7].copy(), k2=cams[:, 8].copy(), file_order=order,
                exact=bool(fs.min() == fs.max() and not cams[:, 7:].any()))
    return x0, (C, P, ci, pi, uv, K), info


def write_bal(path, x, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """Inverse mapping for problems of this back end whose K is ``diag(-f, -f, 1)`` or ``diag(f, f, 1)`` (the
    latter is written with negated pixels, BAL's sign convention): t = -R T, k1 = k2 = 0."""
    K = np.asarray(K, dtype=np.float64)
    off = K - np.diag(np.diag(K))
    if np.any(off != 0.0) or K[2, 2] != 1.0 or K[0, 0] != K[1, 1] or K[0, 0] == 0.0:
        raise ValueError("write_bal needs K = diag(+-f, +-f, 1): BAL has no principal point or skew")
    f0, sign = abs(K[0, 0]), (1.0 if K[0, 0] < 0 else -1.0)
    C, P = int(n_cameras), int(n_points)
    x = np.asarray(x, dtype=np.float64)
    cams = x[:6 * C].reshape(C, 6)
    pts = x[6 * C:].reshape(P, 3)
    uv = sign * np.asarray(points_2d, dtype=np.float64)
    with _open(path, "w") as f:
        f.write(f"{C} {P} {len(camera_indices)}\n")
        for c, p, (u, v) in zip(camera_indices, point_indices, uv):
            f.write(f"{int(c)} {int(p)} {float(u)!r} {float(v)!r}\n")
        for c in cams:
            t = -_matrix_from_rotvec(c[:3]) @ c[3:]
            for v in (*c[:3], *t, f0, 0.0, 0.0):
                f.write(f"{float(v)!r}\n")
        for p in pts:
            for v in p:
                f.write(f"{float(v)!r}\n")
