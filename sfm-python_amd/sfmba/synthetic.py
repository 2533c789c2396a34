"""Synthetic bundle-adjustment problems (SURVEY.md §8d generator).

The reference ships no BA inputs (its SceauxCastle submodule is an empty directory), so every
configuration of BASELINE.json is represented by this seeded generator.  It reproduces the *layout*
the reference's caller hands to ``least_squares`` (``/root/reference/sfm_lite/sfm.py:248-262``):

* observations in point-major order, as ``Graph.pt3ds_pt2ds`` yields them
  (``/root/reference/sfm_lite/graph.py:186-191``);
* integer pixel observations truncated toward zero (``graph.py:112-113``), dtype int64;
* ``x0 = [C x (rotvec, T) | P x (X, Y, Z)]`` float64 (``sfm.py:257``);
* the residual model ``pi(K R(w) (X - T)) - uv`` of
  ``/root/reference/sfm_lite/bundle_adjustment.py:20-42``.

Nothing here touches the GPU or the oracle: the truth projection is computed with a small local
vectorised Rodrigues so that the generator has no dependency on test infrastructure.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# SceauxCastle intrinsics, the only literal the reference holds for that dataset
# (/root/reference/cv2_lite/solve_pnp.py:81-83, recover_pose.py:96-98).
K_SCEAUX = np.array([[2905.88, 0.0, 1416.0],
                     [0.0, 2905.88, 1064.0],
                     [0.0, 0.0, 1.0]], dtype=np.float64)

# BASELINE.json configs (C, P, N); cfg1 and cfg2 share the SceauxCastle-scale size.
CONFIGS = {
    "cfg2": (11, 3000, 10000),
    "cfg3": (200, 20000, 200000),
    "cfg4": (1000, 100000, 1000000),
    "cfg5": (5000, 1000000, 10000000),
}


@dataclasses.dataclass
class BAProblem:
    """Arguments of the reference's least_squares call, plus the generating truth."""
    n_cameras: int
    n_points: int
    camera_indices: np.ndarray   # (N,) int64
    point_indices: np.ndarray    # (N,) int64, non-decreasing (point-major)
    points_2d: np.ndarray        # (N, 2) int64 pixels
    K: np.ndarray                # (3, 3) float64
    x0: np.ndarray               # (6C + 3P,) float64
    x_true: np.ndarray           # (6C + 3P,) float64

    @property
    def n_obs(self) -> int:
        return int(self.camera_indices.shape[0])

    @property
    def args(self):
        """The ``args=`` tuple of sfm.py:268."""
        return (self.n_cameras, self.n_points, self.camera_indices, self.point_indices,
                self.points_2d, self.K)


def _rodrigues_batch(w: np.ndarray) -> np.ndarray:
    """(C,3) rotation vectors -> (C,3,3) matrices (closed-form Rodrigues, series near 0)."""
    th2 = np.einsum("ij,ij->i", w, w)
    th = np.sqrt(th2)
    small = th < 1e-4
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th2 / 6.0, np.sin(ths) / ths)
    half = 0.5 * ths
    b = np.where(small, 0.5 - th2 / 24.0, 0.5 * (np.sin(half) / half) ** 2)
    Wx = np.zeros((w.shape[0], 3, 3))
    Wx[:, 0, 1], Wx[:, 0, 2] = -w[:, 2], w[:, 1]
    Wx[:, 1, 0], Wx[:, 1, 2] = w[:, 2], -w[:, 0]
    Wx[:, 2, 0], Wx[:, 2, 1] = -w[:, 1], w[:, 0]
    return np.eye(3)[None] + a[:, None, None] * Wx + b[:, None, None] * (Wx @ Wx)


def make_problem(n_cameras: int, n_points: int, n_obs: int, seed: int = 0,
                 K: np.ndarray | None = None, pixel_noise: float = 0.5,
                 x0_noise: float = 0.01, point_offset: int = 0,
                 camera_seed: int | None = None) -> BAProblem:
    """SURVEY.md §8d generator.

    ``camera_seed`` (default: ``seed``) seeds the cameras separately so that several ranks can
    generate *different* points/observations of ONE problem that shares its cameras
    (multi-GPU weak scaling: each rank calls this with its own ``seed`` and a common
    ``camera_seed``).  ``point_offset`` is unused by the arithmetic and only recorded by callers.
    """
    if n_obs < n_points:
        raise ValueError("every point needs at least one observation (n_obs >= n_points)")
    K = K_SCEAUX.copy() if K is None else np.asarray(K, dtype=np.float64)
    rng = np.random.default_rng(seed)
    crng = rng if camera_seed is None else np.random.default_rng(camera_seed)
    cam_w = crng.normal(0.0, 0.1, (n_cameras, 3))
    cam_T = crng.normal(0.0, 0.5, (n_cameras, 3))
    pts = rng.normal(0.0, 1.0, (n_points, 3)) + np.array([0.0, 0.0, 10.0])
    pt_idx = np.concatenate([np.arange(n_points, dtype=np.int64),
                             rng.integers(0, n_points, n_obs - n_points, dtype=np.int64)])
    cam_idx = rng.integers(0, n_cameras, n_obs, dtype=np.int64)
    order = np.argsort(pt_idx, kind="stable")
    pt_idx, cam_idx = pt_idx[order], cam_idx[order]

    R = _rodrigues_batch(cam_w)
    v = pts[pt_idx] - cam_T[cam_idx]
    q = np.einsum("nij,nj->ni", R[cam_idx], v)
    p = q @ K.T
    proj = p[:, :2] / p[:, 2:3]
    uv = np.trunc(proj + rng.normal(0.0, pixel_noise, proj.shape)).astype(np.int64)

    x_true = np.concatenate([np.hstack([cam_w, cam_T]).ravel(), pts.ravel()])
    if camera_seed is None:
        x0 = x_true + rng.normal(0.0, x0_noise, x_true.shape)
    else:
        x0 = x_true.copy()
        x0[:6 * n_cameras] += np.random.default_rng(camera_seed + 7919).normal(
            0.0, x0_noise, 6 * n_cameras)
        x0[6 * n_cameras:] += rng.normal(0.0, x0_noise, 3 * n_points)
    return BAProblem(n_cameras, n_points, cam_idx, pt_idx, uv, K, x0, x_true)


def make_config(name: str, seed: int = 0) -> BAProblem:
    C, P, N = CONFIGS[name]
    return make_problem(C, P, N, seed=seed)


def drop_observations(pb: BAProblem, cameras=(), points=()) -> BAProblem:
    """Same cameras and points, minus every observation of the given cameras / points (they stay in
    ``x`` as unobserved parameters: zero Jacobian columns, which scipy's ``x_scale='jac'`` maps to scale 1)."""
    keep = ~np.isin(pb.camera_indices, np.asarray(list(cameras), dtype=np.int64)) & \
           ~np.isin(pb.point_indices, np.asarray(list(points), dtype=np.int64))
    return BAProblem(pb.n_cameras, pb.n_points, pb.camera_indices[keep], pb.point_indices[keep],
                     pb.points_2d[keep], pb.K, pb.x0, pb.x_true)


def _rotvec_from_matrix(R: np.ndarray) -> np.ndarray:
    """Log map for generic rotations (angle away from 0 and pi handled; used by the ring scene only)."""
    tr = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(tr)
    if th < 1e-12:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if np.pi - th < 1e-6:                      # near pi: take the axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        k = int(np.argmax(np.diag(A)))
        v = A[:, k] / np.sqrt(A[k, k])
        return th * v * (1.0 if ax @ v >= 0 else -1.0)
    return th * ax / (2.0 * np.sin(th))


def make_ring_problem(n_cameras: int, n_points: int, n_obs: int, seed: int = 0, radius: float = 8.0,
                      pixel_noise: float = 0.5, x0_noise: float = 0.01) -> BAProblem:
    """Cameras on a circle around the point cloud, all looking at its centre: rotation vectors of every
    magnitude up to pi (the first camera sits at exactly 180 degrees), depths from radius-3 to radius+3.
    Same layout conventions as :func:`make_problem`."""
    rng = np.random.default_rng(seed)
    K = K_SCEAUX.copy()
    ang = np.pi + 2.0 * np.pi * np.arange(n_cameras) / n_cameras
    cam_T = np.stack([radius * np.sin(ang), 0.3 * rng.normal(size=n_cameras), -radius * np.cos(ang)], axis=1)
    cam_w = np.empty((n_cameras, 3))
    for c in range(n_cameras):
        z = -cam_T[c] / np.linalg.norm(cam_T[c])            # optical axis towards the origin
        x = np.cross([0.0, 1.0, 0.0], z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        cam_w[c] = _rotvec_from_matrix(np.stack([x, y, z]))  # rows: camera axes in world coordinates
    pts = rng.normal(0.0, 1.0, (n_points, 3))
    pt_idx = np.concatenate([np.arange(n_points, dtype=np.int64),
                             rng.integers(0, n_points, n_obs - n_points, dtype=np.int64)])
    cam_idx = rng.integers(0, n_cameras, n_obs, dtype=np.int64)
    order = np.argsort(pt_idx, kind="stable")
    pt_idx, cam_idx = pt_idx[order], cam_idx[order]
    R = _rodrigues_batch(cam_w)
    q = np.einsum("nij,nj->ni", R[cam_idx], pts[pt_idx] - cam_T[cam_idx])
    p = q @ K.T
    uv = np.trunc(p[:, :2] / p[:, 2:3] + rng.normal(0.0, pixel_noise, (n_obs, 2))).astype(np.int64)
    x_true = np.concatenate([np.hstack([cam_w, cam_T]).ravel(), pts.ravel()])
    x0 = x_true + rng.normal(0.0, x0_noise, x_true.shape)
    return BAProblem(n_cameras, n_points, cam_idx, pt_idx, uv, K, x0, x_true)


def make_plane_crossing_problem(seed: int = 0, quanta: int = 1, n_cameras: int = 5, n_points: int = 60, n_obs: int = 360,
                                x0_noise: float = 0.05) -> BAProblem:
    """A problem whose FIRST trial step puts a point exactly on a camera's principal plane (p_z = 0), so that the
    reference's unguarded division (``bundle_adjustment.py:30``) yields a non-finite residual and the solver has to take
    scipy's branch for it (``trf.py:504-506``: radius <- 1/4 |step|, retry).

    Crossing the plane is not enough -- floating point would step over it -- so the scene is built to LAND on it, in a
    way that does not depend on the last bits of the step: every z coordinate is shifted by 2^44, where doubles are
    spaced 2^-8 (3.9 mm) apart, so z moves in whole quanta.  Camera 0 (to be held still through ``jac_sparsity`` /
    ``fixed_camera_indices``, rotation vector exactly 0, so p_z = Z - T_z exactly) stands 8 units in front of the others,
    and point 0 sits ``quanta`` x 3.9 mm in front of it, seen by it and by three more cameras from 8 units away.  From the
    far start the first step moves that point back by about its depth, which rounds onto the plane.  Returns the problem
    with ``x0`` (camera 0 at its true pose); hold camera 0 fixed when solving."""
    rng = np.random.default_rng(seed)
    K = K_SCEAUX.copy()
    q = 2.0 ** -8
    z0 = 2.0 ** 44
    C, P, N = n_cameras, n_points, n_obs
    cam_w = rng.normal(0.0, 0.1, (C, 3))
    cam_T = rng.normal(0.0, 0.5, (C, 3))
    cam_w[0] = 0.0
    cam_T[0] = (0.0, 0.0, 8.0)
    pts = rng.normal(0.0, 1.0, (P, 3)) + np.array([0.0, 0.0, 10.0])
    pts[:, 2] = np.maximum(pts[:, 2], 8.6)                     # everything else well in front of camera 0
    pts[0] = (1e-3 * rng.normal(), 1e-3 * rng.normal(), 8.0 + quanta * q)
    pt_idx = np.concatenate([np.arange(P), rng.integers(0, P, N - P)])
    cam_idx = rng.integers(0, C, N)
    sel = pt_idx == 0
    cam_idx[sel] = np.where(cam_idx[sel] == 0, 3, cam_idx[sel])
    pt_idx = np.concatenate([pt_idx, np.zeros(3, dtype=np.int64)])
    cam_idx = np.concatenate([cam_idx, np.array([0, 1, 2])])
    order = np.argsort(pt_idx, kind="stable")
    pt_idx, cam_idx = pt_idx[order].astype(np.int64), cam_idx[order].astype(np.int64)
    cam_T[:, 2] += z0
    pts[:, 2] += z0
    R = _rodrigues_batch(cam_w)
    qq = np.einsum("nij,nj->ni", R[cam_idx], pts[pt_idx] - cam_T[cam_idx])
    p = qq @ K.T
    uv = np.trunc(p[:, :2] / p[:, 2:3] + rng.normal(0.0, 0.5, (len(pt_idx), 2))).astype(np.int64)
    x_true = np.concatenate([np.hstack([cam_w, cam_T]).ravel(), pts.ravel()])
    x0 = x_true + rng.normal(0.0, x0_noise, x_true.shape)
    x0[:6] = x_true[:6]
    x0[6 * C:6 * C + 3] = x_true[6 * C:6 * C + 3] + np.array([1e-3 * rng.normal(), 1e-3 * rng.normal(), 0.0])
    return BAProblem(C, P, cam_idx, pt_idx, uv, K, x0, x_true)


def growing_reconstruction(pb: BAProblem, order=None, first: int = 2):
    """The sequence of bundle-adjustment inputs an incremental reconstruction of ``pb`` produces when BA runs
    after every newly registered camera, the way ``SFM.construct`` grows its graph
    (``/root/reference/sfm_lite/sfm.py:59-71``: initial two-view registration, then one camera per fused edge).

    Yields one ``dict`` per stage with

    * ``registered``: list of C flags (node k registered?), ``new_camera``: node ids registered at this stage;
    * ``cloud``: original point ids in creation order (a point enters the cloud at the first stage where two
      registered cameras see it -- triangulation needs two views -- and keeps its cloud index afterwards,
      ``graph.py:101-119``); ``n_new_points``: how many were appended at this stage;
    * ``observations``: ``(cloud_index, node_id, (x, y))`` tuples in ``Graph.pt3ds_pt2ds`` order
      (``graph.py:186-191``): point-major, observations of unregistered nodes skipped.

    Structure only: poses and coordinates are the caller's state (they change with every BA result).
    """
    C = pb.n_cameras
    order = list(range(C)) if order is None else list(order)
    assert sorted(order) == list(range(C)) and 2 <= first <= C
    ci, pi, uv = pb.camera_indices, pb.point_indices, pb.points_2d
    in_cloud = np.zeros(pb.n_points, dtype=bool)
    cloud: list = []
    registered = np.zeros(C, dtype=bool)
    for k in range(first, C + 1):
        new_cams = [c for c in order[:k] if not registered[c]]
        registered[order[:k]] = True
        seen = registered[ci]
        views = np.bincount(pi[seen], minlength=pb.n_points)
        fresh = np.flatnonzero((views >= 2) & ~in_cloud)
        in_cloud[fresh] = True
        cloud.extend(int(p) for p in fresh)
        cloud_index = -np.ones(pb.n_points, dtype=np.int64)
        cloud_index[np.asarray(cloud, dtype=np.int64)] = np.arange(len(cloud))
        keep = np.flatnonzero(seen & in_cloud[pi])
        keep = keep[np.argsort(cloud_index[pi[keep]], kind="stable")]
        obs = [(int(cloud_index[pi[i]]), int(ci[i]), (int(uv[i, 0]), int(uv[i, 1]))) for i in keep]
        yield dict(registered=[bool(f) for f in registered], new_camera=new_cams, cloud=list(cloud),
                   n_new_points=len(fresh), observations=obs)
