"""CPU ORACLE for the bundle-adjustment hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product (``sfm-python_amd/``) never does: it fails loudly when ``libsfmba.so`` (HIP) is
missing.

What is restated here, and from where (REF = /root/reference, SCIPY = the scipy 1.15.3 of this image;
the reference pins scipy~=1.13.0, REF/requirements.txt:7 -- scipy is a third-party dependency that is
not under /root/reference, so its published TRF algorithm is restated from its call site
REF/sfm_lite/sfm.py:266-268 and from the installed source):

* ``project_points_loop`` / ``compute_residuals_loop``  -- REF/sfm_lite/bundle_adjustment.py:20-42,
  per-observation Python loop (same cost profile as the reference: the "faithful" CPU baseline B1).
* ``compute_residuals``            -- the same arithmetic vectorised with NumPy (baseline B2).
* ``create_sparsity_pattern``      -- REF/sfm_lite/bundle_adjustment.py:6-17 (CSR indices of the 0/1
  pattern, incl. ``fixed_camera_indices``).
* ``rodrigues``                    -- scipy Rotation.from_rotvec(w).as_matrix()
  (REF/sfm_lite/bundle_adjustment.py:25), closed form.
* ``jacobian_blocks``              -- analytic 2x6 / 2x3 blocks of the residual above; replaces scipy's
  sparse 2-point finite differences (SCIPY/optimize/_numdiff.py:628-705).
* ``trf_schur``                    -- SCIPY/optimize/_lsq/trf.py:401-560 (``trf_no_bounds``,
  tr_solver='lsmr', x_scale='jac') with the LSMR call of trf.py:480 replaced by the exact solution
  of the same damped normal equations through the Schur complement on the cameras.
* ``pack_problem`` / ``unpack_result`` -- REF/sfm_lite/sfm.py:248-262 and 271-281.

Parity pinning: the reference holds no tests or golden vectors for this path, so the
oracle is pinned by fixtures captured from the reference itself, imported by file path in the build
container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``) and by the recorded scipy run on the
SceauxCastle-scale synthetic (``tests/golden/scipy_cfg2_run.json``).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

# --------------------------------------------------------------------------------------------
# rotation
# --------------------------------------------------------------------------------------------


def rodrigues(w: np.ndarray) -> np.ndarray:
    """(…,3) rotation vectors -> (…,3,3).  R = I + a [w]x + b [w]x^2.

    a = sin(t)/t, b = (1-cos t)/t^2 = 0.5 (sin(t/2)/(t/2))^2 (no cancellation); series below 1e-4.
    Agrees with scipy's quaternion route (Taylor at t<=1e-3) to <=6e-16 (tests/test_oracle.py).
    """
    w = np.asarray(w, dtype=np.float64)
    th2 = np.sum(w * w, axis=-1)
    th = np.sqrt(th2)
    small = th < 1e-4
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th2 / 6.0 + th2 * th2 / 120.0, np.sin(ths) / ths)
    half = 0.5 * ths
    b = np.where(small, 0.5 - th2 / 24.0 + th2 * th2 / 720.0, 0.5 * (np.sin(half) / half) ** 2)
    Wx = skew(w)
    eye = np.broadcast_to(np.eye(3), Wx.shape)
    return eye + a[..., None, None] * Wx + b[..., None, None] * (Wx @ Wx)


def skew(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros(v.shape[:-1] + (3, 3))
    out[..., 0, 1], out[..., 0, 2] = -v[..., 2], v[..., 1]
    out[..., 1, 0], out[..., 1, 2] = v[..., 2], -v[..., 0]
    out[..., 2, 0], out[..., 2, 1] = -v[..., 1], v[..., 0]
    return out


def so3_bc(w: np.ndarray):
    """Coefficients of the right Jacobian Jr(w) = I - b [w]x + c [w]x^2.

    b = (1-cos t)/t^2, c = (t - sin t)/t^3; c by series below t=0.3 (the closed form cancels).
    """
    th2 = np.sum(np.asarray(w, dtype=np.float64) ** 2, axis=-1)
    th = np.sqrt(th2)
    tiny = th < 1e-4
    ths = np.where(tiny, 1.0, th)
    half = 0.5 * ths
    b = np.where(tiny, 0.5 - th2 / 24.0 + th2 * th2 / 720.0, 0.5 * (np.sin(half) / half) ** 2)
    c_series = (1.0 / 6.0 - th2 / 120.0 + th2 ** 2 / 5040.0 - th2 ** 3 / 362880.0
                + th2 ** 4 / 39916800.0 - th2 ** 5 / 6227020800.0)
    c_closed = (ths - np.sin(ths)) / (ths ** 3)
    c = np.where(th < 0.3, c_series, c_closed)
    return b, c


def rotvec_from_matrix(R: np.ndarray) -> np.ndarray:
    """Rotation.from_matrix(R).as_rotvec() (REF sfm.py:255) for a proper rotation, via quaternion."""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        qw, qx, qy, qz = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        qw, qx, qy, qz = (R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        qw, qx, qy, qz = (R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        qw, qx, qy, qz = (R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s
    if qw < 0:
        qw, qx, qy, qz = -qw, -qx, -qy, -qz
    n = math.sqrt(qx * qx + qy * qy + qz * qz)
    ang = 2.0 * math.atan2(n, qw)
    if n < 1e-12:
        return 2.0 * np.array([qx, qy, qz])
    return np.array([qx, qy, qz]) * (ang / n)


# --------------------------------------------------------------------------------------------
# residual (REF bundle_adjustment.py:20-42)
# --------------------------------------------------------------------------------------------


def project_points_loop(points, camera_params, K):
    """Per-observation restatement of REF bundle_adjustment.py:20-32 (deliberately a Python loop:
    the rotation matrix is rebuilt per observation exactly as the reference does)."""
    out = np.empty((len(points), 2))
    for i in range(len(points)):
        prm = camera_params[i]
        R = rodrigues(prm[:3])
        T = prm[3:]
        M = K @ np.hstack((R, -(R @ T)[:, None]))          # 3x4, bundle_adjustment.py:27
        p = M @ np.append(points[i], 1.0)                  # :29
        out[i, 0] = p[0] / p[2]                            # :30
        out[i, 1] = p[1] / p[2]
    return out


def compute_residuals_loop(x, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """REF bundle_adjustment.py:35-42 with the per-observation loop (baseline B1)."""
    cams = x[:n_cameras * 6].reshape((n_cameras, 6))
    pts = x[n_cameras * 6:].reshape((n_points, 3))
    proj = project_points_loop(pts[point_indices], cams[camera_indices], K)
    return (proj - points_2d).ravel()


def compute_residuals(x, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """Vectorised restatement: r_i = pi(K R(w_c) (X_p - T_c)) - uv_i, interleaved (2N,)."""
    x = np.asarray(x, dtype=np.float64)
    cams = x[:n_cameras * 6].reshape((n_cameras, 6))
    pts = x[n_cameras * 6:].reshape((n_points, 3))
    R = rodrigues(cams[:, :3])
    v = pts[point_indices] - cams[camera_indices, 3:]
    q = np.einsum("nij,nj->ni", R[camera_indices], v)
    p = q @ np.asarray(K, dtype=np.float64).T
    with np.errstate(divide="ignore", invalid="ignore"):
        proj = p[:, :2] / p[:, 2:3]
    return (proj - points_2d).ravel()


def create_sparsity_pattern(n_cameras, n_points, n_obs, camera_indices, point3d_indices,
                            fixed_camera_indices=()):
    """CSR (indptr, indices) of the 0/1 matrix of REF bundle_adjustment.py:6-17, shape
    (2 n_obs, 6 n_cameras + 3 n_points); rows 2i, 2i+1 carry camera columns 6c..6c+5 (unless c is
    fixed) and point columns 6C+3p..6C+3p+2, column indices ascending inside a row."""
    camera_indices = np.asarray(camera_indices)
    point3d_indices = np.asarray(point3d_indices)
    assert len(camera_indices) == len(point3d_indices)
    fixed = np.isin(camera_indices, np.asarray(list(fixed_camera_indices), dtype=np.int64))
    indptr = [0]
    indices = []
    for i in range(n_obs):
        cols = []
        if not fixed[i]:
            cols.extend(range(int(camera_indices[i]) * 6, int(camera_indices[i]) * 6 + 6))
        base = n_cameras * 6 + int(point3d_indices[i]) * 3
        cols.extend(range(base, base + 3))
        for _ in range(2):
            indices.extend(cols)
            indptr.append(len(indices))
    return np.asarray(indptr, dtype=np.int64), np.asarray(indices, dtype=np.int64)


# --------------------------------------------------------------------------------------------
# analytic Jacobian blocks
# --------------------------------------------------------------------------------------------


def jacobian_blocks(x, n_cameras, n_points, camera_indices, point_indices, points_2d, K):
    """Returns (r (N,2), Jc (N,2,6), Jp (N,2,3)).

    v = X-T, q = R v, p = K q, A = dpi/dp K (2x3);  dr/dX = A R;  dr/dT = -A R;
    dr/dw = -A R [v]x Jr(w) with Jr = I - b [w]x + c [w]x^2 (right Jacobian of SO(3)):
    row k of dr/dw = -(m - b (m x w) + c ((m x w) x w)),  m = (row k of A R) x v.
    """
    x = np.asarray(x, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    cams = x[:n_cameras * 6].reshape((n_cameras, 6))
    pts = x[n_cameras * 6:].reshape((n_points, 3))
    w = cams[:, :3]
    R = rodrigues(w)
    b, c = so3_bc(w)
    ci, pi = np.asarray(camera_indices), np.asarray(point_indices)
    v = pts[pi] - cams[ci, 3:]
    Rn = R[ci]
    q = np.einsum("nij,nj->ni", Rn, v)
    p = q @ K.T
    iz = 1.0 / p[:, 2]
    proj = p[:, :2] * iz[:, None]
    r = proj - np.asarray(points_2d, dtype=np.float64)
    # A[k, j] = (K[k, j] - proj_k K[2, j]) / p_z
    A = (K[None, :2, :] - proj[:, :, None] * K[None, 2:3, :]) * iz[:, None, None]
    Jp = A @ Rn                                              # (N,2,3)
    wn = w[ci]
    m = np.cross(Jp, v[:, None, :])                          # rows a_k x v
    s = np.cross(m, wn[:, None, :])
    t = np.cross(s, wn[:, None, :])
    Jw = -(m - b[ci][:, None, None] * s + c[ci][:, None, None] * t)
    Jc = np.concatenate([Jw, -Jp], axis=2)                   # (N,2,6): [d/dw | d/dT]
    return r, Jc, Jp


def jacobian_csr(Jc, Jp, n_cameras, n_points, camera_indices, point_indices):
    """Assemble the blocks into the scipy CSR layout of the pattern above (test helper)."""
    import scipy.sparse as sp
    N = Jc.shape[0]
    ci, pi = np.asarray(camera_indices), np.asarray(point_indices)
    cols_c = ci[:, None] * 6 + np.arange(6)[None, :]
    cols_p = n_cameras * 6 + pi[:, None] * 3 + np.arange(3)[None, :]
    cols = np.concatenate([cols_c, cols_p], axis=1)          # (N,9)
    cols = np.repeat(cols[:, None, :], 2, axis=1).reshape(-1)
    vals = np.concatenate([Jc, Jp], axis=2).reshape(-1)      # (N,2,9)
    indptr = np.arange(0, 18 * N + 1, 9)
    return sp.csr_matrix((vals, cols, indptr), shape=(2 * N, 6 * n_cameras + 3 * n_points))


# --------------------------------------------------------------------------------------------
# normal-equation blocks, Schur complement
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass
class NormalBlocks:
    U: np.ndarray    # (C,6,6)  sum Jc^T Jc
    V: np.ndarray    # (P,3,3)  sum Jp^T Jp
    W: np.ndarray    # (N,6,3)  Jc^T Jp per observation
    gc: np.ndarray   # (C,6)    sum Jc^T r
    gp: np.ndarray   # (P,3)    sum Jp^T r


def normal_blocks(r, Jc, Jp, n_cameras, n_points, camera_indices, point_indices) -> NormalBlocks:
    U = np.zeros((n_cameras, 6, 6))
    V = np.zeros((n_points, 3, 3))
    gc = np.zeros((n_cameras, 6))
    gp = np.zeros((n_points, 3))
    np.add.at(U, camera_indices, np.einsum("nki,nkj->nij", Jc, Jc))
    np.add.at(V, point_indices, np.einsum("nki,nkj->nij", Jp, Jp))
    np.add.at(gc, camera_indices, np.einsum("nki,nk->ni", Jc, r))
    np.add.at(gp, point_indices, np.einsum("nki,nk->ni", Jp, r))
    W = np.einsum("nki,nkj->nij", Jc, Jp)
    return NormalBlocks(U, V, W, gc, gp)


class NoComm:
    """Single-process stand-in for the collectives of an observation-sharded solve."""
    def sum(self, a):
        return a

    def max(self, a):
        return a


def schur_solve(nb: NormalBlocks, Dc, Dp, camera_indices, point_indices, rhs_c, rhs_p,
                method="dense", pcg_tol=1e-10, pcg_maxiter=None, precond="block_u", info=None,
                comm=None):
    """Solve [[U+Dc, W],[W^T, V+Dp]] [dc; dp] = [rhs_c; rhs_p] by eliminating the points.

    Dc (C,6), Dp (P,3) are the diagonal damping terms.  ``method='dense'`` forms S explicitly and
    uses a dense solve; ``method='pcg'`` runs block-Jacobi preconditioned CG on the implicit S (the
    algorithm of the HIP path).  With ``comm`` the observations are one shard of a larger problem: ``nb.U``
    and ``rhs_c`` are already summed over shards, every sum over observations that lands on cameras
    goes through ``comm.sum`` (exactly the all-reduces of the HIP path), points stay local.
    """
    comm = comm or NoComm()
    C, P = nb.U.shape[0], nb.V.shape[0]
    ci, pi = np.asarray(camera_indices), np.asarray(point_indices)
    Ud = nb.U.copy()
    Ud[:, np.arange(6), np.arange(6)] += Dc
    Vd = nb.V.copy()
    Vd[:, np.arange(3), np.arange(3)] += Dp
    Vinv = np.linalg.inv(Vd)
    # reduced rhs: rhs_c - sum_i W_i Vinv_p rhs_p
    t = np.einsum("pij,pj->pi", Vinv, rhs_p)
    acc = np.zeros_like(rhs_c)
    np.add.at(acc, ci, -np.einsum("nij,nj->ni", nb.W, t[pi]))
    red = rhs_c + comm.sum(acc)

    def S_mv(vc):
        vc = vc.reshape(C, 6)
        y = np.zeros((P, 3))
        np.add.at(y, pi, np.einsum("nij,ni->nj", nb.W, vc[ci]))
        z = np.einsum("pij,pj->pi", Vinv, y)
        acc = np.zeros((C, 6))
        np.add.at(acc, ci, -np.einsum("nij,nj->ni", nb.W, z[pi]))
        return (np.einsum("cij,cj->ci", Ud, vc) + comm.sum(acc)).reshape(-1)

    if method == "dense":
        S = np.zeros((C, 6, C, 6))
        S[np.arange(C), :, np.arange(C), :] = Ud
        WV = np.einsum("nij,njk->nik", nb.W, Vinv[pi])       # (N,6,3)
        # pairs of observations sharing a point
        order = np.argsort(pi, kind="stable")
        starts = np.flatnonzero(np.r_[True, pi[order][1:] != pi[order][:-1]])
        ends = np.r_[starts[1:], len(order)]
        for s, e in zip(starts, ends):
            idx = order[s:e]
            blk = np.einsum("aik,bjk->aibj", WV[idx], nb.W[idx])   # (k,6,k,6)
            for a, ia in enumerate(idx):
                for bq, ib in enumerate(idx):
                    S[ci[ia], :, ci[ib], :] -= blk[a, :, bq, :]
        S = S.reshape(6 * C, 6 * C)
        dc = np.linalg.solve(S, red.reshape(-1)).reshape(C, 6)
    else:
        if precond == "block_u":
            Minv = np.linalg.inv(Ud)
        elif precond == "schur":  # diagonal blocks of S, one term per observation (what the camera-major pass of
            Sd = Ud.copy()        # the HIP path sums along with the reduced right-hand side)
            WV = np.einsum("nij,njk->nik", nb.W, Vinv[pi])
            np.add.at(Sd, ci, -np.einsum("nik,njk->nij", WV, nb.W))
            Minv = np.linalg.inv(Sd)
        elif precond == "schur_exact":  # the diagonal blocks of the FORMED S (few-camera path of the HIP code): the
            # same, except that several observations of one (camera, point) pair -- random test problems have
            # them, a reconstruction does not -- contribute (sum W)(Vinv)(sum W)^T, cross terms included
            key = ci.astype(np.int64) * P + pi
            uniq, inv = np.unique(key, return_inverse=True)
            Wsum = np.zeros((len(uniq), 6, 3))
            np.add.at(Wsum, inv, nb.W)
            cu, pu = uniq // P, uniq % P
            Sd = Ud.copy()
            np.add.at(Sd, cu, -np.einsum("nik,nkl,njl->nij", Wsum, Vinv[pu], Wsum))
            Minv = np.linalg.inv(Sd)
        else:
            raise ValueError(f"unknown preconditioner {precond!r}")
        bvec = red.reshape(-1)
        xk = np.zeros_like(bvec)
        rk = bvec.copy()
        zk = np.einsum("cij,cj->ci", Minv, rk.reshape(C, 6)).reshape(-1)
        pk = zk.copy()
        rz = rk @ zk
        rz0 = rz
        maxit = pcg_maxiter or 10 * 6 * C
        it = 0
        while it < maxit and rz > (pcg_tol ** 2) * rz0 and rz > 0:
            Ap = S_mv(pk)
            alpha = rz / (pk @ Ap)
            xk += alpha * pk
            rk -= alpha * Ap
            zk = np.einsum("cij,cj->ci", Minv, rk.reshape(C, 6)).reshape(-1)
            rz_new = rk @ zk
            pk = zk + (rz_new / rz) * pk
            rz = rz_new
            it += 1
        if info is not None:
            info["pcg_iters"] = info.get("pcg_iters", 0) + it
        dc = xk.reshape(C, 6)
    # back-substitution: dp = Vinv (rhs_p - sum_i W_i^T dc_c)
    y = np.zeros((P, 3))
    np.add.at(y, pi, np.einsum("nij,ni->nj", nb.W, dc[ci]))
    dp = np.einsum("pij,pj->pi", Vinv, rhs_p - y)
    return dc, dp


# --------------------------------------------------------------------------------------------
# 2-D trust-region subproblem (SCIPY common.py:171-219) -- eigen/secular form, same minimiser
# --------------------------------------------------------------------------------------------


def solve_trust_region_2d(B, g, Delta):
    """min 0.5 p^T B p + g^T p  s.t. |p| <= Delta, B symmetric 2x2.  Returns (p, newton_step)."""
    B = np.asarray(B, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    det = B[0, 0] * B[1, 1] - B[0, 1] * B[1, 0]
    if B[0, 0] > 0 and det > 0:                      # positive definite -> Newton step
        p = -np.array([B[1, 1] * g[0] - B[0, 1] * g[1], -B[1, 0] * g[0] + B[0, 0] * g[1]]) / det
        if p @ p <= Delta ** 2:
            return p, True
    # boundary solution: scipy parametrises p = Delta (2t/(1+t^2), (1-t^2)/(1+t^2)) and takes the
    # best real root of a quartic (common.py:203-217); restated with the same quartic.
    a = B[0, 0] * Delta ** 2
    b = B[0, 1] * Delta ** 2
    c = B[1, 1] * Delta ** 2
    d = g[0] * Delta
    f = g[1] * Delta
    coeffs = np.array([-b + d, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - d])
    t = np.roots(coeffs)
    t = np.real(t[np.isreal(t)])
    p = Delta * np.vstack((2 * t / (1 + t ** 2), (1 - t ** 2) / (1 + t ** 2)))
    value = 0.5 * np.sum(p * B.dot(p), axis=0) + np.dot(g, p)
    return p[:, np.argmin(value)], False


# --------------------------------------------------------------------------------------------
# TRF outer loop (SCIPY trf.py:401-560) with a Schur step instead of LSMR
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass
class TRFResult:
    x: np.ndarray
    cost: float
    fun: np.ndarray
    grad: np.ndarray
    optimality: float
    nfev: int
    njev: int
    status: int
    history: list


def trf_schur(x0, n_cameras, n_points, camera_indices, point_indices, points_2d, K,
              ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, linear="dense",
              pcg_tol=1e-10, precond="block_u", reg_min=1e-6, verbose=0, comm=None, pcg_tol_max=None,
              fixed_cameras=()):
    """Restatement of trf_no_bounds(tr_solver='lsmr', x_scale='jac', loss='linear').

    ``pcg_tol_max`` (> ``pcg_tol``): the forcing term of the PCG adapts to the outer iteration as the HIP path's
    does (include/sfmba.h, sfmba_options.pcg_tol_max): eta_0 = pcg_tol_max, eta_k = min(pcg_tol_max, max(pcg_tol,
    |g_h(x_k)| / |g_h(x_k-1)|)) with g_h the scaled gradient of the iterate the system is solved at.

    Line references are to SCIPY/optimize/_lsq/trf.py.  Difference from scipy: gn_h of trf.py:480
    is the minimiser of |J_h p + f|^2 + reg |p|^2 through the Schur complement, not an LSMR iterate,
    and J is analytic rather than forward-differenced.  The 2-D subspace of trf.py:481-485 is
    orthonormalised by Gram-Schmidt from dot products (as the HIP path does) instead of numpy's QR.

    ``fixed_cameras``: the ``fixed_camera_indices`` of REF bundle_adjustment.py:6,13-14 -- the pattern handed to
    scipy leaves their camera columns empty, so its finite-difference Jacobian has zeros there (_numdiff.py:628-705
    only fills the pattern): restated by zeroing the camera blocks of their observations.

    ``comm`` (test infrastructure for the N>1 path): when given, (camera_indices, ...) describe ONE
    SHARD -- all cameras, the shard's own points and observations -- and every reduction the HIP path
    all-reduces goes through ``comm.sum`` / ``comm.max``; x holds [all cameras | local points].
    """
    comm = comm or NoComm()
    C, P = n_cameras, n_points
    n6 = 6 * C
    ci, pi = np.asarray(camera_indices), np.asarray(point_indices)
    uv = np.asarray(points_2d, dtype=np.float64)
    args = (C, P, ci, pi, uv, K)

    def Jdot(Jc, Jp, vec):           # J @ vec -> (N,2)
        vc = vec[:n6].reshape(C, 6)
        vp = vec[n6:].reshape(P, 3)
        return np.einsum("nki,ni->nk", Jc, vc[ci]) + np.einsum("nki,ni->nk", Jp, vp[pi])

    def dot(a, b):                   # camera slice is replicated, point slice is summed over shards
        return float(a[:n6] @ b[:n6]) + float(comm.sum(float(a[n6:] @ b[n6:])))

    held = np.isin(ci, np.asarray(list(fixed_cameras), dtype=np.int64))

    def linearise(x):
        r, Jc, Jp = jacobian_blocks(x, *args)
        if held.any():
            Jc[held] = 0.0
        nb = normal_blocks(r, Jc, Jp, C, P, ci, pi)
        nb.U = comm.sum(nb.U)
        nb.gc = comm.sum(nb.gc)
        return r, Jc, Jp, nb

    def col_norms(nb):
        d = np.concatenate([np.einsum("cii->ci", nb.U).ravel(), np.einsum("pii->pi", nb.V).ravel()])
        return np.sqrt(d)

    x = np.asarray(x0, dtype=np.float64).copy()
    r, Jc, Jp, nb = linearise(x)
    cost = 0.5 * float(comm.sum(float(np.sum(r * r))))                         # :418
    if not np.isfinite(cost):
        raise ValueError("Residuals are not finite in the initial point.")   # least_squares.py:844
    nfev = njev = 1
    g = np.concatenate([nb.gc.ravel(), nb.gp.ravel()])                         # :420
    scale_inv = col_norms(nb)                                                  # :424, common.py:598
    scale_inv[scale_inv == 0] = 1
    scale = 1.0 / scale_inv
    Delta = math.sqrt(dot(x * scale_inv, x * scale_inv))                       # :428
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = x.size * 100                                                # :437
    status = None
    iteration = 0
    gh2_prev = None                     # |g_h|^2 of the previous solved iterate (adaptive forcing term)
    step_norm = None
    actual_reduction = None
    history = []
    info = {}
    while True:                                                                # :450
        gmax_p = float(np.max(np.abs(g[n6:]))) if P else 0.0
        g_norm = max(float(np.max(np.abs(g[:n6]))), float(comm.max(gmax_p)))
        if g_norm < gtol:
            status = 1
        history.append(dict(iteration=iteration, nfev=nfev, cost=cost, reduction=actual_reduction,
                            step_norm=step_norm, optimality=g_norm,
                            pcg_iters=info.get("pcg_iters", 0)))
        if verbose:
            print(f"{iteration:6d} {nfev:6d} {cost:16.8e} "
                  f"{'' if actual_reduction is None else f'{actual_reduction:10.2e}':>10s} "
                  f"{'' if step_norm is None else f'{step_norm:10.2e}':>10s} {g_norm:10.2e} "
                  f"pcg={info.get('pcg_iters', 0)}")
        if status is not None or nfev == max_nfev:
            break
        d = scale
        g_h = d * g                                                            # :461
        sg = d * g_h                                                           # D^2 g
        # regularisation from the 1-D Cauchy problem, :471-475
        t1 = Jdot(Jc, Jp, sg)                                                  # J_h g_h
        G11 = float(comm.sum(float(np.sum(t1 * t1))))
        a11 = dot(g_h, g_h)
        a = 0.5 * G11
        b = -a11
        to_tr = Delta / math.sqrt(a11)
        ys = [0.0, to_tr * (a * to_tr + b)]
        if a != 0:
            ext = -0.5 * b / a
            if 0.0 < ext < to_tr:
                ys.append(ext * (a * ext + b))
        reg_term = max(-min(ys) / Delta ** 2, reg_min)   # floor: see DESIGN.md (rank-2 V_p blocks)
        # damped Gauss-Newton step, :477-480:  (J^T J + reg diag(scale_inv^2)) p = -g,  gn_h = p/d
        damp = reg_term * scale_inv ** 2
        info["pcg_iters"] = 0
        eta = pcg_tol
        if pcg_tol_max is not None and pcg_tol_max > pcg_tol:
            eta = pcg_tol_max if gh2_prev is None else min(pcg_tol_max, max(pcg_tol, math.sqrt(a11 / gh2_prev)))
        gh2_prev = a11
        dc, dp = schur_solve(nb, damp[:n6].reshape(C, 6), damp[n6:].reshape(P, 3), ci, pi,
                             -nb.gc, -nb.gp, method=linear, pcg_tol=eta, precond=precond,
                             info=info, comm=comm)
        p = np.concatenate([dc.ravel(), dp.ravel()])
        gn_h = p * scale_inv
        # 2-D subspace span(g_h, gn_h), Gram-Schmidt (:481-485)
        t2 = Jdot(Jc, Jp, p)                                                   # J_h gn_h
        G12 = float(comm.sum(float(np.sum(t1 * t2))))
        G22 = float(comm.sum(float(np.sum(t2 * t2))))
        a12, a22 = dot(g_h, gn_h), dot(gn_h, gn_h)
        b11, b12, b22 = dot(sg, sg), dot(sg, p), dot(p, p)
        x_norm = math.sqrt(dot(x, x))
        s11 = math.sqrt(a11)
        r12 = a12 / s11
        r22sq = a22 - r12 * r12
        two_d = r22sq > 1e-28 * a22 and r22sq > 0.0
        r22 = math.sqrt(r22sq) if two_d else 1.0
        if two_d:
            B_S = np.array([[G11 / a11, (G12 / s11 - r12 * G11 / a11) / r22],
                            [0.0, (G22 - 2.0 * r12 * G12 / s11 + r12 * r12 * G11 / a11) / (r22 * r22)]])
        else:
            B_S = np.array([[G11 / a11, 0.0], [0.0, 1.0]])
        B_S[1, 0] = B_S[0, 1]
        g_S = np.array([s11, 0.0])
        actual_reduction = -1.0
        while actual_reduction <= 0 and nfev < max_nfev:                       # :488
            p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
            if not two_d:
                p_S = np.array([p_S[0], 0.0])
            predicted_reduction = -(0.5 * float(p_S @ B_S @ p_S) + float(g_S @ p_S))
            c2 = p_S[1] / r22 if two_d else 0.0
            c1 = (p_S[0] - (p_S[1] * r12 / r22 if two_d else 0.0)) / s11
            step_h_norm = float(np.linalg.norm(p_S))
            x_new = x + c1 * sg + c2 * p                                       # step = D step_h
            r_new = compute_residuals(x_new, *args).reshape(-1, 2)
            nfev += 1
            cost_new = 0.5 * float(comm.sum(float(np.sum(r_new * r_new))))
            if not np.isfinite(cost_new):                                      # :504
                Delta = 0.25 * step_h_norm
                continue
            actual_reduction = cost - cost_new
            # update_tr_radius, common.py:222-245
            if predicted_reduction > 0:
                ratio = actual_reduction / predicted_reduction
            elif predicted_reduction == actual_reduction == 0:
                ratio = 1.0
            else:
                ratio = 0.0
            Delta_new = Delta
            if ratio < 0.25:
                Delta_new = 0.25 * step_h_norm
            elif ratio > 0.75 and step_h_norm > 0.95 * Delta:
                Delta_new = 2.0 * Delta
            step_norm = math.sqrt(max(0.0, c1 * c1 * b11 + 2.0 * c1 * c2 * b12 + c2 * c2 * b22))
            # check_termination, common.py:705-717
            ftol_ok = actual_reduction < ftol * cost and ratio > 0.25
            xtol_ok = step_norm < xtol * (xtol + x_norm)
            status = 4 if (ftol_ok and xtol_ok) else 2 if ftol_ok else 3 if xtol_ok else None
            if status is not None:
                break
            Delta = Delta_new
        if actual_reduction > 0:                                               # :528
            x = x_new
            r, Jc, Jp, nb = linearise(x)
            njev += 1
            cost = cost_new
            g = np.concatenate([nb.gc.ravel(), nb.gp.ravel()])
            scale_inv = np.maximum(col_norms(nb), scale_inv)                   # common.py:606
            scale = 1.0 / scale_inv
        else:
            step_norm = 0
            actual_reduction = 0
        iteration += 1
    if status is None:
        status = 0
    return TRFResult(x=x, cost=cost, fun=r.ravel(), grad=g, optimality=g_norm, nfev=nfev,
                     njev=njev, status=status, history=history)


# --------------------------------------------------------------------------------------------
# gauge alignment: comparing parameter vectors of two solvers ("gauge-aligned parameters")
# --------------------------------------------------------------------------------------------


def multi_view_points(n_cameras, n_points, camera_indices, point_indices, min_views=2):
    """Mask of the points seen by at least ``min_views`` DISTINCT cameras.  A point seen by one camera only has no
    defined depth (its 3x3 block has rank 2: it can slide along its ray at no cost), so two minimisers agree on
    everything except where such points sit; parameter comparisons are made on the others."""
    key = np.unique(np.asarray(point_indices, dtype=np.int64) * n_cameras + np.asarray(camera_indices, dtype=np.int64))
    return np.bincount(key // n_cameras, minlength=n_points) >= min_views


def similarity_align(x_src, x_dst, n_cameras, n_points, fit_points=None):
    """No camera is held fixed in the reference's call (REF bundle_adjustment.py:6, sfm.py:264), so the residual
    pi(K R_c (X - T_c)) is invariant under the 7-dof similarity X -> s R X + t, T_c -> s R T_c + t, R_c -> R_c R^T:
    two minimisers of the same problem may differ by one.  Fits (s, R, t) on the POINTS (least squares, Umeyama's
    closed form; ``fit_points``: bool mask of the points to fit on, see multi_view_points) so that the mapped
    ``x_src`` is closest to ``x_dst`` and returns the mapped parameter vector together with (s, R, t)."""
    n6 = 6 * n_cameras
    A = np.asarray(x_src[n6:], dtype=np.float64).reshape(n_points, 3)
    B = np.asarray(x_dst[n6:], dtype=np.float64).reshape(n_points, 3)
    sel = np.ones(n_points, dtype=bool) if fit_points is None else np.asarray(fit_points, dtype=bool)
    ma, mb = A[sel].mean(axis=0), B[sel].mean(axis=0)
    A0, B0 = A[sel] - ma, B[sel] - mb
    n_fit = int(sel.sum())
    Uu, Sv, Vt = np.linalg.svd(B0.T @ A0 / n_fit)
    d = np.ones(3)
    if np.linalg.det(Uu) * np.linalg.det(Vt) < 0:
        d[2] = -1.0
    R = Uu @ np.diag(d) @ Vt
    s = float(np.sum(Sv * d) / (np.sum(A0 * A0) / n_fit))
    t = mb - s * R @ ma
    cams = np.asarray(x_src[:n6], dtype=np.float64).reshape(n_cameras, 6)
    out = np.empty_like(np.asarray(x_src, dtype=np.float64))
    oc = out[:n6].reshape(n_cameras, 6)
    Rc = rodrigues(cams[:, :3])
    for c in range(n_cameras):
        oc[c, :3] = rotvec_from_matrix(Rc[c] @ R.T)
        oc[c, 3:] = s * R @ cams[c, 3:] + t
    out[n6:] = (s * A @ R.T + t).ravel()
    return out, (s, R, t)


def parameter_distance(x_a, x_b, n_cameras, n_points, observed_cameras=None, points=None):
    """How far apart two parameter vectors of one problem are, in the units a user reads them in: RMS and maximum
    distance of corresponding points (``points``: bool mask, see multi_view_points), maximum distance of camera
    centres, maximum angle (degrees) between camera rotations.  Cameras nobody observes carry no information and can be
    left out (``observed_cameras``: bool mask)."""
    n6 = 6 * n_cameras
    ca = np.asarray(x_a[:n6]).reshape(n_cameras, 6)
    cb = np.asarray(x_b[:n6]).reshape(n_cameras, 6)
    pa = np.asarray(x_a[n6:]).reshape(n_points, 3)
    pb = np.asarray(x_b[n6:]).reshape(n_points, 3)
    keep = np.ones(n_cameras, dtype=bool) if observed_cameras is None else np.asarray(observed_cameras, dtype=bool)
    dp = np.linalg.norm(pa - pb, axis=1)
    if points is not None:
        dp = dp[np.asarray(points, dtype=bool)]
    dR = np.einsum("cij,ckj->cik", rodrigues(ca[:, :3]), rodrigues(cb[:, :3]))
    ang = np.degrees(np.arccos(np.clip(0.5 * (np.trace(dR, axis1=1, axis2=2) - 1.0), -1.0, 1.0)))
    return dict(points_rms=float(np.sqrt(np.mean(dp ** 2))), points_max=float(dp.max()),
                centres_max=float(np.linalg.norm(ca[keep, 3:] - cb[keep, 3:], axis=1).max()),
                rot_deg_max=float(ang[keep].max()))


# --------------------------------------------------------------------------------------------
# pack / unpack (REF sfm.py:248-262, 271-281)
# --------------------------------------------------------------------------------------------


def pack_problem(H_list, registered, X3d, observations):
    """``observations`` = iterable of (point_idx, cam_id, (x, y)) in the order of
    Graph.pt3ds_pt2ds (REF graph.py:186-191).  ``H_list[k]`` is the 4x4 pose of node k,
    ``registered[k]`` its flag.  Returns (x0, n_cam, n_points, camera_indices, pt_indices, pt2ds,
    camera_map) exactly as sfm.py:248-262 builds them (T is H[:3,3] verbatim, sfm.py:252)."""
    data = list(observations)
    pt_indices = np.array([d[0] for d in data])
    cam_ids = np.array([d[1] for d in data])
    pt2ds = np.array([d[2] for d in data])
    reg = [k for k, f in enumerate(registered) if f]
    camera_map = {k: i for i, k in enumerate(reg)}
    params = []
    for k in reg:
        H = np.asarray(H_list[k], dtype=np.float64)
        params.append(np.hstack([rotvec_from_matrix(H[:3, :3]), H[:3, 3].flatten()]))
    camera_indices = np.array([camera_map[c] for c in cam_ids])
    X3d = np.asarray(X3d, dtype=np.float64)
    x0 = np.hstack([np.hstack(params).ravel(), X3d.ravel()])
    return x0, len(reg), len(X3d), camera_indices, pt_indices, pt2ds, camera_map


def unpack_result(x, n_cam, n_points, camera_map, H_list):
    """REF sfm.py:271-281: writes R(w), T back into the 4x4 poses; returns (H_list, X3d)."""
    cams = np.asarray(x[:n_cam * 6]).reshape((n_cam, 6))
    H_out = [np.array(H, dtype=np.float64, copy=True) for H in H_list]
    for cam_id, n in camera_map.items():
        H = np.eye(4)
        H[:3, :3] = rodrigues(cams[n, :3])
        H[:3, 3] = cams[n, 3:]
        H_out[cam_id] = H
    return H_out, np.asarray(x[n_cam * 6:]).reshape((n_points, 3))
