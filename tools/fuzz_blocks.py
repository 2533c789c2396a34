"""Randomised parity of the kernels' building blocks at medium sizes (scratch stress tool): residual, Jacobian, normal
equation blocks (point rows summed inside K1 over 64-observation tiles, cut runs finished by riders) and the implicit
Schur product against the oracle, for random track-length distributions, fp64 and fp32 storage.
python tools/fuzz_blocks.py [n_cases] [seed]

Round 2, MI355X: seed 3, 40 cases and seed 11, 200 cases (3 ... 1400 cameras, up to 700 observations per point, every
third case in fp32 storage): no case outside the tolerances of tests/test_gpu_parity.py; worst relative errors V 3e-15,
g_p 2e-15, U 1e-13, g_c 2e-14, Schur product 7e-14."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np
import sfmba
from oracle import ba_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
be = sfmba.Backend(0)
iu3, iu6 = np.triu_indices(3), np.triu_indices(6)
worst = dict(r=0.0, V=0.0, gp=0.0, U=0.0, gc=0.0, mv=0.0)
t0 = time.time()
for case in range(n_cases):
    C = int(rng.integers(3, 1400)); P = int(rng.integers(50, 6000))
    kind = int(rng.integers(0, 5))
    lens = (rng.geometric(0.12, P) if kind == 0 else np.minimum(1 + (rng.pareto(1.1, P) * 3).astype(int), 700) if kind == 1
            else rng.integers(0, 5, P) if kind == 2 else rng.integers(1, 200, P) if kind == 3 else np.full(P, int(rng.integers(1, 70))))
    lens = np.asarray(lens, dtype=np.int64); lens[-1] = max(lens[-1], 1)
    N = int(lens.sum())
    base = sfmba.make_problem(C, P, max(N, P), seed=int(rng.integers(1 << 30)))
    pi = np.repeat(np.arange(P, dtype=np.int64), lens)
    ci = rng.integers(0, C, N).astype(np.int64)
    uv = base.points_2d[:N]
    args = (C, P, ci, pi, uv, base.K)
    bits = 32 if case % 3 == 2 else 64
    be.set_precision(bits)
    be.set_problem(*args)
    U, V, gc, gp = be.normal_blocks(base.x0)
    r, Jc, Jp = orc.jacobian_blocks(base.x0, *args)
    nb = orc.normal_blocks(r, Jc, Jp, C, P, ci, pi)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))
    e = dict(V=rel(V, nb.V[:, iu3[0], iu3[1]]), gp=rel(gp, nb.gp), U=rel(U, nb.U[:, iu6[0], iu6[1]]), gc=rel(gc, nb.gc))
    dc = 1e-3 * np.einsum("cii->ci", nb.U) + 1e-6
    dp = 1e-3 * np.einsum("pii->pi", nb.V) + 1e-6
    v = rng.normal(size=6 * C)
    y = be.schur_matvec(base.x0, dc, dp, v)
    Vd = nb.V.copy(); Vd[:, np.arange(3), np.arange(3)] += dp
    Vinv = np.linalg.inv(Vd); vc = v.reshape(C, 6)
    yy = np.zeros((P, 3)); np.add.at(yy, pi, np.einsum("nij,ni->nj", nb.W, vc[ci]))
    z = np.einsum("pij,pj->pi", Vinv, yy)
    ref = np.einsum("cij,cj->ci", nb.U, vc) + dc * vc
    np.add.at(ref, ci, -np.einsum("nij,nj->ni", nb.W, z[pi]))
    e["mv"] = rel(y, ref.ravel())
    e["r"] = float(np.abs(be.residuals(base.x0) - r.ravel()).max() / max(3000.0, np.abs(r).max()))
    tol = dict(V=1e-11, gp=1e-10, U=1e-11, gc=1e-10, mv=1e-9, r=(1e-6 if bits == 32 else 1e-11))
    bad = [k for k in e if not e[k] <= tol[k]]
    for k in e:
        if not (k == "r" and bits == 32): worst[k] = max(worst[k], e[k])
    print(f"case {case}: C={C} P={P} N={N} kind={kind} bits={bits} " + " ".join(f"{k}={e[k]:.1e}" for k in e) + ("  BAD " + ",".join(bad) if bad else ""), flush=True)
print("worst", {k: f"{v:.1e}" for k, v in worst.items()}, f"{time.time() - t0:.0f} s")
