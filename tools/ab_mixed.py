"""A/B of the mixed-precision Schur product (debug options pcg_mixed, pcg_mixed_b): kernel times, product accuracy /
symmetry, whole solves.   python tools/ab_mixed.py cfg4 [storage_bits]      (scratch tool, not a test)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np
import sfmba

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if "," in cfg:
    pb = sfmba.make_problem(*(int(v) for v in cfg.split(",")), seed=0)
else:
    pb = sfmba.make_config(cfg)
C, P = pb.n_cameras, pb.n_points
rng = np.random.default_rng(0)
dc, dp = rng.uniform(0.5, 2.0, 6 * C) * 1e3, rng.uniform(0.5, 2.0, 3 * P) * 1e3
u, v = rng.normal(size=6 * C), rng.normal(size=6 * C)
ref = {}
for mixed, mixed_b in ((0, 0), (1, 0), (1, 1), (0, 0), (1, 0), (1, 1)):
    be = sfmba.Backend(0)
    be.set_precision(bits)
    be.debug_option("pcg_mixed", mixed)
    be.debug_option("pcg_mixed_b", mixed_b)
    be.debug_option("dense", 0)
    be.set_problem(*pb.args)
    t = {name: be.time_kernel(pb.x0, which, 30) for which, name in ((3, "A+B"), (4, "A"), (5, "B"))}
    Su, Sv = be.schur_matvec(pb.x0, dc, dp, u), be.schur_matvec(pb.x0, dc, dp, v)
    if mixed == 0:
        ref["Su"] = Su
    err = np.abs(Su - ref["Su"]).max() / np.abs(ref["Su"]).max()
    sym = abs(v @ Su - u @ Sv) / abs(v @ Su)
    opt = be.default_options(); opt.ftol = 1e-10
    runs = []
    for k in range(10):
        x, res, _, _ = be.solve(pb.x0, opt, want_fun=False, want_grad=False)
        runs.append(res.seconds_total)
    hist = be.pcg_history()
    print(f"{cfg} bits {bits} mixed A {mixed} B {mixed_b}: A+B {t['A+B']:.1f} us  A {t['A']:.1f}  B {t['B']:.1f} | product vs fp64 {err:.2e} "
          f"asymmetry {sym:.2e} | solve {1e3 * np.median(runs[4:]):.3f} ms  it {res.iterations} nfev {res.nfev} pcg {hist} "
          f"cost {res.cost:.12e} status {res.status} -> {res.iterations / np.median(runs[4:]):.0f} it/s", flush=True)
    be.close()
