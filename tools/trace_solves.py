"""Back-to-back solves WITHOUT the event-bracketed K1 launches of bench.py (profile = 0): the subject for
`rocprofv3 --kernel-trace` + tools/solve_gaps.py when the queue gaps of the plain solver are of interest.
    python tools/trace_solves.py [cfg4] [n_solves]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import sfmba

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
pb = sfmba.make_config(cfg)
be = sfmba.get_backend(0)
be.set_problem(*pb.args)
opt = be.default_options()
opt.ftol = 1e-10
opt.profile = 0
for _ in range(n):
    x, res, _, _ = be.solve(pb.x0, opt, want_fun=False, want_grad=False)
print(f"{cfg}: {n} solves, last: {res.iterations} iterations, {1e3 * res.seconds_total:.3f} ms, rmse {res.rmse:.9f}")
