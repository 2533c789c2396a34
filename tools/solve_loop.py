"""Back-to-back solves of one config (scratch tool for rocprofv3 traces): python tools/solve_loop.py cfg4 20 [storage_bits]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import sfmba
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if "," in cfg:                       # custom size "C,P,N" (e.g. one rank's share of a sharded problem)
    C_, P_, N_ = (int(v) for v in cfg.split(","))
    pb = sfmba.make_problem(C_, P_, N_, seed=0)
else:
    pb = sfmba.make_config(cfg)
be = sfmba.Backend(0)
be.set_precision(bits)
for kv in os.environ.get("SFMBA_DEBUG", "").split(","):      # e.g. SFMBA_DEBUG=cost_rider=0,pcg_split=1 (tool-level, not the library)
    if "=" in kv:
        be.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
be.set_problem(*pb.args)
opt = be.default_options()
opt.ftol = 1e-10
opt.profile = 1 if os.environ.get("SFMBA_K1_EVENTS") else 0      # K1 bracketed by HIP events (in-solve duration)
out = []
for k in range(n):
    x, res, _, _ = be.solve(pb.x0, opt, want_fun=False, want_grad=False)
    out.append((res.iterations, res.nfev, res.pcg_iterations, res.seconds_total, res.rmse, res.resjac_avg_us))
it = sum(o[0] for o in out[n // 2:]); t = sum(o[3] for o in out[n // 2:])
print(cfg, bits, "iterations/solve", sorted(set(o[0] for o in out)), "pcg", sorted(set(o[2] for o in out)),
      "rmse", out[-1][4], "it/s %.1f" % (it / t), "ms/solve %.3f" % (1e3 * t / (n - n // 2)),
      *(("K1 in-solve us %.2f" % (sum(o[5] for o in out[n // 2:]) / (n - n // 2)),) if opt.profile else ()))
