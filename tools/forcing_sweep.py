"""Outer iterations, PCG iterations, time per solve and final cost against the two parameters of the PCG forcing
term, eta_k = min(pcg_tol_max, max(pcg_tol, |g_h(x_k)| / |g_h(x_k-1)|)) (scratch tool):
python tools/forcing_sweep.py cfg4 [storage_bits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import sfmba
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pb = sfmba.make_config(cfg)
be = sfmba.Backend(0)
be.set_precision(bits)
be.set_problem(*pb.args)
for tol, tol_max in ((1e-3, 0.0), (1e-2, 0.0), (3e-2, 0.0), (1e-1, 0.0), (1e-2, 0.1), (1e-2, 0.3), (3e-3, 0.1), (1e-3, 0.1),
                     (1e-4, 0.1), (3e-2, 0.1)):
    opt = be.default_options()
    opt.ftol = 1e-10
    opt.pcg_tol = tol
    opt.pcg_tol_max = tol_max
    out = []
    for k in range(8):
        x, res, _, _ = be.solve(pb.x0, opt, want_fun=False, want_grad=False)
        out.append((res.iterations, res.nfev, res.pcg_iterations, res.seconds_total, res.rmse, res.cost, res.status))
    t = sum(o[3] for o in out[4:]) / 4
    o = out[-1]
    print(cfg, bits, "tol %.0e max %.0e" % (tol, tol_max), "outer", o[0], "nfev", o[1], "pcg", o[2], "status", o[6],
          "cost %.12e" % o[5], "ms/solve %.3f" % (1e3 * t), "it/s %.1f" % (o[0] / t), flush=True)
