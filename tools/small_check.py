"""One-launch solver of small problems (csrc/small_solve.hpp) against the multi-launch loop and the oracle (scratch check):
python tools/small_check.py [with_oracle]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np
import sfmba

with_oracle = len(sys.argv) > 1
if with_oracle:
    from oracle import ba_oracle as orc
cases = [("cfg2", None), ("2,300,700", 1), ("5,900,3000", 3), ("21,5000,16000", 7), ("7,40,200", 8)]
for name, seed in cases:
    if seed is None:
        pb = sfmba.make_config(name)
    else:
        C_, P_, N_ = (int(v) for v in name.split(","))
        pb = sfmba.make_problem(C_, P_, N_, seed=seed, x0_noise=0.03)
    rows = []
    for small in (1, 2, 0):
        be = sfmba.Backend(0)
        be.debug_option("small", min(small, 1))
        be.debug_option("small_agent", 1 if small == 2 else 0)
        be.set_problem(*pb.args)
        opt = be.default_options(); opt.ftol = 1e-10
        if small: be.debug_option("trace_timing", 1)
        for mi in (0,):                     # the first iterations one by one
            opt.max_iter = mi
            x, res, _, _ = be.solve(pb.x0, opt, want_fun=False, want_grad=False)
            print(f"   small={small} max_iter={mi}: cost {res.cost:.15g} nfev {res.nfev} pcg {res.pcg_iterations} reg {res.last_reg:.6e} step {res.last_step_norm:.6e}", flush=True)
        opt.max_iter = 0
        be.debug_option("trace_timing", 0)
        t = []
        for k in range(6):
            t0 = time.perf_counter()
            x, res, fun, grad = be.solve(pb.x0, opt, want_fun=(k == 0), want_grad=(k == 0))
            t.append(time.perf_counter() - t0)
            if k == 0: f0, g0, x0_ = fun, grad, x
        rows.append((small, res.status, res.nfev, res.njev, res.iterations, res.pcg_iterations, res.cost, res.rmse, res.optimality,
                     1e3 * min(t), 1e3 * res.seconds_total, be.pcg_history(), x0_, f0, g0, be.counters()[0] if hasattr(be, "counters") else -1))
        be.close()
    a, a2, b = rows
    print('   agent-scope barriers: status', a2[1], 'nfev', a2[2], 'cost', a2[6], 'ms %.3f' % a2[9])
    same = a[1:6] == b[1:6]
    print(f"{name:16s} small: status {a[1]} nfev {a[2]} njev {a[3]} it {a[4]} pcg {a[5]} cost {a[6]:.12g} rmse {a[7]:.9f} opt {a[8]:.3e} "
          f"{a[9]:.3f} ms | loop: status {b[1]} nfev {b[2]} njev {b[3]} pcg {b[5]} cost {b[6]:.12g} {b[9]:.3f} ms | "
          f"{'SAME' if same else 'DIFF'} dcost {abs(a[6] - b[6]) / b[6]:.1e} dx {np.abs(a[12] - b[12]).max():.1e} "
          f"dfun {np.abs(a[13] - b[13]).max():.1e} dgrad {np.abs(a[14] - b[14]).max():.1e} hist {a[11] == b[11]}", flush=True)
    if with_oracle and pb.n_obs <= 20000:
        o = orc.trf_schur(pb.x0, *pb.args, ftol=1e-10, linear="pcg", pcg_tol=1e-3, precond="schur_exact")
        print(f"{'':16s} oracle: status {o.status} nfev {o.nfev} njev {o.njev} cost {o.cost:.12g}  dcost {abs(a[6] - o.cost) / o.cost:.1e}", flush=True)
