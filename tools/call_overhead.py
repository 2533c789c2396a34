"""Where the time of ONE drop-in call goes (scratch measurement): sfmba.least_squares(...) as the reference
calls it -- create_sparsity_matrix, argument conversion, set_problem (host preprocessing + upload), solve."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np
import sfmba

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
pb = sfmba.make_config(cfg)
C, P, N = pb.n_cameras, pb.n_points, pb.n_obs
t = time.perf_counter(); S = sfmba.create_sparsity_matrix(C, P, N, pb.camera_indices, pb.point_indices)
print(f"create_sparsity_matrix {1e3 * (time.perf_counter() - t):8.2f} ms", flush=True)
be = sfmba.get_backend(0)
be.debug_option("trace_timing", 1)
for rep in range(3):
    t = time.perf_counter(); be.set_problem(*pb.args); t1 = time.perf_counter()
    opt = be.default_options(); opt.ftol = 1e-10
    x, res, fun, grad = be.solve(pb.x0, opt); t2 = time.perf_counter()
    x, res, _, _ = be.solve(pb.x0, opt, want_fun=False, want_grad=False); t3 = time.perf_counter()
    print(f"backend: set_problem {1e3 * (t1 - t):8.2f} ms   solve+fun+grad {1e3 * (t2 - t1):8.2f} ms   solve only {1e3 * (t3 - t2):8.2f} ms "
          f"(C call {1e3 * res.seconds_total:.2f} ms, device {1e3 * res.seconds_device:.2f} ms, {res.iterations} iterations)", flush=True)
be.debug_option("trace_timing", 0)
for rep in range(3):
    t = time.perf_counter()
    r = sfmba.least_squares(sfmba.compute_residuals, pb.x0, jac_sparsity=S, x_scale="jac", ftol=1e-10, method="trf",
                            args=pb.args)
    print(f"sfmba.least_squares total {1e3 * (time.perf_counter() - t):8.2f} ms  (solve {1e3 * r.seconds:.2f} ms)", flush=True)
