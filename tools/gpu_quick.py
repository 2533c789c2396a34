"""Quick GPU bring-up: smoke, then kernel timings at a given config (scratch tool, not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np
import sfmba

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    import __graft_entry__ as ge
    t = time.time(); ge.smoke(); print("smoke s", time.time() - t, flush=True)
    C, P, N = sfmba.synthetic.CONFIGS[cfg]
    t = time.time(); pb = sfmba.make_config(cfg); print("gen s", time.time() - t, flush=True)
    be = sfmba.get_backend(0)
    t = time.time(); be.set_problem(*pb.args); print("set_problem s", time.time() - t, flush=True)
    for which, name, nbytes in [(0, "resjac+blocks", 136 * N + 96 * P + 48 * C), (7, "resjac alone", 136 * N + 24 * P + 48 * C),
                                (1, "residual", 40 * N + 24 * P + 48 * C), (2, "camera_blocks", 52 * N + 216 * C),
                                (3, "schur_product", 64 * N + 72 * P), (4, "pass_A", 8 * N + 72 * P),
                                (5, "pass_B", 52 * N + 232 * C), (6, "rhs_pass", 52 * N + 184 * C)]:
        us = be.time_kernel(pb.x0, which, 20)
        print(f"{name:14s} {us:9.2f} us  {nbytes / us / 1e3:8.1f} GB/s (algorithmic)", flush=True)
    for it in range(2):
        t = time.time()
        res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                  args=pb.args, verbose=2 if it == 0 else 0, profile=True)
        dt = time.time() - t
        print(f"solve: {dt:.3f}s status {res.status} it {res.iterations} nfev {res.nfev} pcg {res.pcg_iterations} "
              f"rmse {res.rmse0:.4f}->{res.rmse:.9f} dev {res.seconds_device:.4f}s  "
              f"{res.iterations / res.seconds_device:.1f} it/s  K1 {res.resjac_avg_us:.1f} us", flush=True)

main()
