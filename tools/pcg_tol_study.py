"""Outer/inner iteration counts and final RMSE as a function of the PCG tolerance (scratch study)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import numpy as np, sfmba
for cfg in ("cfg2", "cfg3", "cfg4"):
    pb = sfmba.make_config(cfg)
    for tol in (1e-8, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1):
        best = None
        for rep in range(2):
            res = sfmba.least_squares(sfmba.compute_residuals, pb.x0, x_scale="jac", ftol=1e-10, method="trf",
                                      args=pb.args, pcg_tol=tol)
            best = res if best is None or res.seconds_device < best.seconds_device else best
        print(f"{cfg} pcg_tol {tol:7.0e}: status {best.status} it {best.iterations} nfev {best.nfev} pcg {best.pcg_iterations:4d} "
              f"rmse {best.rmse:.9f} cost {best.cost:.6f} opt {best.optimality:.2e} dev {1e3*best.seconds_device:7.2f} ms", flush=True)
