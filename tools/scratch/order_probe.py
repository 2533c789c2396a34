"""Scratch: libsfmba first, torch.cuda afterwards (and the other way round) in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
order = sys.argv[1]
if order == "sfmba_first":
    import sfmba
    be = sfmba.Backend(0)
    pb = sfmba.make_config("cfg2"); be.set_problem(*pb.args)
    r = be.residuals(pb.x0)
    import torch
    torch.cuda.set_device(0)
    t = torch.ones(4, device="cuda") * 2
    print(order, "ok", float(t.sum()), float(abs(r).max()) > 0, [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
else:
    import torch
    torch.cuda.set_device(0)
    t = torch.ones(4, device="cuda") * 2
    import sfmba
    be = sfmba.Backend(0)
    pb = sfmba.make_config("cfg2"); be.set_problem(*pb.args)
    r = be.residuals(pb.x0)
    print(order, "ok", float(t.sum()), float(abs(r).max()) > 0, [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
