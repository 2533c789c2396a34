"""Time individual kernels (HIP events, back-to-back launches) at one config: scratch tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import sfmba
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
which = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
pb = sfmba.make_config(cfg)
be = sfmba.get_backend(0)
be.set_problem(*pb.args)
names = {0: "resjac", 1: "residual", 2: "normal_blocks", 3: "schur_sweep", 10: "fill16_512wg(96MB)", 11: "fill16_2048wg(96MB)", 12: "resjac_alt_buffers", 13: "fill16_cold(144MB/rep,2 launches)"}
for w in which:
    us = [be.time_kernel(pb.x0, w, 20) for _ in range(3)]
    print(os.environ.get("SFMBA_LIB", "default").split("/")[-1], names[w], " ".join("%.2f" % u for u in us), flush=True)
