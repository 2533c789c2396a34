"""Time individual kernels (HIP events, back-to-back launches) at one config: scratch tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sfm-python_amd")]
import sfmba
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
which = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 7, 1, 2, 3, 4, 5, 6]
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 64
pb = sfmba.make_config(cfg)
be = sfmba.get_backend(0)
be.set_precision(bits)
for kv in os.environ.get("SFMBA_DEBUG", "").split(","):      # e.g. SFMBA_DEBUG=xcd_chunks=0 (tool-level, not the library)
    if "=" in kv:
        be.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
be.set_problem(*pb.args)
names = {0: "resjac+point_blocks", 7: "resjac_alone", 1: "residual", 2: "camera_blocks(K3+riders)", 3: "schur_product(A+B)", 4: "pass_A", 5: "pass_B", 6: "rhs_pass", 8: "rhs+precond_blocks", 10: "fill16_512wg(96MB)"}
for w in which:
    us = [be.time_kernel(pb.x0, w, 20) for _ in range(3)]
    print(os.environ.get("SFMBA_LIB", "default").split("/")[-1], os.environ.get("SFMBA_DEBUG", ""), cfg, bits, names[w], " ".join("%.2f" % u for u in us), flush=True)
