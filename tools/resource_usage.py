"""Register / scratch / LDS usage of every kernel from `make -C sfm-python_amd asm` (build/resource-usage.txt)."""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = open(os.path.join(ROOT, "sfm-python_amd", "build", "resource-usage.txt")).read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
n_scratch = 0
for b in blocks:
    name = b.split("\n")[0].split(" [")[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    v, sc, oc, lds = g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    n_scratch += sc > 0
    if (pat and pat in name) or sc > 0:
        print(f"{name[:90]:90s} vgpr {v:4d} scratch {sc:4d} occ {oc} lds {lds}")
print(len(blocks), "kernels,", n_scratch, "with scratch")
