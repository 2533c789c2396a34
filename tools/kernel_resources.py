"""Summarise build/resource-usage.txt (hipcc -Rpass-analysis=kernel-resource-usage) per kernel."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/resource-usage.txt").read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    sc, oc, ld = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print("%-70s VGPR %4s AGPR %3s SGPR %3s scratch %3s occ %s LDS %s" % (name[:70], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), sc, oc, ld))
