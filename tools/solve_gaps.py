"""Span / busy / gap accounting of the last solve in a rocprofv3 kernel trace (scratch analysis)."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a solve starts with k_cam_table (initial evaluation); find the last one
starts = [i for i, r in enumerate(rows) if "k_cam_table" in r["Kernel_Name"]]
i0 = starts[-1]
seg = rows[i0:]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0; busy = 0; gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - prev > 2000: gaps.append(((s - prev) / 1e3, r["Kernel_Name"][:40]))
    busy += e - s; prev = e
print("kernels", len(seg), "span us %.1f busy us %.1f" % ((prev - t0) / 1e3, busy / 1e3))
print("gaps > 2us:", len(gaps), "total %.1f us" % sum(g for g, _ in gaps))
for g, n in gaps: print("   %.1f before %s" % (g, n))
import collections, re
tot = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sfmba::", "")
    tot[name][0] += 1; tot[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("per kernel in this solve:")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s %4d calls %8.1f us  %5.1f %%" % (n[:40], c, t, 100 * t / (busy / 1e3)))
