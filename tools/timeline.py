"""Print the kernel timeline (start offset, duration, gap before) of the tail of a rocprofv3 kernel trace."""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    busy += e - s; prev_end = e
print("span us", (prev_end - t0) / 1e3, "busy us", busy / 1e3)
