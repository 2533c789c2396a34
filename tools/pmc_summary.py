"""Per-kernel averages of the counters of one rocprofv3 --pmc run: python tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:90])
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(d.items())) + f"  (n={len(next(iter(d.values())))})")
