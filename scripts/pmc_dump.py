"""Print per-kernel PMC sums from a rocprofv3 --output-format csv counter_collection file."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "msda" not in k: continue
    calls = len({r["Dispatch_Id"] for r in rows if r["Kernel_Name"][:60] == k})
    print(k, "calls", calls)
    for c, v in sorted(d.items()): print("   %-28s %16.0f  per call %14.0f" % (c, v, v / calls))
