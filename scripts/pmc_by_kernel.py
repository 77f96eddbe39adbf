"""Sum of every counter per kernel over the dispatches of a rocprofv3 counter_collection csv.   usage: python scripts/pmc_by_kernel.py <csv>"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "?").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = (k, r.get("Dispatch_Id"))
    if d not in seen:
        seen.add(d); nd[k] += 1
for k in sorted(acc, key=lambda k: -max(acc[k].values())):
    print("%-40s n=%d  " % (k[:40], nd[k]) + "  ".join("%s=%.6g" % (c, v) for c, v in sorted(acc[k].items())))
