"""Per-kernel means of the counters in a rocprofv3 --pmc ... --output-format csv run.
    python scripts/pmc_summary.py <dir>      (searches *counter_collection.csv below <dir>)"""
import csv, glob, os, sys, collections
root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0][:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    n = max(len(v) for v in cs.values())
    print("%-60s launches=%d" % (k, n))
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.4g   sum %.4g" % (c, sum(v) / len(v), sum(v)))
