"""Effective shader clock per kernel = GRBM_GUI_ACTIVE cycles / kernel duration, from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace
--output-format csv` run (MI355X_MICROARCH.md, "DVFS give-back": the chip clocks to its power budget).   python scripts/eff_clock.py <dir>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
cnt = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        d = cnt[row["Dispatch_Id"]]
        d["name"], d["cycles"] = row["Kernel_Name"].split("(")[0][:70], float(row["Counter_Value"])
        if "Start_Timestamp" in row and row.get("End_Timestamp"):
            d["ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        d = cnt.get(row.get("Dispatch_Id"))
        if d is not None and "ns" not in d:
            d["ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d in cnt.values():
    if "ns" in d and d["ns"] > 2e5:          # kernels longer than 0.2 ms only
        a = agg[d["name"]]
        a[0] += d["cycles"]; a[1] += d["ns"]; a[2] += 1
print("%-72s %8s %10s %10s" % ("kernel (> 0.2 ms launches)", "launches", "avg ms", "GHz"))
for k, (c, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %8d %10.3f %10.3f" % (k, n, ns / n * 1e-6, c / ns))
