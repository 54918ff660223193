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

# --json <out> <commit> <box> <steps> <res> <spp>: HBM bytes per point of the MLP kernels = (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over ALL
# launches of the profiled run / the points those launches processed (steps = warm-up + timed steps of the profiled bench command; the
# backward pair runs once per slab, the SDF kernel once per sampling stage), from a directory holding both a FETCH_SIZE and a WRITE_SIZE pass
# (bench.py reads the file: roofline.traffic)
if "--json" in sys.argv:
    import json
    i = sys.argv.index("--json")
    out, commit, box = sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3]
    steps, res_, spp = int(sys.argv[i + 4]), int(sys.argv[i + 5]), int(sys.argv[i + 6])
    pts_train = float(steps * res_ * res_ * spp)
    pts_sdf = float(steps * res_ * res_ * (spp // 2 + (spp // 2) * 3 // 4))
    # kernel names appear mangled or demangled depending on the rocprofv3 pass: match either form
    names = {"mlp_render_kernel<train>": (("mlp_render_kernelI4NetTILi256ELi2ELi1EELb1", "mlp_render_kernel<NetT<256, 2, 1>, true>"), pts_train),
             "mlp_bwd_kernel": (("mlp_bwd_kernelI4NetTILi256", "mlp_bwd_kernel<NetT<256"), pts_train),
             "weight_grad_all_kernel": (("weight_grad_all_kernel",), pts_train),
             "mlp_sdf_kernel": (("mlp_sdf_kernelI4NetTILi256", "mlp_sdf_kernel<NetT<256"), pts_sdf)}
    res = {}
    for label, (keys, npts) in names.items():
        ks = [k for k in acc if any(key in k for key in keys)]
        if not ks:
            continue
        cs = acc[ks[0]]
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f, w = sum(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"])           # KiB over all launches of the run
            res[label] = {"fetch_x2_bytes_per_step": 2 * f * 1024 / steps, "write_bytes_per_step": w * 1024 / steps,
                          "launches_per_step": len(cs["FETCH_SIZE"]) / steps, "bytes_per_point": (2 * f + w) * 1024 / npts}
    json.dump({"commit": commit, "box": box, "steps_profiled": steps, "config": "%dx%d rays, %d spp" % (res_, res_, spp),
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 wide-load correction of "
                         "MI355X_MICROARCH.md); sums over all launches of the profiled steps / the points processed", "kernels": res},
              open(out, "w"), indent=1)
    print("wrote", out, res)
