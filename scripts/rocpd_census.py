"""Launch census of ONE steady-state training step from a rocprofv3 rocpd database (development aid): the ordered kernel sequence
between two consecutive weight_grad_all launches that end a step, cut into phases at the big MLP kernels.
    python scripts/rocpd_census.py <dir|db> [full_sequence_out.txt]"""
import collections, glob, os, sqlite3, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
BIG = ("mlp_sdf_kernel", "mlp_render_kernel", "mlp_bwd_kernel", "weight_grad_all_kernel")
tag = lambda n: next((b for b in BIG if b in n), None)
# a step ends with the last weight_grad launch before the next mlp_sdf launch
ends = [i for i, r in enumerate(rows) if "weight_grad_all" in r[2] and (i + 1 == len(rows) or not any("weight_grad_all" in x[2] or "mlp_bwd" in x[2] for x in rows[i + 1:i + 40]))]
if len(ends) < 3:
    sys.exit("need at least 3 steps in the trace")
i0, i1 = ends[-3] + 1, ends[-2] + 1
step = rows[i0:i1]
span = (step[-1][1] - step[0][0]) / 1e6
print("one step: %d launches, span %.3f ms, busy %.3f ms" % (len(step), span, sum(e - s for s, e, _ in step) / 1e6))
phases, cur, name = [], [], "head (camera, rays, prior)"
for s, e, n in step:
    t = tag(n)
    if t:
        if cur:
            phases.append((name, cur))
        phases.append(("[%s]" % t, [(s, e, n)]))
        cur, name = [], "after " + t
    else:
        cur.append((s, e, n))
if cur:
    phases.append((name, cur))
# merge consecutive identical big phases (slabs / sampling stages) for the summary
for name, ks_ in phases:
    busy = sum(e - s for s, e, _ in ks_) / 1e6
    if name.startswith("["):
        print("%-42s %4d launch  %9.3f ms" % (name, len(ks_), busy))
        continue
    cnt = collections.Counter(n.split("(")[0][:60] for _, _, n in ks_)
    tim = collections.defaultdict(float)
    for s, e, n in ks_:
        tim[n.split("(")[0][:60]] += (e - s) / 1e6
    wall = (ks_[-1][1] - ks_[0][0]) / 1e6
    print("%-42s %4d launches %8.3f ms busy %8.3f ms wall" % (name, len(ks_), busy, wall))
    for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:8]:
        print("        %-62s x%-4d %7.3f ms" % (k, cnt[k], v))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        for s, e, n in step:
            fh.write("%10.3f %8.1f  %s\n" % ((s - step[0][0]) / 1e3, (e - s) / 1e3, n[:110]))
