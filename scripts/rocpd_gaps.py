"""List the largest idle gaps between consecutive kernel dispatches of a rocprofv3 rocpd database (development aid)."""
import glob, os, sqlite3, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
    gaps.append(((s1 - e0) / 1e6, n0[:50], n1[:50], (e0 - s0) / 1e6))
tot = (rows[-1][1] - rows[0][0]) / 1e6
busy = sum((e - s) for s, e, _ in rows) / 1e6
print("span %.1f ms, kernel time %.1f ms, idle %.1f ms over %d dispatches" % (tot, busy, tot - busy, len(rows)))
for g in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("gap %8.3f ms  after %-50s (%.3f ms)  before %s" % (g[0], g[1], g[3], g[2]))
# steady state: the window of the last K dispatches of the dominant kernel
big = [i for i, r in enumerate(rows) if "mlp_bwd_kernel" in r[2]]
if len(big) >= 3:
    i0, i1 = big[-3], big[-1]
    sub = rows[i0:i1 + 1]
    span = (sub[-1][0] - sub[0][0]) / 1e6
    busy = sum(min(e, sub[-1][0]) - s for s, e, _ in sub[:-1]) / 1e6
    gs = [(b[0] - a[1]) / 1e6 for a, b in zip(sub[:-1], sub[1:])]
    print("steady state over 2 steps: span %.1f ms, busy %.1f ms, idle %.1f ms per step; gaps >0.2ms: %d (%.1f ms), 0.02-0.2ms: %d (%.1f ms), <0.02: %d (%.1f ms)" % (
        span, busy, (span - busy) / 2, sum(g > 0.2 for g in gs), sum(g for g in gs if g > 0.2), sum(0.02 < g <= 0.2 for g in gs),
        sum(g for g in gs if 0.02 < g <= 0.2), sum(g <= 0.02 for g in gs), sum(g for g in gs if g <= 0.02)))
    for g, a, b in sorted(((b[0] - a[1]) / 1e6, a[2][:44], b[2][:44]) for a, b in zip(sub[:-1], sub[1:]))[-12:]:
        print("   %.3f ms  %s -> %s" % (g, a, b))
