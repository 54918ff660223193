"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / average duration."""
import glob, os, sqlite3, sys
def main(path, limit=30):
    if os.path.isdir(path):   # a rocprofv3 -d directory: take the (first) database below it
        path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc limit {limit}"
    rows = list(c.execute(q))
    tot = sum(r[2] for r in c.execute(f"select 0,0,sum(end-start)/1e6 from {kd}"))
    print("%-70s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for r in rows:
        print("%-70s %7d %12.3f %12.1f %12.1f %12.1f %6.1f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    print("total kernel time %.3f ms" % tot)
if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
