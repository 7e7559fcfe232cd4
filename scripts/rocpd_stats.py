"""Per-kernel statistics from a rocprofv3 (rocpd sqlite) kernel trace.

    python scripts/rocpd_stats.py gpurun_out/prof/xxx_results.db > profiles/xxx_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    kcols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else "display_name"
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, cnt, s, a, mn, mx in rows:
        print(f"{n[:70]:70s} {cnt:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
    mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
    if mc:
        try:
            r = list(c.execute(f"select count(*), sum(end-start), sum(size) from {mc[0]}"))[0]
            if r[0]:
                print(f"\nmemory copies: {r[0]} calls, {r[1]/1e6:.3f} ms total, {r[2]/1e6:.1f} MB")
        except Exception as e:  # schema differences
            print("memory copy table:", e)


if __name__ == "__main__":
    main(sys.argv[1])
