#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as text for profiles/.

usage: tools/rocprof_summary.py <results.db> [title]   -> prints per-kernel calls / total / average / min / max (us)
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {title}")
    print("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite 'kernels' view); durations in microseconds")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'min_us':>12} {'max_us':>12} {'pct':>6} {'vgpr':>5} {'sgpr':>5} {'lds':>6} {'grid':>8} {'wg':>4}  kernel")
    for name, n, tot, avg, mn, mx, vg, sg, lds, grid, wg in rows:
        print(f"{n:6d} {tot/1e3:12.3f} {avg/1e3:12.3f} {mn/1e3:12.3f} {mx/1e3:12.3f} {100*tot/total:6.2f} {vg or 0:5d} {sg or 0:5d} {lds or 0:6d} {grid or 0:8d} {wg or 0:4d}  {name}")


if __name__ == "__main__":
    main()
