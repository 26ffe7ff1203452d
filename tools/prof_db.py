#!/usr/bin/env python
"""Summarise a rocprofv3 results database (rocpd sqlite, the default output of rocprofv3 7.x): per-kernel
totals and, optionally, the timeline (durations and gaps) around the middle launch of one kernel.
usage: python tools/prof_db.py path/to/x_results.db [top_n] [--gaps KERNEL_SUBSTR [count]] [--region bench_line.json]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = con.execute(f"select {name}, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    if "--region" in sys.argv:
        # keep the dispatches that start inside bench.py's timed region (its JSON line carries the region on several host
        # clocks: the one whose window overlaps the trace is the profiler's)
        import json
        with open(sys.argv[sys.argv.index("--region") + 1]) as fh:
            line = json.loads(fh.read().strip().splitlines()[-1])
        lo_all, hi_all = rows[0][1], rows[-1][2]
        kept = None
        for clock, (a, b) in line.get("timed_region_clock_ns", {}).items():
            if a < hi_all and b > lo_all:
                kept = [r for r in rows if a <= r[1] <= b]
                print(f"# dispatches inside the timed region ({clock}: {(b - a) / 1e6:.1f} ms): {len(kept)} of {len(rows)}")
                break
        if kept is None:
            print("# no clock of the timed region overlaps the trace: every dispatch is listed")
        else:
            rows = kept
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 25
    agg = {}
    for n, s, e, *_ in rows:
        a = agg.setdefault(n, [0, 0, 10 ** 18, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# {sys.argv[1]}\n# total kernel time {total / 1e6:.3f} ms, {len(rows)} dispatches")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{n[:70]:70s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} "
              f"{100 * a[1] / total:6.2f}")
    if "--gaps" in sys.argv:
        i = sys.argv.index("--gaps")
        key = sys.argv[i + 1]
        count = int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 12
        idx = [j for j, r in enumerate(rows) if key in r[0]]
        j0 = idx[len(idx) // 2]
        prev = None
        print(f"# timeline around the middle '{key}' launch")
        for n, s, e, gx, wx in rows[j0:j0 + count]:
            gap = "" if prev is None else f"{(s - prev) / 1e3:7.2f}"
            print(f"  {n[:50]:50s} dur_us={(e - s) / 1e3:8.2f} gap_us={gap:>8s} grid={gx} wg={wx}")
            prev = e


if __name__ == "__main__":
    main()
