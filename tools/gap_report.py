#!/usr/bin/env python
"""Where is the GPU idle?  From a rocprofv3 results database: the gaps between consecutive kernels (start of a dispatch
minus the latest end seen so far), summed by the pair (kernel before the gap -> kernel after it).
usage: python tools/gap_report.py DB [N] [--region LINE.json]"""
import collections
import json
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 15
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(con.execute(f"select {name}, start, end from kernels order by start"))
    if "--region" in sys.argv:
        line = json.loads(open(sys.argv[sys.argv.index("--region") + 1]).read().strip().splitlines()[-1])
        for clock, (lo, hi) in line.get("timed_region_clock_ns", {}).items():
            if lo < rows[-1][2] and hi > rows[0][1]:  # the clock whose window overlaps the trace is the profiler's
                rows = [r for r in rows if lo <= r[1] <= hi]
                break
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    acc = collections.defaultdict(lambda: [0.0, 0])
    end, prev, idle = rows[0][2], rows[0][0], 0.0
    for n, s, e in rows[1:]:
        if s > end:
            a = acc[(short(prev), short(n))]
            a[0] += s - end
            a[1] += 1
            idle += s - end
        if e > end:
            end, prev = e, n
    # the dispatches around the largest gap
    best, at, end2 = 0, 1, rows[0][2]
    for i, (n, s0, e) in enumerate(rows[1:], 1):
        if s0 - end2 > best:
            best, at = s0 - end2, i
        end2 = max(end2, e)
    span = rows[-1][2] - rows[0][1]
    print(f"# {db}: {len(rows)} dispatches over {span / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %)")
    for (a, b), (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{t / 1e6:8.2f} ms in {c:5d} gaps ({t / c / 1e3:7.1f} us each)  {a}  ->  {b}")
    print(f"# around the largest gap ({best / 1e3:.1f} us; start relative to the dispatch behind it, duration, kernel):")
    context(rows, at)


def context(rows, at, width=5):
    t0 = rows[at][1]
    for n, s0, e in rows[max(0, at - width):at + width]:
        print(f"   {(s0 - t0) / 1e3:10.1f} us  {(e - s0) / 1e3:9.1f} us  {n[:120]}")


if __name__ == "__main__":
    main()
