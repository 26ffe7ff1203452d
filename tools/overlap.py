#!/usr/bin/env python
"""How do the two streams of a bench run share the GPU?  From a rocprofv3 results database: time with 0 / 1 / 2+
kernels in flight, and the overlapped time per pair of kernel classes.
usage: python tools/overlap.py path/to/results.db [t_begin_ms t_end_ms]"""
import collections
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:34]


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    qs = collections.Counter(r[3] for r in rows)
    work = [q for q, c in qs.items() if c > 300]  # the group streams (the default stream generates the volumes)
    rows = [r for r in rows if r[3] in work and "at::native" not in r[0]]
    t0 = rows[0][1]
    team = [r for r in rows if "gram128" in r[0] or "gram_wide" in r[0]]
    if team and len(sys.argv) <= 3:  # default window: from the first big Gram launch (volume generation precedes it)
        rows = [r for r in rows if r[1] >= team[0][1]]
    lo = float(sys.argv[2]) * 1e6 + t0 if len(sys.argv) > 3 else rows[0][1]
    hi = float(sys.argv[3]) * 1e6 + t0 if len(sys.argv) > 3 else max(r[2] for r in rows)
    ev = []
    for n, s, e, q in rows:
        s, e = max(s, lo), min(e, hi)
        if e > s:
            ev.append((s, 1, short(n), q))
            ev.append((e, -1, short(n), q))
    ev.sort()
    active = collections.Counter()
    depth_time = collections.Counter()
    pair_time = collections.Counter()
    solo_time = collections.Counter()
    prev = lo
    for t, d, n, q in ev:
        dt = t - prev
        if dt > 0:
            live = [k for k, c in active.items() if c > 0]
            depth_time[min(len(live), 3)] += dt
            if len(live) == 1:
                solo_time[live[0][0]] += dt
            elif len(live) >= 2:
                names = sorted(k[0] for k in live)
                pair_time[(names[0], names[1])] += dt
        active[(n, q)] += d
        prev = t
    total = hi - lo
    print(f"window {total / 1e6:.1f} ms, streams {work}")
    for k in sorted(depth_time):
        print(f"  {k} kernel(s) in flight: {depth_time[k] / 1e6:8.2f} ms  {100 * depth_time[k] / total:5.1f} %")
    print(f"  idle: {(total - sum(depth_time.values())) / 1e6:8.2f} ms")
    print("solo (one stream busy):")
    for k, v in solo_time.most_common(12):
        print(f"  {v / 1e6:8.2f} ms  {k}")
    print("overlapped pairs:")
    for k, v in pair_time.most_common(14):
        print(f"  {v / 1e6:8.2f} ms  {k[0]} + {k[1]}")


if __name__ == "__main__":
    main()
