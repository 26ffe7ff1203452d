#!/usr/bin/env python
"""Which kernel classes are in flight, and for how long per step, in the timed loop of bench.py?  From a rocprofv3
results database (`rocprofv3 --kernel-trace --stats -- python3 bench.py --skip-single --no-cpu-baseline`).
Classes: T = resident tridiagonalisation (trd_team_kernel), G = Gram (gram128_kernel), S = the solver's other kernels
(tail, bisect, inverse iteration, orthonormalisation, back-transformation, bookkeeping), M = GEMM tile kernels,
o = everything else of the library; the turn-taking spin kernels are ignored.  The steady-state window runs from
the third to the last but one raw Gram launch of the loop.
usage: python tools/phase_time.py gpurun_out/prof_xxx/bench_results.db"""
import collections
import sqlite3
import sys

SOLVER = ("trd_tail", "trd_bisect", "trd_invit", "trd_ortho", "trd_wy", "trd_back", "trd_rank", "trd_status", "trd_load",
          "trd_setdesc", "trd_setk")


def cls(name):
    if "trd_team" in name:
        return "T"
    if "gram128_kernel" in name:
        return "G"
    if "turn_" in name or "spin" in name:
        return "W"
    if any(k in name for k in SOLVER):
        return "S"
    if "gemm" in name:
        return "M"
    return "o"


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = [r for r in con.execute(f"select {name}, start, end from kernels order by start") if "at::native" not in r[0]]
    raw = [r for r in rows if "gram128_kernel<float, 2>" in r[0]]  # two raw Gram launches per step (one per group)
    lo, hi = raw[4][1], raw[-2][1]
    steps = (len(raw) - 6) / 2.0
    events = []
    for n, s, e in rows:
        s, e = max(s, lo), min(e, hi)
        if e > s:
            events += [(s, 1, cls(n)), (e, -1, cls(n))]
    events.sort()
    active, acc, last = collections.Counter(), collections.Counter(), lo
    for t, d, c in events:
        if t > last:
            key = "".join(sorted(k for k in active if active[k] > 0 and k != "W"))
            acc[key or "-"] += t - last
            last = t
        active[c] += d
    total = hi - lo
    print(f"# {sys.argv[1]}: window {total / 1e6:.1f} ms = {steps:.1f} steps, {total / 1e6 / steps:.2f} ms per step")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {k:6s} {v / 1e6 / steps:7.2f} ms per step  {100.0 * v / total:5.1f} %")
    for c in "TGSM":
        print(f"  any {c}: {sum(v for k, v in acc.items() if c in k) / 1e6 / steps:6.2f} ms per step")


if __name__ == "__main__":
    main()
