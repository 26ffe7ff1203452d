#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats output directory (CSV) as text.
usage: python tools/prof_summary.py gpurun_out/prof_xxx [top_n] [--gaps KERNEL_SUBSTR]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 20
    stats = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {stats}\n# total kernel time {total / 1e6:.3f} ms")
    print(f"{'kernel':72s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows[:top]:
        print(f"{r['Name'][:72]:72s} {r['Calls']:>8s} {float(r['TotalDurationNs']) / 1e6:10.3f} "
              f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} "
              f"{float(r['Percentage']):6.2f}")
    if "--gaps" in sys.argv:
        key = sys.argv[sys.argv.index("--gaps") + 1]
        trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
        t = list(csv.DictReader(open(trace)))
        t.sort(key=lambda r: int(r["Start_Timestamp"]))
        idx = [i for i, r in enumerate(t) if key in r["Kernel_Name"]]
        i0 = idx[len(idx) // 2]
        prev = None
        print(f"# timeline around the middle '{key}' launch")
        for r in t[i0:i0 + 12]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = "" if prev is None else f"{(s - prev) / 1e3:7.2f}"
            print(f"  {r['Kernel_Name'][:48]:48s} dur_us={(e - s) / 1e3:8.2f} gap_us={gap} grid={r.get('Grid_Size_X', '')} wg={r.get('Workgroup_Size_X', '')}")
            prev = e


if __name__ == "__main__":
    main()
