#!/usr/bin/env python
"""Per-kernel mean of one rocprofv3 --pmc counter (CSV).  usage: pmc_summary.py DIR [substr ...]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
keys = sys.argv[2:]
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
cname = None
for r in csv.DictReader(open(f)):
    cname = r["Counter_Name"]
    a = acc[r["Kernel_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"# {f}  counter={cname}")
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    if keys and not any(x in k for x in keys):
        continue
    print(f"{k[:80]:80s} dispatches={n:6d} mean={s / n:14.1f} total={s:16.1f}")
