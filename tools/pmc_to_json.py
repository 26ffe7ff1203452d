#!/usr/bin/env python
"""profiles/r0N_pmc_summary.json from two rocprofv3 PMC passes (separate runs, --pmc FETCH_SIZE and --pmc
WRITE_SIZE, --output-format csv, no tracing -- the combination gpurun allows).  Per kernel: mean KB per dispatch
of both counters and the HBM bytes per launch bench.py reports as roofline.traffic.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies a wide coalesced streaming read
(16 bytes per lane) at exactly half its bytes; kernels listed in DOUBLE_FETCH read that way and get x2.
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.

usage: python tools/pmc_to_json.py FETCH_DIR WRITE_DIR "command that was profiled" [out.json [kernel,kernel,...]]
(with a kernel list, only those entries are written and the other entries of an existing out.json are kept)"""
import collections
import csv
import glob
import json
import os
import sys

DOUBLE_FETCH = ("trd_column_kernel", "gram128_kernel", "gram64_stream_kernel", "proj64_stream_kernel")  # 16 B / lane streaming loads (double2 / float4)
KEEP = ("gram128_kernel", "gram128_reduce_kernel", "trd_team_kernel", "trd_column_kernel", "trd_tail_kernel", "trd_invit_kernel", "trd_back_kernel", "trd_bisect_kernel",
        "gram_wide_kernel", "gram64_stream_kernel", "proj64_stream_kernel", "tile_reduce_batched_kernel", "encode_tiled_kernel", "decode_tiled_kernel", "gemm_kernel", "gemm_bf16_kernel")


def means(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: (s / n, n) for k, (s, n) in acc.items()}


def main():
    fetch, write = means(sys.argv[1]), means(sys.argv[2])
    out = {}
    for key in KEEP:
        names = [k for k in fetch if key in k]
        if not names:
            continue
        name = max(names, key=lambda k: fetch[k][1])
        f_kb, n = fetch[name]
        w_kb = write.get(name, (0.0, 0))[0]
        corr = 2.0 if key in DOUBLE_FETCH else 1.0
        out[key] = {
            "kernel": name[:120],
            "dispatches": n,
            "fetch_kb_per_launch_raw": f_kb,
            "fetch_correction": corr,
            "write_kb_per_launch": w_kb,
            "hbm_bytes_per_launch": (corr * f_kb + w_kb) * 1024.0,
            "command": sys.argv[3],
        }
    path = sys.argv[4] if len(sys.argv) > 4 else "profiles/r02_pmc_summary.json"
    only = sys.argv[5].split(",") if len(sys.argv) > 5 else None  # keep other kernels' entries of an existing file
    if only is not None:
        out = {k: v for k, v in out.items() if k in only}
        if os.path.exists(path):
            with open(path) as fh:
                merged = json.load(fh)
            merged.update(out)
            out = merged
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
