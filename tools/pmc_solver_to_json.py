#!/usr/bin/env python
"""profiles/r0N_pmc_summary.json entries for the latency-bound solver kernels from ONE rocprofv3 PMC pass
(--pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
--output-format csv, no tracing: the combination gpurun allows).  Per kernel: mean counter values per dispatch and
what they say -- vector instructions per wave, the share of a wave's cycles spent parked (SQ_WAIT_ANY: s_waitcnt /
barrier) and stalled at issue (SQ_WAIT_INST_ANY), GPU-active cycles per dispatch (GRBM_GUI_ACTIVE is summed over the
8 XCDs: / 8 / 2.4 GHz ~ the dispatch's wall time in us at full clock).

usage: python tools/pmc_solver_to_json.py PMC_DIR "command that was profiled" out.json [kernel,kernel,...] [suffix]
(entries of an existing out.json for other kernels are kept; `suffix` is appended to the keys, e.g. " @ order 2048",
so that one kernel profiled on two workloads keeps both entries)"""
import collections
import csv
import glob
import json
import os
import sys

KEEP = ("trd_team_kernel", "trd_tail_kernel", "trd_tail_reg_kernel", "trd_bisect_kernel", "trd_invit_kernel", "trd_ortho_kernel",
        "trd_ortho_blocks_kernel", "trd_back_kernel", "trd_column_kernel", "pnl_vec_kernel", "pnl_symv_kernel", "pnl_update_kernel",
        "wide_trsm_kernel", "wide_chol_diag_kernel", "wide_chol_trail_kernel", "back_wide_kernel", "blk_step_kernel",
        "back_rows_step_kernel", "back_wide_t_kernel")


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    path = sys.argv[3]
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    suffix = sys.argv[5] if len(sys.argv) > 5 else ""
    out = {}
    if os.path.exists(path):
        with open(path) as fh:
            out = json.load(fh)
    for key in KEEP:
        if only is not None and key not in only:
            continue
        names = [k for k in acc if key + "<" in k or key + "(" in k or k.endswith(key)]
        if not names:
            continue
        name = max(names, key=lambda k: max(v[1] for v in acc[k].values()))
        c = {cn: v[0] / v[1] for cn, v in acc[name].items()}
        n = max(v[1] for v in acc[name].values())
        e = {"kernel": name[:140], "dispatches": n, "counters_per_dispatch": c, "command": sys.argv[2]}
        waves = c.get("SQ_WAVES", 0.0)
        if waves:
            e["valu_instructions_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / waves
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["share_of_wave_cycles_parked_waitcnt_or_barrier"] = c.get("SQ_WAIT_ANY", 0.0) / wc
            e["share_of_wave_cycles_stalled_at_issue"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
        if "GRBM_GUI_ACTIVE" in c:
            e["gpu_active_us_per_dispatch_at_2.4GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / 2400.0
        out[key + suffix] = e
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("command", "kernel")} for k, v in out.items()}, indent=1)[:6000])


if __name__ == "__main__":
    main()
