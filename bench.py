#!/usr/bin/env python
"""Benchmark of the hot path: Mvoxels/s of compress + reconstruct (NDMPS.from_tensor with the bond cap applied in
the sweep, then NDMPS.to_tensor), input and output resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--config metric|2|3|4|5] [--total-volumes V]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`; `metric` = the configuration its headline metric is quoted on, the default):

    metric  batch of 64 x 256^3 fp32, chi = 64 per GPU and step                       weak scaling
    2       batch of 64 x 256^3 fp32, chi = 32 per GPU and step                       weak scaling
    3       batch of  8 x 512^3 fp32, DCT mode, chi = 64 per GPU and step             weak scaling
    4       64 x 128^3 fp32, chi = 32 IN TOTAL, sharded over the ranks (8 per GPU at N = 8)   strong scaling
    5       ONE 128 x 128 x 64 x 256 tensor, chi = 128: bf16 storage on one GPU; rows sharded over the ranks
            (core/sharded.py, fp32 storage, one all-reduce of a Gram matrix per site) at N > 1       strong scaling

`--total-volumes V` turns any of the batch workloads into a strong-scaling run: V volumes in total, V / N per rank.

One process per GPU; a step is one pass of the hot path over the rank's volumes (NDMPS.from_tensors in lockstep
groups, NDMPS.to_tensors); independent volumes shard over the ranks with no data-path collective; the
only collective before the timed region is the RCCL broadcast of the job descriptor.  Rank 0 prints one JSON line.
At N = 1 it also carries the CPU baseline (the NumPy oracle on a bounded sample of the workload, timed on the host
cores) and the parity of the GPU path against the oracle on that sample.

The rank / shard / timing logic lives in functions (job_descriptor, volume_seeds, timed_steps, throughput) that
tests/test_batch_sharding.py drives at world size 2 under gloo with a stub step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
F64_MFMA_PEAK_TFLOPS = 78.6  # dense v_mfma_f64_16x16x4_f64: 2048 flop / 64 clk per SIMD at 2.4 GHz (measured 77.7)
METRIC = "Mvoxels/s compress+reconstruct, 256^3 fp32, bond chi=64; SSIM vs ref"
FIRST_SEED = 2025  # the reference tests' seed; volume j of the job uses FIRST_SEED + j

# BASELINE.json configs[1..4] and the metric's own configuration; flags given on the command line override a field
CONFIGS = {
    "metric": {"kind": "cubes", "size": 256, "chi": 64, "mode": "Std", "batch": 64, "groups": 0, "total_volumes": 0},
    "2": {"kind": "cubes", "size": 256, "chi": 32, "mode": "Std", "batch": 64, "groups": 0, "total_volumes": 0},
    "3": {"kind": "cubes", "size": 512, "chi": 64, "mode": "DCT", "batch": 8, "groups": 1, "total_volumes": 0},
    "4": {"kind": "cubes", "size": 128, "chi": 32, "mode": "Std", "batch": 64, "groups": 1, "total_volumes": 64},
    "5": {"kind": "tensor", "shape": (128, 128, 64, 256), "chi": 128, "mode": "Std", "batch": 1, "groups": 1,
          "total_volumes": 1, "size": 0},
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS),
                    help="BASELINE.json workload (see the module docstring); the flags below override its fields")
    ap.add_argument("--size", type=int, default=None, help="edge of the cubic volume")
    ap.add_argument("--chi", type=int, default=None)
    ap.add_argument("--mode", default=None, choices=["Std", "DCT"])
    ap.add_argument("--batch", type=int, default=None, help="independent volumes per GPU per step (weak scaling)")
    ap.add_argument("--total-volumes", type=int, default=None,
                    help="strong scaling: this many volumes IN TOTAL per step, sharded over the ranks")
    ap.add_argument("--groups", type=int, default=None,
                    help="concurrent groups (host thread + HIP stream each) the rank's volumes are cut into; "
                         "volumes of a group are encoded in lockstep (0 / not given: two from 32 volumes per GPU on, else one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="profiling aid: the timed loop only")
    ap.add_argument("--lanes", type=int, default=0,
                    help="cube configs: sets of streams that consecutive batches alternate between (0: core/batch.py "
                         "default_lanes -- three when a batch is one lockstep group, two for two groups of eigenproblems of "
                         "order <= 256, else one)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="cube configs: build the objects of a batch before the next batch is enqueued (A/B of the default)")
    ap.add_argument("--no-configs", action="store_true",
                    help="the default line carries short runs of BASELINE configs 2 .. 5 (`configs`); this leaves them out")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses N ranks on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    # an explicit per-GPU batch without --total-volumes asks for weak scaling even on a strong-scaling config
    if args.batch is not None and args.total_volumes is None:
        args.total_volumes = 0
    for key in ("size", "chi", "mode", "batch", "groups", "total_volumes"):
        if getattr(args, key) is None:
            setattr(args, key, cfg[key])
    args.kind = cfg["kind"]
    args.shape = tuple(cfg.get("shape", (args.size,) * 3))
    return args


# ------------------------------------------------------------------ rank / shard / timing logic (GPU-free)
def job_descriptor(args, world):
    """What rank 0 broadcasts: everything that decides which volumes exist and how they are processed.
    Weak scaling (the default): every rank owns `batch` volumes.  Strong scaling (--total-volumes, configs 4
    and 5): the job is `total_volumes` volumes whatever the number of ranks."""
    strong = int(args.total_volumes) > 0
    n_volumes = int(args.total_volumes) if strong else int(args.batch) * int(world)
    per_gpu = -(-n_volumes // int(world)) if strong else int(args.batch)
    groups = int(args.groups)
    if groups <= 0:  # not given: what the library would pick for this many volumes per GPU
        from imgcompressionmps_amd.core.batch import default_groups, default_stream_shape
        if getattr(args, "kind", "cubes") == "cubes" and not getattr(args, "no_pipeline", False):
            groups = default_stream_shape(per_gpu, min(8 * int(args.chi), 512))[0]  # batch after batch (run_cubes)
        else:
            groups = default_groups(per_gpu)
    return {"config": str(getattr(args, "config", "metric")), "kind": str(getattr(args, "kind", "cubes")),
            "shape": [int(v) for v in getattr(args, "shape", (args.size,) * 3)],
            "size": int(args.size), "chi": int(args.chi), "mode": str(args.mode),
            "batch_per_gpu": -(-n_volumes // int(world)) if strong else int(args.batch),
            "groups": groups, "world": int(world), "first_seed": FIRST_SEED,
            "n_volumes": n_volumes, "scaling": "strong" if strong else "weak"}


def volume_seeds(job, rank):
    """Seeds of the volumes rank `rank` owns: the job's volumes are numbered 0 .. n_volumes-1 with seeds
    first_seed + j (2025 .. 2088 for the 64 volumes of BASELINE's batch config); ranks own contiguous blocks
    (core/batch.shard_indices), so every volume of the job is distinct and owned exactly once."""
    from imgcompressionmps_amd.core.batch import shard_indices

    return [job["first_seed"] + j for j in shard_indices(job["n_volumes"], rank, job["world"])]


def timed_steps(step, steps, warmup, barrier, reduce_max, after_warmup=None, drain=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier() on both sides (barrier = process
    group barrier + device synchronise); returns the MAX over ranks of the elapsed seconds.  drain: for steps that
    leave host work pending (the objects of the last batch, see run_cubes): run in front of BOTH barriers, so the timed
    region holds the host work of exactly K steps."""
    for _ in range(warmup):
        step()
    if drain is not None:
        drain()
    if after_warmup is not None:
        after_warmup()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if drain is not None:
        drain()
    barrier()
    return float(reduce_max(time.perf_counter() - t0))


def throughput(job, n_vox, steps, elapsed):
    """Whole-job Mvoxels/s: all ranks together processed n_volumes volumes per step (weak scaling: world x
    batch_per_gpu; strong scaling: the fixed total)."""
    return job["n_volumes"] * n_vox * steps / elapsed / 1e6


# ------------------------------------------------------------------------------------------ data
def synthetic_mri_device(shape, seed, device):
    """SURVEY 8(d)'s synthetic MRI volume generated on the device: the oracle generator's recipe and its
    parameter draws (oracle/metrics.py synthetic_mri: same blobs and shell for the same seed; 4-D = the 3-D phantom
    times a smooth temporal modulation), evaluated with torch; the white noise comes from a torch generator with
    that seed (NumPy's stream is not reproduced)."""
    import torch

    rng = np.random.default_rng(seed)
    sp = tuple(shape[:3])
    axes = [torch.linspace(-1.0, 1.0, n, dtype=torch.float64, device=device) for n in sp]
    vol = torch.zeros(sp, dtype=torch.float64, device=device)
    for _ in range(12):
        amp = rng.uniform(0.3, 1.0)
        fac = []
        for a in axes:
            c = rng.uniform(-0.6, 0.6)
            w = rng.uniform(0.08, 0.45)
            fac.append(torch.exp(-0.5 * ((a - c) / w) ** 2))
        vol += amp * fac[0][:, None, None] * fac[1][None, :, None] * fac[2][None, None, :]
    r2 = torch.zeros(sp, dtype=torch.float64, device=device)
    for j, a in enumerate(axes):
        e = (a / rng.uniform(0.75, 0.95)) ** 2
        r2 += e.reshape((1,) * j + (-1,) + (1,) * (2 - j))
    vol += 0.8 * torch.exp(-0.5 * ((torch.sqrt(r2) - 1.0) / 0.04) ** 2)
    gen = torch.Generator(device=device).manual_seed(int(seed))
    if len(shape) == 4:
        t = torch.linspace(0.0, 1.0, shape[3], dtype=torch.float64, device=device)
        mod = 1.0 + 0.25 * torch.sin(2 * np.pi * (t * rng.uniform(1, 3) + rng.uniform()))
        vol = (vol[..., None] * mod).to(torch.float32)  # 4-D: fp32 from here on (1 GiB at BASELINE's config 5)
        vol += 0.01 * torch.randn(vol.shape, dtype=torch.float32, device=device, generator=gen)
    else:
        vol += 0.01 * torch.randn(sp, dtype=torch.float64, device=device, generator=gen)
    vol -= vol.min()
    vol /= vol.max()
    return vol.to(torch.float32)


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/r0*_pmc_summary.json, written
    by tools/pmc_to_json.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs; the newest round that has the kernel),
    or None."""
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                entry = json.load(f)[kernel_key]
            return float(entry["hbm_bytes_per_launch"]), entry
        except (OSError, KeyError, ValueError):
            continue
    return None, None


# ------------------------------------------------------------------------------------------ launch spans
SLOT_COLUMNS, SLOT_TEAM, SLOT_GRAM, SLOT_GRAM_SMALL, SLOT_PANEL, SLOT_TAIL, SLOT_VECTORS = 1, 2, 3, 4, 5, 6, 7
SLOT_INFO = {
    SLOT_COLUMNS: ("trd_column_kernel (one launch per column of the Householder tridiagonalisation, orders 513 .. 1535: "
                   "the trailing matrices read and written once per launch)", "hbm"),
    SLOT_TEAM: ("trd_team_kernel (register-resident Householder tridiagonalisation of a lockstep group's matrices, one "
                "launch for all columns; latency-bound: one exchange between the workgroups of a matrix per column)", "latency"),
    SLOT_GRAM: ("gram128_kernel, the launches that fill the GPU for milliseconds and take the device-side turn (fp64 Gram "
                "matrices A^T A of a lockstep group's raw unfoldings in one launch, v_mfma_f64_16x16x4_f64)", "mfma"),
    SLOT_GRAM_SMALL: ("gram128_kernel / gram_wide_kernel, the Gram launches of the later sites (fp64 MFMA)", "mfma"),
    SLOT_PANEL: ("pnl_vec_kernel + pnl_symv_kernel + pnl_update_kernel (panel-blocked Householder tridiagonalisation, orders "
                 "from 1536 on: two dependent launches per column, the lower tiles of the trailing matrix read once per "
                 "column; each launch costs ~5 us whatever it moves)", "hbm"),
    SLOT_TAIL: ("trd_tail_kernel + trd_bisect_kernel (last 128 columns of the reduction in LDS, eigenvalues by "
                "65-section: one or a few workgroups per matrix)", "latency"),
    SLOT_VECTORS: ("trd_invit_kernel + trd_ortho_*_kernel + trd_wy_kernel + trd_back_kernel (eigenvectors of T by inverse "
                   "iteration, Cholesky-QR, back-transformation: one or a few workgroups per matrix)", "latency"),
}
BOUNDED_SLOTS = (SLOT_COLUMNS, SLOT_GRAM, SLOT_GRAM_SMALL, SLOT_PANEL)


def collect_slots(lib):
    import ctypes as C

    from imgcompressionmps_amd import _lib

    out = {}
    for slot in SLOT_INFO:
        ms, launches, amount = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(lib.ndmps_profile_collect(slot, C.byref(ms), C.byref(launches), C.byref(amount)))
        out[slot] = (ms.value, launches.value, amount.value)
    return out


def roofline_of(slots, steps, eig_order=None):
    """The roofline object.  `roofline` itself prices the instrumented kernel with the most device time in the timed
    region (HIP events on the launching streams around every launch, csrc/util.hip) AMONG THOSE A ROOFLINE BOUNDS -- the
    MFMA-bound Gram launches, the HBM-bound column / panel launches.  `dominant` names the instrumented kernel class
    with the most device time OF ALL, whatever bounds it: in the batch configurations that is the latency-bound resident
    tridiagonalisation, priced against HBM (2 x 8 n^2 bytes per matrix: the matrix in, the reflectors out) and against
    the fp64 vector peak (4/3 n^3 flops per matrix) so that its distance from any roofline is a number.  The
    end-to-end figure SURVEY 8(d) names as the headline, 2 s N / t_total, is `end_to_end_frac` (filled in by the
    caller)."""
    busy = {s: v for s, v in slots.items() if v[1] > 0}

    def line(slot):
        ms, launches, amount = busy[slot]
        name, bound = SLOT_INFO[slot]
        per_us = ms * 1e3 / launches
        per_amount = amount / launches
        out = {"kernel": name, "bound": bound, "launch_us": per_us, "launches_per_step": launches / steps,
               "device_ms_per_step": ms / steps}
        if bound == "mfma":
            achieved = per_amount / (per_us * 1e-6) / 1e12
            out.update({"achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F64_MFMA_PEAK_TFLOPS,
                        "flops_per_launch": per_amount})
        elif amount > 0:
            achieved = per_amount / (per_us * 1e-6) / 1e9
            out.update({"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_launch": per_amount})
        if slot == SLOT_TEAM:
            out["bound"] = "latency of the per-column exchange (no trailing-matrix traffic; priced against HBM and fp64 for scale)"
            out["span_covers"] = "the launch and its wait for the device-side turn"
            if eig_order:
                flops = per_amount / 16.0 * (4.0 / 3.0) * eig_order  # bytes = 16 n^2 per matrix
                out["fp64_vector_frac"] = flops / (per_us * 1e-6) / 1e12 / F64_MFMA_PEAK_TFLOPS
        return out

    bounded = {s: v for s, v in busy.items() if s in BOUNDED_SLOTS}
    if not bounded:
        roof = {"kernel": None, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None,
                "note": "no MFMA- or HBM-bound instrumented kernel ran in the timed region"}
        dom = None
    else:
        dom = max(bounded, key=lambda s: bounded[s][0])
        roof = line(dom)
    roof["traffic"] = None
    roof["measured"] = "HIP events on the launching streams around every launch of this kernel in the timed region"
    if dom == SLOT_GRAM:
        roof["traffic"], roof["traffic_source"] = pmc_traffic("gram128_kernel")
    if dom in (SLOT_GRAM, SLOT_GRAM_SMALL):
        roof["peak_source"] = ("AMD MI355X spec, FP64 matrix 78.6 TFLOP/s = 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz; "
                               "77.7 measured with tools/scratch/mfma_f64_rate.hip (MI355X_MICROARCH.md lists no f64 row)")
    roof["other_instrumented_kernels"] = [line(s) for s in sorted(busy) if s != dom]
    if busy:
        top = max(busy, key=lambda s: busy[s][0])
        d = line(top)
        d["is_the_roofline_kernel"] = bool(top == dom)
        d["share_of_instrumented_device_time"] = busy[top][0] / sum(v[0] for v in busy.values())
        d["note"] = ("the instrumented kernel class with the most device time in the timed region; spans of different "
                     "streams overlap, so the shares are of summed span time, not of the step")
        roof["dominant"] = d
    if SLOT_TEAM in busy:
        t = line(SLOT_TEAM)
        t["larger_than_the_roofline_kernel"] = bool(dom is None or busy[SLOT_TEAM][0] > busy[dom][0])
        roof["tridiagonalisation"] = t
    return roof


class ProfiledRegion:
    """The timed region on the host clocks a profiler stamps its kernels with: the bench line carries the window
    (`timed_region_clock_ns`) and tools/prof_db.py --region keeps the dispatches that start inside it, so the committed
    kernel summaries list the timed region and not the volume generator, the warm-up or the reference points.
    (rocprofv3 --selected-regions with roctxProfilerResume / Pause recorded nothing for launches from worker threads,
    and merely loading the roctx library made every eager launch slower: config 5, 8 000 launches per step, 66.5 -> 75 ms.)"""

    def __init__(self):
        self.lib = None

    def resume(self):
        self.t0 = self.clocks()
        if self.lib is not None:
            self.lib.roctxProfilerResume(0)

    def pause(self):
        self.t1 = self.clocks()
        if self.lib is not None:
            self.lib.roctxProfilerPause(0)

    @staticmethod
    def clocks():
        out = {}
        for name in ("CLOCK_MONOTONIC", "CLOCK_BOOTTIME", "CLOCK_REALTIME", "CLOCK_MONOTONIC_RAW"):
            if hasattr(time, name):
                out[name] = time.clock_gettime_ns(getattr(time, name))
        return out

    def window(self):
        """The timed region on the host clocks a profiler may stamp its kernels with: tools/prof_db.py --region keeps the
        dispatches that start inside it (rocprofv3 --selected-regions recorded nothing for launches from worker threads)."""
        t0, t1 = getattr(self, "t0", {}), getattr(self, "t1", {})
        return {k: [t0[k], t1[k]] for k in t0 if k in t1}


# ------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from imgcompressionmps_amd import NDMPS, _lib
    from imgcompressionmps_amd.core import batch as batch_mod
    from imgcompressionmps_amd.core import ndmps as ndmps_mod

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    lib = _lib.load()
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    job = job_descriptor(args, world)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
        # the one collective of the run: rank 0's descriptor is authoritative (RCCL broadcast over xGMI)
        job = batch_mod.broadcast_job(job if rank == 0 else None, src=0,
                                      device=device if args.backend == "nccl" else "cpu")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx = {"args": args, "job": job, "rank": rank, "world": world, "device": device, "lib": lib, "torch": torch,
           "dist": dist, "NDMPS": NDMPS, "_lib": _lib, "batch_mod": batch_mod, "ndmps_mod": ndmps_mod,
           "barrier": barrier, "reduce_max": reduce_max}
    line = run_tensor(ctx) if job["kind"] == "tensor" else run_cubes(ctx)
    if (rank == 0 and world == 1 and job["config"] == "metric" and not args_overridden(args) and not args.skip_single
            and not args.no_configs):
        line["configs"] = other_configs(ctx)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def other_configs(ctx):
    """Short runs (6 steps behind 2 warm-up steps) of BASELINE configs 2 .. 5 at their full sizes, in the same process and
    inside whatever clock is put around it: ms per step, throughput, the end-to-end fraction of the HBM roofline and the
    dominant instrumented kernel of each.  `python bench.py --config N` gives the full line of one of them."""
    import copy
    import gc

    torch = ctx["torch"]
    out = {}
    for name in ("2", "3", "4", "5"):
        a2 = copy.copy(ctx["args"])
        cfg = CONFIGS[name]
        a2.config, a2.kind = name, cfg["kind"]
        for key in ("size", "chi", "mode", "batch", "groups", "total_volumes"):
            setattr(a2, key, cfg[key])
        a2.shape = tuple(cfg.get("shape", (cfg["size"],) * 3))
        # two warm-up steps: steps are enqueued without a wait, and the caching allocator only reaches its steady state
        # once it holds the buffers of two steps in flight (one warm-up step left config 3's 4.3 GB of volumes per step
        # on fresh hipMallocs inside the timed region: 445 ms per step against 30)
        # six timed steps: with batches issued as a stream the objects of the last `lanes` batches are built at the end of the
        # timed region, which three steps would over-weigh (config 5: 33 ms per step over three, 28 over ten)
        a2.steps, a2.warmup, a2.skip_single, a2.no_cpu_baseline = 6, 2, True, True
        c2 = dict(ctx, args=a2, job=job_descriptor(a2, 1))
        t0 = time.perf_counter()
        full = run_tensor(c2) if cfg["kind"] == "tensor" else run_cubes(c2)
        roof = full["roofline"]
        dom = roof.get("dominant") or {}
        out[name] = {"workload": full["config"]["workload"], "scaling": full["scaling"], "dtype": full["dtype"],
                     "steps": a2.steps, "warmup": a2.warmup, "ms_per_step": full["ms_per_step"], "Mvoxels_per_s": full["value"],
                     "end_to_end_frac": roof.get("end_to_end_frac"), "bonds": full["config"].get("bonds"),
                     "roofline_kernel": {k: roof.get(k) for k in ("kernel", "bound", "frac", "device_ms_per_step")},
                     "dominant": {k: dom.get(k) for k in ("kernel", "bound", "frac", "device_ms_per_step",
                                                           "share_of_instrumented_device_time")},
                     "team_fallbacks": full.get("team_fallbacks"), "wall_s_with_setup": None}
        del full, c2
        gc.collect()
        torch.cuda.empty_cache()
        out[name]["wall_s_with_setup"] = time.perf_counter() - t0
    return out


def base_line(job, args, world, value, ms_per_step, dtype, workload, extra_config):
    shape = "x".join(str(v) for v in job["shape"])
    metric = METRIC if job["config"] == "metric" and not args_overridden(args) else (
        f"Mvoxels/s compress+reconstruct, {shape} {dtype}, {job['mode']} mode, bond chi={job['chi']}; SSIM vs ref")
    config = {"workload": workload, "baseline_config": job["config"]}
    config.update(extra_config)
    return {"metric": metric, "value": value, "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": job["scaling"],
            "vs_baseline": None, "dtype": dtype, "data": "synthetic", "config": config}


def args_overridden(args):
    cfg = CONFIGS[args.config]
    return any(getattr(args, k) != cfg[k] for k in ("size", "chi", "mode"))


def oracle_sample(shape, seed, chi, mode, x_host=None, bf16=False):
    """CPU baseline: the NumPy oracle on one sample volume, timed on the host cores; returns (reference object,
    its reconstruction, seconds, threads, the sample itself)."""
    from oracle.metrics import synthetic_mri
    from oracle.ndmps_oracle import OracleNDMPS

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    if x_host is None:
        x_host = synthetic_mri(shape, seed=seed)
    if bf16:
        import torch

        x_host = torch.from_numpy(x_host).to(torch.bfloat16).to(torch.float32).numpy()
    t0 = time.perf_counter()
    ref = OracleNDMPS.from_tensor(x_host, mode=mode, max_bond=chi, materialise_map=False)
    rec_ref = ref.to_tensor()
    return ref, rec_ref, time.perf_counter() - t0, int(threads), x_host


def parity_block(x_host, rec_gpu, rec_ref, bonds_gpu, bonds_ref):
    from oracle.metrics import compute_ssim_by_dim

    x64 = x_host.astype(np.float64)
    rec_gpu = np.asarray(rec_gpu, dtype=np.float64)
    ssim_gpu = float(compute_ssim_by_dim(x64, rec_gpu))
    ssim_ref = float(compute_ssim_by_dim(x64, rec_ref))
    return {"ssim_gpu": ssim_gpu, "ssim_oracle": ssim_ref, "ssim_gap": abs(ssim_gpu - ssim_ref),
            "rel_frobenius_vs_oracle": float(np.linalg.norm(rec_gpu - rec_ref) / np.linalg.norm(rec_ref)),
            "max_abs_vs_oracle": float(np.abs(rec_gpu - rec_ref).max()), "bonds_equal": list(bonds_gpu) == list(bonds_ref)}


# ------------------------------------------------------------------------------------------ batches of cubic volumes
def run_cubes(ctx):
    args, job, rank, world, device = ctx["args"], ctx["job"], ctx["rank"], ctx["world"], ctx["device"]
    torch, lib, NDMPS, _lib = ctx["torch"], ctx["lib"], ctx["NDMPS"], ctx["_lib"]
    batch_mod, ndmps_mod = ctx["batch_mod"], ctx["ndmps_mod"]
    from oracle.metrics import synthetic_mri  # the one host-generated volume (the one the oracle re-encodes)

    shape = tuple(job["shape"])
    n_vox = int(np.prod(shape))
    seeds = volume_seeds(job, rank)
    if not seeds:
        raise SystemExit(f"rank {rank} owns no volume: {job['n_volumes']} volumes over {world} ranks")
    # volume 0 of every rank comes from the host generator (it is the one the oracle re-encodes at N = 1 when it is
    # small enough); the others are generated on the device with the same recipe, one distinct seed each
    host_first = n_vox <= 256 ** 3
    x_host = synthetic_mri(shape, seed=seeds[0]) if host_first else None
    xs = [torch.from_numpy(x_host).to(device) if host_first else synthetic_mri_device(shape, seeds[0], device)]
    xs += [synthetic_mri_device(shape, sd, device) for sd in seeds[1:]]
    x = xs[0]
    groups = max(1, min(job["groups"], len(xs)))

    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(groups)
    last = {}

    pipelined = not args.no_pipeline
    # consecutive batches of ONE lockstep group each (fewer than 32 volumes per GPU: a rank of a strong-scaling run) alternate
    # between three sets of streams, so that they overlap on the GPU the way the two groups of a large batch do
    eig_order = min(8 * job["chi"], 512)
    lanes = max(1, args.lanes) if args.lanes else (
        batch_mod.default_stream_shape(len(xs), eig_order)[1] if int(args.groups) <= 0 else batch_mod.default_lanes(groups, eig_order))
    if not pipelined:
        lanes = 1
    in_flight, counter = [], [0]

    def collect(pending):
        objs, recs = pending.result()
        last["obj"], last["rec"] = objs[0], recs[0]

    def step():
        # The step returns when everything is enqueued; the next step queues behind it on the same streams (like the
        # steps of a training loop) and the barrier of the timed region synchronises the device.  Pipelined (the default):
        # batch k is ENQUEUED (encode_decode_begin), then the NDMPS objects of batch k - 1 are built -- under batch k's
        # kernels instead of in front of them; drain() below builds the last batch's inside the timed region.
        # --no-pipeline: the one-call form (objects of batch k built before batch k + 1 is enqueued).
        if not pipelined:
            objs, recs = batch_mod.encode_decode_concurrent(xs, groups=groups, mode=job["mode"], max_bond=job["chi"],
                                                             pool=pool, wait=False)
            last["obj"], last["rec"] = objs[0], recs[0]
            return
        in_flight.append(batch_mod.encode_decode_begin(xs, groups=groups, mode=job["mode"], max_bond=job["chi"], pool=pool,
                                                       lane=counter[0], lanes=lanes))
        counter[0] += 1
        while len(in_flight) > lanes:
            collect(in_flight.pop(0))

    def drain():
        while in_flight:
            collect(in_flight.pop(0))

    def group_step(vols):  # one lockstep group on the current stream
        objs = NDMPS.from_tensors(vols, mode=job["mode"], max_bond=job["chi"])
        return [o.to_tensor(as_torch=True) for o in objs]

    def single_step():
        o = NDMPS.from_tensor(x, mode=job["mode"], max_bond=job["chi"])
        return o, o.to_tensor(as_torch=True)

    timer = ndmps_mod.StageTimer()
    region = ProfiledRegion()

    def start_profiling():
        ndmps_mod.set_stage_timer(timer)
        _lib.check(lib.ndmps_profile_enable(1))
        torch.cuda.synchronize()
        region.resume()

    # set-up, not a step: every lane's streams get their buffers from the caching allocator (its pools are per stream, and a
    # lane holds two batches' reconstructions at a time: with W = 2 warm-up steps the third lane would meet fresh hipMallocs
    # of several GB inside the timed region)
    for _ in range(2 * lanes if lanes > 1 else 0):
        step()
    drain()
    torch.cuda.synchronize()
    elapsed = timed_steps(step, args.steps, args.warmup, ctx["barrier"], ctx["reduce_max"], after_warmup=start_profiling,
                          drain=drain)
    region.pause()
    ndmps_mod.set_stage_timer(None)
    _lib.check(lib.ndmps_profile_enable(0))
    obj, rec = last["obj"], last["rec"]
    stages = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps}
              for k, v in timer.totals_ms().items()}
    value = throughput(job, n_vox, args.steps, elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    roofline = roofline_of(collect_slots(lib), args.steps, eig_order=min(8 * job["chi"], 512))
    algo_bytes = 2 * 4 * n_vox  # read the volume once, write the reconstruction once
    roofline["end_to_end_algorithmic_GBps"] = len(xs) * algo_bytes / (ms_per_step * 1e-3) / 1e9
    roofline["end_to_end_frac"] = roofline["end_to_end_algorithmic_GBps"] / HBM_PEAK_GBPS

    line = base_line(
        job, args, world, value, ms_per_step, "f32",
        f"{job['n_volumes']} independent {job['size']}^3 fp32 synthetic MRI volumes per step "
        f"({'in total, sharded over' if job['scaling'] == 'strong' else 'i.e. ' + str(job['batch_per_gpu']) + ' on each of'} "
        f"{world} GPU(s); {groups} concurrent group(s) per GPU, lockstep inside a group), "
        f"NDMPS.from_tensors(max_bond={job['chi']}, mode={job['mode']}) + to_tensor each, device-resident in/out",
        {"volumes_per_step": job["n_volumes"], "batch_per_gpu": len(xs), "groups_per_gpu": groups,
         "host_pipeline": (f"batch k is enqueued (encode_decode_begin, lane k mod {lanes} of {lanes} set(s) of streams), then "
                           f"the NDMPS objects of batch k - {lanes} are built; the last batches' objects are built inside the "
                           "timed region, in front of the closing barrier"
                           + (f"; set-up ran two untimed batches per lane ({2 * lanes}) so that every lane's allocator pool is "
                              "filled" if lanes > 1 else "")
                           if pipelined else "none (--no-pipeline): objects of batch k built before batch k + 1 is enqueued"),
         "volume_source": f"every volume distinct: seeds {job['first_seed']}..{job['first_seed'] + job['n_volumes'] - 1}, "
                          "block-sharded over the ranks; volume 0 of a rank from the host generator (up to 256^3), the "
                          "rest from its device-side twin (same blobs / shell per seed, torch noise stream)",
         "bonds": obj.bond_sizes(),
         "parallelism": f"{world} independent volume shard(s), no data-path collective; job descriptor broadcast "
                        f"from rank 0" + (f" ({args.backend})" if world > 1 else " (single rank: none)")})
    line["roofline"] = roofline
    line["stages"] = stages
    line["team_fallbacks"] = int(lib.ndmps_syevd_topk_team_fallbacks())
    line["timed_region_clock_ns"] = region.window()

    if not args.skip_single and rank == 0:
        extras_cubes(ctx, line, xs, x, shape, n_vox, group_step, single_step, ms_per_step)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample: one volume of the workload up to 256^3; for 512^3 a 256^3 volume of the same recipe, mode
        # and bond cap (an eighth of the voxels: the oracle's SVD of a 262144 x 512 unfolding alone takes minutes)
        if host_first:
            s_shape, s_x, s_rec, s_obj = shape, x_host, rec, obj
            what = f"one {job['size']}^3 volume of the batch"
        else:
            s_shape = (256, 256, 256)
            s_x = synthetic_mri(s_shape, seed=seeds[0])
            s_obj = NDMPS.from_tensor(torch.from_numpy(s_x).to(device), mode=job["mode"], max_bond=job["chi"])
            s_rec = s_obj.to_tensor(as_torch=True)
            what = (f"one 256^3 volume (same generator, mode and bond cap; 1/{n_vox // 256 ** 3} of the voxels of a "
                    f"{job['size']}^3 volume, whose oracle sweep takes minutes)")
        ref, rec_ref, cpu_s, threads, _ = oracle_sample(s_shape, seeds[0], job["chi"], job["mode"], x_host=s_x)
        line["cpu_baseline"] = {
            "value": int(np.prod(s_shape)) / cpu_s / 1e6, "unit": "Mvoxels/s", "cores": threads, "kind": "port",
            "sample": f"{what}, NumPy fp64 oracle (closed-form index permutation, SVD sweep with max_bond={job['chi']}, "
                      f"mode {job['mode']}, chain contraction), {cpu_s:.1f} s wall"}
        line["parity"] = parity_block(s_x, s_rec.cpu().numpy(), rec_ref, s_obj.bond_sizes(), ref.bond_sizes())
        line["parity"]["sample"] = what
    pool.shutdown()
    return line


def extras_cubes(ctx, line, xs, x, shape, n_vox, group_step, single_step, ms_per_step):
    """Reference points next to `value` (same kernels, not part of it): the Gram launch alone, the stand-alone reshape
    stage, one lockstep group of 8, a single volume, the strong-scaling projection and the reference's literal flow."""
    import ctypes as C

    args, job, device = ctx["args"], ctx["job"], ctx["device"]
    torch, lib, NDMPS, _lib, ndmps_mod, batch_mod = (ctx["torch"], ctx["lib"], ctx["NDMPS"], ctx["_lib"], ctx["ndmps_mod"],
                                                     ctx["batch_mod"])
    roofline = line["roofline"]
    # the dominant Gram launch with nothing else on the GPU: one lockstep group's raw Gram (N / 512 rows, 512 columns)
    if job["size"] >= 64 and job["chi"] >= 64:
        nb, n_g = min(32, len(xs)), 512
        m_g = n_vox // n_g
        nbytes_g = lib.ndmps_gram_batched_workspace_bytes(nb, m_g, n_g)
        if nbytes_g > 0:
            gws = torch.empty(int(nbytes_g), dtype=torch.uint8, device=device)
            gout = torch.empty((nb, n_g, n_g), dtype=torch.float64, device=device)
            ptrs = (C.c_void_p * nb)(*[xs[j].data_ptr() for j in range(nb)])

            def gram_alone():
                _lib.check(lib.ndmps_gram_batched_f32(nb, ptrs, m_g, n_g, n_g, gout.data_ptr(), n_g * n_g, gws.data_ptr(),
                                                      nbytes_g, _lib.stream_ptr()))

            gram_alone()
            torch.cuda.synchronize()
            _lib.check(lib.ndmps_profile_enable(1))
            for _ in range(3):
                gram_alone()
            torch.cuda.synchronize()
            _lib.check(lib.ndmps_profile_enable(0))
            slots = collect_slots(lib)
            i_ms, i_launches, i_flops = slots[SLOT_GRAM] if slots[SLOT_GRAM][1] else slots[SLOT_GRAM_SMALL]
            iso_us = max(i_ms * 1e3 / max(i_launches, 1), 1e-9)
            iso_tf = i_flops / max(i_launches, 1) / (iso_us * 1e-6) / 1e12
            roofline["isolated"] = {
                "workload": f"{nb} x ({m_g} x {n_g}) fp32 matrices (the C-order volumes), one launch, nothing else on the GPU",
                "launch_us": iso_us, "achieved": iso_tf, "frac": iso_tf / F64_MFMA_PEAK_TFLOPS}
            del gws, gout
    # reshape stage alone (the kernel north_star's ">= 50 % of the HBM-read roofline" refers to): the bond-capped fp32
    # path does not run it in Std mode (the permutation rides on the Gram pass, the projection and the last chain
    # product), other paths do: the tiled permute of one volume, HIP events around 10 launches
    from imgcompressionmps_amd.core.ndmps import _plan_for

    plan = _plan_for(shape, device.index or 0)
    dense = torch.empty(n_vox, dtype=torch.float32, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.check(lib.ndmps_encode_permute(plan.handle, x.data_ptr(), dense.data_ptr(), 4, _lib.stream_ptr()))
    e0.record()
    for _ in range(10):
        _lib.check(lib.ndmps_encode_permute(plan.handle, x.data_ptr(), dense.data_ptr(), 4, _lib.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    perm_ms = e0.elapsed_time(e1) / 10
    roofline["reshape_stage"] = {
        "kernel": "encode_tiled_kernel<uint32, vec>", "launch_us": perm_ms * 1e3,
        "read_frac": (4 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
        "read_write_frac": (8 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS}
    del dense
    # ... and of a lockstep group in ONE launch (ndmps_encode_permute_many, what bf16 / fp64 storage and the DCT-free
    # unfused paths run since round 4): per-volume launches of a 128^3 volume are mostly launch
    ng = min(32, len(xs))
    if ng > 1:
        import ctypes as C

        dense_g = torch.empty((ng, n_vox), dtype=torch.float32, device=device)
        src_p = (C.c_void_p * ng)(*[v.data_ptr() for v in xs[:ng]])
        dst_p = (C.c_void_p * ng)(*[d.data_ptr() for d in dense_g.unbind(0)])
        _lib.check(lib.ndmps_encode_permute_many(plan.handle, ng, src_p, dst_p, 4, _lib.stream_ptr()))
        e0.record()
        for _ in range(5):
            _lib.check(lib.ndmps_encode_permute_many(plan.handle, ng, src_p, dst_p, 4, _lib.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        grp_ms = e0.elapsed_time(e1) / 5
        roofline["reshape_stage_group"] = {
            "kernel": "encode_tiled_kernel<uint32, vec>, one launch for the group", "volumes": ng, "launch_us": grp_ms * 1e3,
            "read_frac": (4 * ng * n_vox / (grp_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
            "read_write_frac": (8 * ng * n_vox / (grp_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS}
        del dense_g

    def streamed_ms(vols, chi, batches=24):
        """ms per batch of a STREAM of such batches (what a rank of a strong-scaling run sees step after step):
        encode_decode_begin in the default stream shape (one group, three alternating sets of streams), objects built
        `lanes` batches later."""
        n_groups, n_lanes = batch_mod.default_stream_shape(len(vols), min(8 * chi, 512))
        flight = []

        def go(k):
            flight.append(batch_mod.encode_decode_begin(vols, groups=n_groups, mode=job["mode"], max_bond=chi, lane=k,
                                                        lanes=n_lanes))
            while len(flight) > n_lanes:
                flight.pop(0).result()

        for k in range(2 * n_lanes):
            go(k)
        while flight:
            flight.pop(0).result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(batches):
            go(k)
        while flight:
            flight.pop(0).result()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / batches * 1e3

    # one lockstep group of 8 on one stream (per-stage device times undisturbed by concurrent groups), a single volume
    n8 = min(8, len(xs))
    group_step(xs[:n8])
    timer8 = ndmps_mod.StageTimer()
    ndmps_mod.set_stage_timer(timer8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        group_step(xs[:n8])
    torch.cuda.synchronize()
    group8_ms = (time.perf_counter() - t0) / 3 * 1e3
    ndmps_mod.set_stage_timer(None)
    single_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        single_step()
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t0) / 5 * 1e3
    line["single_volume"] = {"ms": single_ms, "Mvoxels_per_s": n_vox / single_ms / 1e3}
    if ctx["world"] == 1:  # ... and volume after volume (encode_decode_begin on three lanes): what a caller with a queue of them sees
        one_ms = streamed_ms([x], job["chi"], batches=30)
        line["single_volume"].update({"ms_per_volume_in_a_stream": one_ms, "Mvoxels_per_s_in_a_stream": n_vox / one_ms / 1e3})
    line["one_group_of_8"] = {"ms": group8_ms, "Mvoxels_per_s": n8 * n_vox / group8_ms / 1e3, "volumes": n8,
                              "stages": {k: {"ms_per_step": v[0] / 3, "launches_per_step": v[1] / 3}
                                         for k, v in timer8.totals_ms().items()}}

    # BASELINE config 4's question asked of ONE GPU: 64 volumes in total over 8 GPUs put 8 on each; the best a perfect
    # 8-GPU run can do is t(64 volumes) / t(8 volumes), both measured here on one GPU.  Not a measured scaling curve.
    if job["config"] == "metric" and not args_overridden(args) and ctx["world"] == 1 and len(xs) >= 64:
        s8 = streamed_ms(xs[:n8], job["chi"])
        proj = {"256^3 chi=64": {"ms_64_volumes": ms_per_step, "ms_8_volumes": group8_ms, "ratio": ms_per_step / group8_ms,
                                 "ms_8_volumes_streamed": s8, "ratio_streamed": ms_per_step / s8}}
        small = [synthetic_mri_device((128, 128, 128), FIRST_SEED + j, device) for j in range(64)]

        def small_step(vols, groups):
            batch_mod.encode_decode_concurrent(vols, groups=groups, mode="Std", max_bond=32, wait=True)

        times = {}
        for name, vols, groups in (("64", small, 1), ("8", small[:8], 1)):
            small_step(vols, groups)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                small_step(vols, groups)
            torch.cuda.synchronize()
            times[name] = (time.perf_counter() - t0) / 5 * 1e3
        s8 = streamed_ms(small[:8], 32)
        s64 = streamed_ms(small, 32, batches=12)
        proj["128^3 chi=32"] = {"ms_64_volumes": times["64"], "ms_8_volumes": times["8"], "ratio": times["64"] / times["8"],
                                "ms_64_volumes_streamed": s64, "ms_8_volumes_streamed": s8, "ratio_streamed": s64 / s8}
        proj["note"] = ("upper bound of the 8-GPU speed-up on 64 volumes in total (8 per GPU): time of 64 volumes / time of "
                        "8 volumes, both on this one GPU; no multi-GPU run is behind it.  `ratio`: one batch alone, waited "
                        "for (latency); `ratio_streamed`: batch after batch (steady state of the timed loop of `bench.py "
                        "--total-volumes 64 --gpus 8`: consecutive batches alternate between three sets of streams, "
                        "core/batch.py default_lanes)")
        line["strong_scaling_projection"] = proj
        del small

        # the reference's literal flow (core/ndmps.py:74 without a bond cap, then :94-108 compress(cutoff), then
        # :131-153): exact sweep at cutoff 1e-10, compress(0.01), to_tensor; fp32 and fp64 storage, one volume
        flow = {}
        for name, dtype in (("fp32_storage", None), ("fp64_storage", torch.float64)):
            def once():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = NDMPS.from_tensor(x, dtype=dtype)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                exact_bonds = o.bond_sizes()
                back = o.to_tensor(as_torch=True)
                round_trip = float((back.double() - x.double()).abs().max())
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                o.compress(0.01)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                r = o.to_tensor(as_torch=True)
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                err = float((r.double() - x.double()).norm() / x.double().norm())
                return {"from_tensor_exact_ms": (t1 - t0) * 1e3, "compress_0.01_ms": (t3 - t2) * 1e3,
                        "to_tensor_ms": (t4 - t3) * 1e3, "exact_bonds": exact_bonds, "round_trip_max_abs": round_trip,
                        "bonds_after_compress": o.bond_sizes(), "rel_frobenius_error_after_compress": err,
                        "Mvoxels_per_s": n_vox / ((t1 - t0) + (t3 - t2) + (t4 - t3)) / 1e6}

            once()  # warm-up: allocator, plans
            flow[name] = once()
        # the same flow through the reference's list functions (evaluation/benchmark.py:58-118: conv_to_mps, compress_list,
        # conv_to_tensors) on 8 volumes: chunks on three streams, a compress per host thread
        def list_flow():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            objs = batch_mod.conv_to_mps(xs[:8], mode="Std")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            batch_mod.compress_list(objs, 0.01)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            recs = batch_mod.conv_to_tensors(objs, as_torch=True)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            err = float((recs[0].double() - xs[0].double()).norm() / xs[0].double().norm())
            return {"volumes": 8, "conv_to_mps_ms_per_volume": (t1 - t0) / 8 * 1e3, "compress_list_0.01_ms_per_volume": (t2 - t1) / 8 * 1e3,
                    "conv_to_tensors_ms_per_volume": (t3 - t2) / 8 * 1e3, "bonds_after_compress": objs[0].bond_sizes(),
                    "rel_frobenius_error_after_compress": err, "Mvoxels_per_s": 8 * n_vox / (t3 - t0) / 1e6}

        list_flow()
        flow["fp32_storage_list_functions"] = list_flow()
        flow["note"] = ("NDMPS.from_tensor(x) with quimb's default cutoff 1e-10 and NO bond cap, compress(0.01), to_tensor -- "
                        "what evaluation/benchmark.py:176-178 does per volume; one 256^3 volume, not part of `value`")
        line["reference_flow"] = flow


# ------------------------------------------------------------------------------------------ one large tensor
def run_tensor(ctx):
    """BASELINE config 5: ONE 4-D tensor.  One GPU: bf16 storage end to end.  N > 1: the rows of every unfolding are
    sharded over the ranks (rank r holds the top-level blocks with digits [r, r + 1) d_0 / N), one all-reduce of an
    n x n fp64 Gram matrix per sharded site (the path's one real exchange step), fp32 storage; every rank reconstructs
    its own rows."""
    args, job, rank, world, device = ctx["args"], ctx["job"], ctx["rank"], ctx["world"], ctx["device"]
    torch, lib, NDMPS, _lib, ndmps_mod = ctx["torch"], ctx["lib"], ctx["NDMPS"], ctx["_lib"], ctx["ndmps_mod"]
    from imgcompressionmps_amd.core import sharded

    shape = tuple(job["shape"])
    n_vox = int(np.prod(shape))
    chi = job["chi"]
    x = synthetic_mri_device(shape, job["first_seed"], device)  # same tensor on every rank (same seed, same stream)
    last = {}
    drain = None
    lanes = 1
    if world == 1:
        xb = x.to(torch.bfloat16)
        del x
        # A stream of tensors: tensor k runs on lane k mod lanes (a host thread and a stream each), the step returns once it
        # is handed over and waits for tensor k - lanes; the resident tridiagonalisations of one tensor (latency-bound, 21 ms
        # of its 37) then overlap the Gram / projection / permute work of the other.  --no-pipeline / --lanes 1: one tensor
        # at a time, waited for.
        lanes = 1 if args.no_pipeline else (max(1, args.lanes) if args.lanes else 2)
        if lanes > 1:
            from concurrent.futures import ThreadPoolExecutor

            from imgcompressionmps_amd.core import batch as batch_mod

            lane_streams = batch_mod.group_streams(lanes)
            pool = ThreadPoolExecutor(lanes)
            in_flight, counter = [], [0]
            device_index = torch.cuda.current_device()

            def work(slot):
                torch.cuda.set_device(device_index)
                with torch.cuda.stream(lane_streams[slot]):
                    o = NDMPS.from_tensor(xb, mode=job["mode"], max_bond=chi, dtype=torch.bfloat16)
                    rec = o.to_tensor(as_torch=True)
                lane_streams[slot].synchronize()
                return o, rec

            def collect(fut):
                last["obj"], last["rec"] = fut.result()

            def step():
                in_flight.append(pool.submit(work, counter[0] % lanes))
                counter[0] += 1
                while len(in_flight) > lanes:
                    collect(in_flight.pop(0))

            def drain():
                while in_flight:
                    collect(in_flight.pop(0))
        else:
            def step():
                o = NDMPS.from_tensor(xb, mode=job["mode"], max_bond=chi, dtype=torch.bfloat16)
                last["obj"], last["rec"] = o, o.to_tensor(as_torch=True)

        dtype, storage = "bf16", "bf16 storage (volume, carried matrices, cores, reconstruction), fp64 Gram / eigen"
    else:
        from imgcompressionmps_amd.utils import core as hcore

        d0 = int(hcore.site_dims(shape)[0])
        if d0 % world:
            raise SystemExit(f"config 5 shards the {d0} top-level blocks: --gpus must divide {d0}")
        per = d0 // world
        blocks = [x[sharded.top_block_slices(shape, d)].contiguous() for d in range(rank * per, (rank + 1) * per)]
        del x

        def step():
            mps = sharded.from_volume_sharded(blocks, shape, max_bond=chi)
            last["obj"], last["rec"] = mps, sharded.local_dense(mps, rank, world)

        dtype, storage = "f32", "fp32 storage, rows of the unfoldings sharded over the ranks (core/sharded.py)"

    region = ProfiledRegion()

    def start_profiling():
        _lib.check(lib.ndmps_profile_enable(1))
        torch.cuda.synchronize()
        region.resume()

    if lanes > 1:  # set-up, not a step: every lane's allocator pool is filled (two tensors per lane)
        for _ in range(2 * lanes):
            step()
        drain()
        torch.cuda.synchronize()
    elapsed = timed_steps(step, args.steps, args.warmup, ctx["barrier"], ctx["reduce_max"], after_warmup=start_profiling,
                          drain=drain)
    region.pause()
    _lib.check(lib.ndmps_profile_enable(0))
    value = n_vox * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    roofline = roofline_of(collect_slots(lib), args.steps)
    esize = 2 if dtype == "bf16" else 4
    roofline["end_to_end_algorithmic_GBps"] = 2 * esize * n_vox / (ms_per_step * 1e-3) / 1e9
    roofline["end_to_end_frac"] = roofline["end_to_end_algorithmic_GBps"] / HBM_PEAK_GBPS
    obj = last["obj"]
    line = base_line(
        job, args, world, value, ms_per_step, dtype,
        f"ONE {'x'.join(str(v) for v in shape)} synthetic fMRI tensor per step, bond cap {chi}, {storage}; "
        f"encode (bond cap applied in the sweep) + reconstruct, device-resident in/out",
        {"volumes_per_step": 1, "bonds": obj.bond_sizes(),
         "host_pipeline": (f"a stream of tensors: tensor k on lane k mod {lanes} (a host thread and a stream each), the step waits "
                           f"for tensor k - {lanes}; the last tensors are waited for inside the timed region; set-up ran two "
                           "untimed tensors per lane" if lanes > 1 else "none: one tensor at a time, waited for"),
         "parallelism": ("one GPU" if world == 1 else
                         f"{world} ranks, each holding {n_vox // world} voxels (its top-level blocks); one all-reduce of an "
                         f"n x n fp64 Gram matrix per sharded site ({args.backend}), the replicated tail of the sweep on every rank")})
    line["roofline"] = roofline
    line["team_fallbacks"] = int(lib.ndmps_syevd_topk_team_fallbacks())
    line["timed_region_clock_ns"] = region.window()
    if world == 1 and not args.skip_single:
        t = {}
        for name, fn in (("encode", lambda: NDMPS.from_tensor(xb, mode=job["mode"], max_bond=chi, dtype=torch.bfloat16)),
                         ("reconstruct", lambda: obj.to_tensor(as_torch=True))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t[name + "_ms"] = (time.perf_counter() - t0) / 3 * 1e3
        line["stages"] = t
    if rank == 0 and world == 1 and not args.skip_single:
        # ---- what bf16 costs, and where: bf16 everywhere (the storage type of `value`: volume, carried matrices, cores,
        # reconstruction) against bf16 volume / cores / reconstruction with the SWEEP in fp32 (carry_dtype: the volume is
        # widened once, nothing is rounded between the sites), both against the NumPy oracle on the bf16-rounded sample
        # and timed at full size
        from oracle.metrics import compute_ssim_by_dim as ssim_host

        s_shape = (64, 64, 32, 64)
        ref_b, rec_ref_b, _cpu, _thr, s_xb = oracle_sample(s_shape, job["first_seed"], chi, job["mode"], bf16=True)
        x64 = s_xb.astype(np.float64)
        ssim_ref = ssim_host(x64, rec_ref_b)
        price = {"sample": "64 x 64 x 32 x 64 of the same generator rounded to bf16, chi = 128 binding on two bonds; ms at full size"}
        sx = torch.from_numpy(s_xb).to(device).to(torch.bfloat16)
        for name, carry in (("bf16_everywhere", None), ("bf16_storage_fp32_sweep", torch.float32)):
            so = NDMPS.from_tensor(sx, mode=job["mode"], max_bond=chi, dtype=torch.bfloat16, carry_dtype=carry)
            srec = so.to_tensor(as_torch=True).float().cpu().numpy().astype(np.float64)

            def full_step():
                o = NDMPS.from_tensor(xb, mode=job["mode"], max_bond=chi, dtype=torch.bfloat16, carry_dtype=carry)
                return o.to_tensor(as_torch=True)

            full_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                full_step()
            torch.cuda.synchronize()
            ms_full = (time.perf_counter() - t0) / 2 * 1e3
            price[name] = {"ssim_gap": abs(ssim_host(x64, srec) - ssim_ref),
                           "rel_frobenius_vs_oracle": float(np.linalg.norm(srec - rec_ref_b) / np.linalg.norm(rec_ref_b)),
                           "bonds_equal": so.bond_sizes() == ref_b.bond_sizes(), "ms_per_step_full_size": ms_full}
        price["default"] = ("bf16 everywhere: the error is the rounding of the volume, the cores and the reconstruction to bf16; "
                            "an fp32 sweep in between changes neither figure and costs the fp32 copy of the volume")
        line["bf16_price"] = price
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample: the same recipe at 64 x 64 x 32 x 64 (1/32 of the voxels; site dims [64, 16, 16, 16, 32],
        # exact bonds [64, 1024, 512, 32]: chi = 128 binds on two bonds, the order-2048 eigenproblem is on the path)
        s_shape = (64, 64, 32, 64)
        ref, rec_ref, cpu_s, threads, s_x = oracle_sample(s_shape, job["first_seed"], chi, job["mode"], bf16=True)
        s_obj = NDMPS.from_tensor(torch.from_numpy(s_x).to(device).to(torch.bfloat16), mode=job["mode"], max_bond=chi,
                                  dtype=torch.bfloat16)
        s_rec = s_obj.to_tensor(as_torch=True).float().cpu().numpy()
        what = ("a 64 x 64 x 32 x 64 tensor of the same generator rounded to bf16 (1/32 of the voxels; the oracle's sweep of "
                "the full tensor takes minutes), chi = 128 binding on two bonds")
        line["cpu_baseline"] = {
            "value": int(np.prod(s_shape)) / cpu_s / 1e6, "unit": "Mvoxels/s", "cores": threads, "kind": "port",
            "sample": f"{what}; NumPy fp64 oracle (closed-form index permutation, SVD sweep with max_bond={chi}, chain "
                      f"contraction), {cpu_s:.1f} s wall"}
        line["parity"] = parity_block(s_x, s_rec, rec_ref, s_obj.bond_sizes(), ref.bond_sizes())
        line["parity"]["sample"] = what
        line["parity"]["tolerance"] = "bf16 storage: 2e-2 relative (tests/test_gpu_parity.py BF16_TOL)"
    return line


if __name__ == "__main__":
    main()
