#!/usr/bin/env python
"""Headline benchmark: Mvoxels/s of compress + reconstruct (NDMPS.from_tensor with the bond cap
applied in the sweep, then NDMPS.to_tensor) on synthetic 256^3 fp32 volumes at chi = 64
(BASELINE.json "metric"), input and output resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; a step is one batch of --batch independent volumes per GPU (encoded in lockstep
groups by NDMPS.from_tensors, reconstructed by NDMPS.to_tensors); independent volumes shard over the ranks with
no data-path collective -> weak scaling; the only collective before the timed region is the
RCCL broadcast of the job descriptor.  Rank 0 prints one JSON line.  At N = 1 it also carries the CPU
baseline (the NumPy oracle on one volume of the batch, timed on the host cores) and the SSIM gap between
the GPU and the oracle reconstruction of that volume.

The rank / shard / timing logic lives in functions (job_descriptor, volume_seeds, timed_steps, throughput)
that tests/test_batch_sharding.py drives at world size 2 under gloo with a stub step.
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256, help="edge of the cubic volume")
    ap.add_argument("--chi", type=int, default=64)
    ap.add_argument("--mode", default="Std", choices=["Std", "DCT"])
    ap.add_argument("--batch", type=int, default=64, help="independent volumes per GPU per step")
    ap.add_argument("--groups", type=int, default=2,
                    help="concurrent groups (host thread + HIP stream each) the batch is cut into; "
                         "volumes of a group are encoded in lockstep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="profiling aid: no single-volume phase")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses N ranks on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ rank / shard / timing logic (GPU-free)
def job_descriptor(args, world):
    """What rank 0 broadcasts: everything that decides which volumes exist and how they are processed."""
    return {"size": int(args.size), "chi": int(args.chi), "mode": str(args.mode), "batch_per_gpu": int(args.batch),
            "groups": int(args.groups), "world": int(world), "first_seed": FIRST_SEED,
            "n_volumes": int(args.batch) * int(world)}


def volume_seeds(job, rank):
    """Seeds of the volumes rank `rank` owns: the job's volumes are numbered 0 .. n_volumes-1 with seeds
    first_seed + j (2025 .. 2088 for the 64 volumes of BASELINE's batch config); ranks own contiguous blocks
    (core/batch.shard_indices), so every volume of the job is distinct and owned exactly once."""
    from imgcompressionmps_amd.core.batch import shard_indices

    return [job["first_seed"] + j for j in shard_indices(job["n_volumes"], rank, job["world"])]


def timed_steps(step, steps, warmup, barrier, reduce_max, after_warmup=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier() on both sides (barrier = process
    group barrier + device synchronise); returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step()
    if after_warmup is not None:
        after_warmup()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return float(reduce_max(time.perf_counter() - t0))


def throughput(job, n_vox, steps, elapsed):
    """Whole-job Mvoxels/s: every rank processed batch_per_gpu volumes per step."""
    return job["world"] * job["batch_per_gpu"] * n_vox * steps / elapsed / 1e6


# ------------------------------------------------------------------------------------------ data
def synthetic_mri_device(shape, seed, device):
    """SURVEY 8(d)'s synthetic MRI volume generated on the device: the oracle generator's recipe and its
    parameter draws (oracle/metrics.py synthetic_mri: same blobs and shell for the same seed), evaluated with
    torch; the white noise comes from a torch generator with that seed (NumPy's stream is not reproduced)."""
    import torch

    rng = np.random.default_rng(seed)
    axes = [torch.linspace(-1.0, 1.0, n, dtype=torch.float64, device=device) for n in shape]
    vol = torch.zeros(shape, dtype=torch.float64, device=device)
    for _ in range(12):
        amp = rng.uniform(0.3, 1.0)
        fac = []
        for a in axes:
            c = rng.uniform(-0.6, 0.6)
            w = rng.uniform(0.08, 0.45)
            fac.append(torch.exp(-0.5 * ((a - c) / w) ** 2))
        vol += amp * fac[0][:, None, None] * fac[1][None, :, None] * fac[2][None, None, :]
    r2 = torch.zeros(shape, dtype=torch.float64, device=device)
    for j, a in enumerate(axes):
        e = (a / rng.uniform(0.75, 0.95)) ** 2
        r2 += e.reshape((1,) * j + (-1,) + (1,) * (2 - j))
    vol += 0.8 * torch.exp(-0.5 * ((torch.sqrt(r2) - 1.0) / 0.04) ** 2)
    gen = torch.Generator(device=device).manual_seed(int(seed))
    vol += 0.01 * torch.randn(shape, dtype=torch.float64, device=device, generator=gen)
    vol -= vol.min()
    vol /= vol.max()
    return vol.to(torch.float32)


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/r02_pmc_*.json, written
    by tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs), or None."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_summary.json")
    try:
        with open(path) as f:
            entry = json.load(f)[kernel_key]
        return float(entry["hbm_bytes_per_launch"]), entry
    except (OSError, KeyError, ValueError):
        return None, None


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

    from oracle.metrics import compute_ssim_by_dim, synthetic_mri  # checker + the one host-generated volume

    shape = (job["size"],) * 3
    n_vox = int(np.prod(shape))
    seeds = volume_seeds(job, rank)
    # volume 0 of every rank comes from the host generator (it is the one the oracle re-encodes at N = 1);
    # the others are generated on the device with the same recipe, one distinct seed each
    x_host = synthetic_mri(shape, seed=seeds[0])
    xs = [torch.from_numpy(x_host).to(device)] + [synthetic_mri_device(shape, sd, device) for sd in seeds[1:]]
    x = xs[0]

    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(max(1, min(job["groups"], job["batch_per_gpu"])))
    last = {}

    def step():
        # wait=False: the step returns when everything is enqueued; the next step queues behind it on the same
        # streams (like the steps of a training loop) and the barrier of the timed region synchronises the device
        objs, recs = batch_mod.encode_decode_concurrent(xs, groups=job["groups"], mode=job["mode"],
                                                         max_bond=job["chi"], pool=pool, wait=False)
        last["obj"], last["rec"] = objs[0], recs[0]

    def group_step(n):  # one lockstep group of n volumes on the current stream
        objs = NDMPS.from_tensors(xs[:n], mode=job["mode"], max_bond=job["chi"])
        return [o.to_tensor(as_torch=True) for o in objs]

    def single_step():
        o = NDMPS.from_tensor(x, mode=job["mode"], max_bond=job["chi"])
        return o, o.to_tensor(as_torch=True)

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

    timer = ndmps_mod.StageTimer()

    def start_profiling():
        ndmps_mod.set_stage_timer(timer)
        _lib.check(lib.ndmps_profile_enable(1))

    elapsed = timed_steps(step, args.steps, args.warmup, barrier, reduce_max, after_warmup=start_profiling)
    ndmps_mod.set_stage_timer(None)
    _lib.check(lib.ndmps_profile_enable(0))
    obj, rec = last["obj"], last["rec"]

    stages = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps}
              for k, v in timer.totals_ms().items()}
    value = throughput(job, n_vox, args.steps, elapsed)
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (largest share of device time, profiles/r02_*): gram128_kernel, the fp64
    # Gram matrices of a lockstep group in one launch.  MFMA-bound (v_mfma_f64_16x16x4_f64): algorithmic flops =
    # m n (n + 1) per matrix (upper triangle incl. the diagonal, 2 flops per product), counted by the library
    # per launch; duration = HIP events on the launching stream around every Gram launch of the TIMED REGION.
    import ctypes as C

    def collect(slot):
        ms, launches, amount = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(lib.ndmps_profile_collect(slot, C.byref(ms), C.byref(launches), C.byref(amount)))
        return ms.value, launches.value, amount.value

    SLOT_TEAM, SLOT_GRAM = 2, 3
    g_ms, g_launches, g_flops = collect(SLOT_GRAM)
    t_ms, t_launches, t_bytes = collect(SLOT_TEAM)
    gram_us = g_ms * 1e3 / max(g_launches, 1)
    gram_flops = g_flops / max(g_launches, 1)
    achieved = gram_flops / (gram_us * 1e-6) / 1e12 if g_launches else float("nan")
    traffic, pmc_entry = pmc_traffic("gram128_kernel")
    algo_bytes_e2e = 2 * 4 * n_vox
    roofline = {
        "kernel": "gram128_kernel<float> (fp64 Gram matrices A^T A of a lockstep group in one launch, "
                  "v_mfma_f64_16x16x4_f64; all sites' launches averaged, the 32768 x 512 raw Gram dominates)",
        "bound": "mfma",
        "achieved": achieved,
        "peak": F64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / F64_MFMA_PEAK_TFLOPS,
        "traffic": traffic,
        "flops_per_launch": gram_flops,
        "launch_us": gram_us,
        "launches_per_step": g_launches / args.steps,
        "measured": "HIP events on the launching streams around every Gram launch of the timed region "
                    f"({job['groups']} groups in flight share the matrix cores)",
        "peak_source": "AMD MI355X spec, FP64 matrix 78.6 TFLOP/s = 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz; "
                       "77.7 measured with tools/scratch/mfma_f64_rate.hip (MI355X_MICROARCH.md lists no f64 row)",
        "traffic_source": pmc_entry,
        "end_to_end_algorithmic_GBps": job["batch_per_gpu"] * algo_bytes_e2e / (ms_per_step * 1e-3) / 1e9,
        "end_to_end_frac": job["batch_per_gpu"] * algo_bytes_e2e / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        # second kernel by device time: the register-resident Householder tridiagonalisation (latency-bound: one
        # exchange between the workgroups of a matrix per column, no trailing-matrix traffic)
        "tridiagonalisation": {
            "kernel": "trd_team_kernel<2>",
            "launch_ms": t_ms / max(t_launches, 1),
            "launches_per_step": t_launches / args.steps,
            "hbm_bytes_per_launch_algorithmic": t_bytes / max(t_launches, 1),
            "bound": "latency of the per-column exchange (384 columns per order-512 matrix)",
        },
    }

    # the same Gram launch with nothing else on the GPU: one lockstep group's raw Gram (m = N / 512 rows, 512 columns)
    if not args.skip_single and job["size"] >= 64:
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
            i_ms, i_launches, i_flops = collect(SLOT_GRAM)
            iso_us = max(i_ms * 1e3 / max(i_launches, 1), 1e-9)
            iso_tf = i_flops / max(i_launches, 1) / (iso_us * 1e-6) / 1e12
            roofline["isolated"] = {
                "workload": f"{nb} x ({m_g} x {n_g}) fp32 matrices (the C-order volumes), one launch, nothing else on the GPU",
                "launch_us": iso_us,
                "achieved": iso_tf,
                "frac": iso_tf / F64_MFMA_PEAK_TFLOPS,
            }
            del gws, gout
        # reshape stage alone (the kernel north_star's ">= 50 % of the HBM-read roofline" refers to): the bond-capped
        # fp32 path never runs it (the permutation rides on the Gram pass, the projection and the last chain
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
            "kernel": "encode_tiled_kernel<uint32, vec>",
            "launch_us": perm_ms * 1e3,
            "read_frac": (4 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
            "read_write_frac": (8 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
        }
        del dense

    # reference points, same kernels, not part of `value`: one lockstep group of 8 on one stream
    # (with per-stage device times undisturbed by concurrent groups) and a single volume
    single_ms = group8_ms = float("nan")
    stages_group8 = {}
    if not args.skip_single:
        n8 = min(8, job["batch_per_gpu"])
        group_step(n8)
        timer8 = ndmps_mod.StageTimer()
        ndmps_mod.set_stage_timer(timer8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            group_step(n8)
        torch.cuda.synchronize()
        group8_ms = (time.perf_counter() - t0) / 3 * 1e3
        ndmps_mod.set_stage_timer(None)
        stages_group8 = {k: {"ms_per_step": v[0] / 3, "launches_per_step": v[1] / 3}
                         for k, v in timer8.totals_ms().items()}
        single_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            single_step()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t0) / 5 * 1e3

    line = {
        "metric": METRIC,
        "value": value,
        "unit": "Mvoxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"batch of {job['batch_per_gpu']} independent {job['size']}^3 fp32 synthetic MRI volumes per GPU "
                        f"per step ({job['groups']} concurrent groups, lockstep inside a group), "
                        f"NDMPS.from_tensors(max_bond={job['chi']}, mode={job['mode']}) + to_tensor each, "
                        f"device-resident in/out",
            "volumes_per_step": job["n_volumes"],
            "volume_source": f"every volume distinct: seeds {job['first_seed']}..{job['first_seed'] + job['n_volumes'] - 1}, "
                             "block-sharded over the ranks; volume 0 of a rank from the host generator, the rest "
                             "from its device-side twin (same blobs / shell per seed, torch noise stream)",
            "batch_per_gpu": job["batch_per_gpu"],
            "groups_per_gpu": job["groups"],
            "bonds": obj.bond_sizes(),
            "parallelism": f"{world} independent volume shard(s), no data-path collective; job descriptor broadcast "
                           f"from rank 0" + (f" ({args.backend})" if world > 1 else " (single rank: none)"),
        },
        "roofline": roofline,
        "stages": stages,
        "single_volume": {"ms": single_ms, "Mvoxels_per_s": n_vox / single_ms / 1e3},
        "one_group_of_8": {"ms": group8_ms, "Mvoxels_per_s": min(8, job["batch_per_gpu"]) * n_vox / group8_ms / 1e3,
                           "stages": stages_group8},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.ndmps_oracle import OracleNDMPS

        try:
            from threadpoolctl import threadpool_info

            threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        except Exception:
            threads = os.cpu_count() or 1
        t0 = time.perf_counter()
        ref = OracleNDMPS.from_tensor(x_host, mode=job["mode"], max_bond=job["chi"], materialise_map=False)
        rec_ref = ref.to_tensor()
        cpu_s = time.perf_counter() - t0
        rec_gpu = rec.cpu().numpy().astype(np.float64)
        x64 = x_host.astype(np.float64)
        ssim_gpu = float(compute_ssim_by_dim(x64, rec_gpu))
        ssim_ref = float(compute_ssim_by_dim(x64, rec_ref))
        line["cpu_baseline"] = {
            "value": n_vox / cpu_s / 1e6,
            "unit": "Mvoxels/s",
            "cores": int(threads),
            "kind": "port",
            "sample": f"one {job['size']}^3 volume, NumPy fp64 oracle (closed-form index permutation, SVD sweep "
                      f"with max_bond={job['chi']}, chain contraction), {cpu_s:.1f} s wall",
        }
        line["parity"] = {
            "ssim_gpu": ssim_gpu,
            "ssim_oracle": ssim_ref,
            "ssim_gap": abs(ssim_gpu - ssim_ref),
            "rel_frobenius_vs_oracle": float(np.linalg.norm(rec_gpu - rec_ref) / np.linalg.norm(rec_ref)),
            "max_abs_vs_oracle": float(np.abs(rec_gpu - rec_ref).max()),
            "bonds_equal": obj.bond_sizes() == ref.bond_sizes(),
        }

    if rank == 0:
        print(json.dumps(line))
    pool.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
