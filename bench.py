#!/usr/bin/env python
"""Headline benchmark: Mvoxels/s of compress + reconstruct (NDMPS.from_tensor with the bond cap
applied in the sweep, then NDMPS.to_tensor) on synthetic 256^3 fp32 volumes at chi = 64
(BASELINE.json "metric"), input and output resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; a step is one batch of --batch independent volumes per GPU (encoded in lockstep
groups by NDMPS.from_tensors, reconstructed one by one); independent volumes shard over the ranks with
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
        objs, recs = batch_mod.encode_decode_concurrent(xs, groups=job["groups"], mode=job["mode"],
                                                         max_bond=job["chi"], pool=pool)
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

    # ---- roofline of the dominant kernel (largest share of device time, profiles/r02_*): the column launches of
    # the Householder tridiagonalisation.  Every launch reads and writes the trailing matrices once (fp64):
    # algorithmic bytes = 2 * 8 * sum over its matrices of (n - j - 1)^2, counted by the library per launch;
    # duration = HIP events on the launching stream around every column sweep of the TIMED REGION.
    import ctypes as C

    ms, launches, nbytes = C.c_double(), C.c_int64(), C.c_int64()
    _lib.check(lib.ndmps_profile_collect(1, C.byref(ms), C.byref(launches), C.byref(nbytes)))
    col_us = max(ms.value * 1e3 / max(launches.value, 1), 1e-9)
    col_bytes = nbytes.value / max(launches.value, 1)
    achieved = col_bytes / (col_us * 1e-6) / 1e9 if launches.value else float("nan")
    traffic, pmc_entry = pmc_traffic("trd_column_kernel")
    algo_bytes_e2e = 2 * 4 * n_vox
    roofline = {
        "kernel": "trd_column_kernel<2, 32> (Householder tridiagonalisation, one launch per column, 16 order-512 "
                  "matrices per launch and group)",
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": traffic,
        "bytes_per_launch": col_bytes,
        "launch_us": col_us,
        "launches_per_step": launches.value / args.steps,
        "measured": "HIP events on the launching streams around every column sweep of the timed region "
                    f"({job['groups']} groups in flight share the HBM)",
        "traffic_source": pmc_entry,
        "end_to_end_algorithmic_GBps": job["batch_per_gpu"] * algo_bytes_e2e / (ms_per_step * 1e-3) / 1e9,
        "end_to_end_frac": job["batch_per_gpu"] * algo_bytes_e2e / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
    }

    # the same launches with nothing else on the GPU: 16 Gram matrices of order 512 built from the batch (the
    # shape of the three big eigenproblems of every volume), one stream
    if not args.skip_single and job["size"] >= 64:
        nb_eig, n_eig = 16, 512
        a = torch.stack([xs[j % len(xs)].reshape(n_eig, -1).to(torch.float64) for j in range(nb_eig)])
        g0 = torch.bmm(a, a.transpose(1, 2)).contiguous()
        del a
        vv = torch.empty_like(g0)
        ww = torch.empty((nb_eig, n_eig), dtype=torch.float64, device=device)
        wsb_n = lib.ndmps_syevd_topk_workspace_bytes(n_eig, nb_eig, job["chi"])
        wsb = torch.empty(int(wsb_n), dtype=torch.uint8, device=device)
        sizes = _lib.i64_array([n_eig] * nb_eig)

        def values():
            _lib.check(lib.ndmps_syevd_topk_values_f64(nb_eig, g0.data_ptr(), n_eig * n_eig, sizes, vv.data_ptr(),
                                                       n_eig * n_eig, ww.data_ptr(), n_eig, min(job["chi"], 128),
                                                       wsb.data_ptr(), wsb_n, _lib.stream_ptr()))

        values()
        torch.cuda.synchronize()
        _lib.check(lib.ndmps_profile_enable(1))
        for _ in range(3):
            values()
        torch.cuda.synchronize()
        _lib.check(lib.ndmps_profile_enable(0))
        _lib.check(lib.ndmps_profile_collect(1, C.byref(ms), C.byref(launches), C.byref(nbytes)))
        iso_us = max(ms.value * 1e3 / max(launches.value, 1), 1e-9)
        iso_bytes = nbytes.value / max(launches.value, 1)
        roofline["isolated"] = {
            "workload": f"{nb_eig} x ({n_eig} x {n_eig}) fp64 Gram matrices, one stream, nothing else on the GPU",
            "launch_us": iso_us,
            "bytes_per_launch": iso_bytes,
            "achieved": iso_bytes / (iso_us * 1e-6) / 1e9,
            "frac": iso_bytes / (iso_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
        }
        del g0, vv, ww, wsb

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
        # reshape stage (the kernel north_star's ">= 50 % of the HBM-read roofline" refers to): read fraction of
        # the tiled permute alone, from the one-group phase (nothing else on the GPU)
        e = stages_group8.get("encode_permute", {})
        perm_ms = e.get("ms_per_step", float("nan")) / max(e.get("launches_per_step", 1.0), 1.0)
        roofline["reshape_stage"] = {
            "kernel": "encode_tiled_kernel<uint32, vec>",
            "launch_us": perm_ms * 1e3,
            "read_frac": (4 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
            "read_write_frac": (8 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
        }

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
