#!/usr/bin/env python
"""Headline benchmark: Mvoxels/s of compress + reconstruct (NDMPS.from_tensor with the bond cap
applied in the sweep, then NDMPS.to_tensor) on a synthetic 256^3 fp32 volume at chi = 64
(BASELINE.json "metric"), input and output resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; a step is one batch of --batch independent volumes per GPU (encoded in
lockstep by NDMPS.from_tensors, reconstructed one by one); independent volumes shard with no
data-path collective -> weak scaling.  Rank 0 prints one JSON line; `single_volume`
in it is the latency of a batch of one.  At N=1 it also carries the CPU baseline (the NumPy
oracle on one volume of the batch, timed on the host cores) and the SSIM gap between the GPU and
the oracle reconstruction of that volume.
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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=256, help="edge of the cubic volume")
    ap.add_argument("--chi", type=int, default=64)
    ap.add_argument("--mode", default="Std", choices=["Std", "DCT"])
    ap.add_argument("--batch", type=int, default=64, help="independent volumes per GPU per step")
    ap.add_argument("--groups", type=int, default=4,
                    help="concurrent groups (host thread + HIP stream each) the batch is cut into; "
                         "volumes of a group are encoded in lockstep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="profiling aid: no single-volume phase")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + --share-gpu rehearses N ranks on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from imgcompressionmps_amd import NDMPS, _lib
    from imgcompressionmps_amd.core import ndmps as ndmps_mod

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    _lib.load()
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    from oracle.metrics import compute_ssim_by_dim, synthetic_mri  # data generator + checker only

    shape = (args.size,) * 3
    n_vox = int(np.prod(shape))
    # one batch of independent volumes per GPU (seeded, distinct per rank and per slot)
    # 16 volumes are generated on the host (the generator is NumPy and takes ~0.5 s per 256^3 volume);
    # further slots are distinct device-side variants of them: rolled by 8 * (j // 16) voxels along every
    # axis and mirrored along the first axis for odd multiples, same statistics, different voxels at
    # every position
    x_host = synthetic_mri(shape, seed=2025 + rank)
    n_host = min(args.batch, 16)
    xs = [torch.from_numpy(x_host).to(device)]
    for j in range(1, n_host):
        xs.append(torch.from_numpy(synthetic_mri(shape, seed=2025 + 1000 * j + rank)).to(device))
    for j in range(n_host, args.batch):
        rep = j // n_host
        v = torch.roll(xs[j % n_host], shifts=(8 * rep,) * len(shape), dims=tuple(range(len(shape))))
        xs.append((torch.flip(v, dims=(0,)) if rep % 2 else v).contiguous())
    x = xs[0]

    from concurrent.futures import ThreadPoolExecutor

    from imgcompressionmps_amd.core import batch as batch_mod

    pool = ThreadPoolExecutor(max(1, min(args.groups, args.batch)))

    def step():
        objs, recs = batch_mod.encode_decode_concurrent(xs, groups=args.groups, mode=args.mode,
                                                         max_bond=args.chi, pool=pool)
        return objs[0], recs[0]

    def group_step(n):  # one lockstep group of n volumes on the current stream
        objs = NDMPS.from_tensors(xs[:n], mode=args.mode, max_bond=args.chi)
        return [o.to_tensor(as_torch=True) for o in objs]

    def single_step():
        o = NDMPS.from_tensor(x, mode=args.mode, max_bond=args.chi)
        return o, o.to_tensor(as_torch=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timer = ndmps_mod.StageTimer()
    ndmps_mod.set_stage_timer(timer)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        obj, rec = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ndmps_mod.set_stage_timer(None)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stages = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps}
              for k, v in timer.totals_ms().items()}
    value = world * args.batch * n_vox * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # reference points, same kernels, not part of `value`: one lockstep group of 8 on one stream
    # (with per-stage device times undisturbed by concurrent groups) and a single volume
    single_ms = group8_ms = float("nan")
    stages_group8 = {}
    if not args.skip_single:
        n8 = min(8, args.batch)
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
        for _ in range(3):
            single_step()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t0) / 3 * 1e3

    # roofline of the reshape stage (the HBM-bound kernel north_star names): the tiled encode_permute
    # kernel reads 4 B and writes 4 B per voxel, one launch per volume.  Its launch duration is taken
    # from the HIP events of the one-group phase of THIS run (nothing else on the GPU); the events of
    # the concurrent main region also count the time a launch shares HBM with, or queues behind, the
    # other groups' kernels and are reported next to it.
    def per_launch_ms(st):
        e = st.get("encode_permute", {})
        return e.get("ms_per_step", float("nan")) / max(e.get("launches_per_step", 1.0), 1.0)

    perm_ms_concurrent = per_launch_ms(stages)
    perm_ms = per_launch_ms(stages_group8) if stages_group8 else perm_ms_concurrent
    algo_bytes = 2 * 4 * n_vox
    achieved = algo_bytes / (perm_ms * 1e-3) / 1e9
    roofline = {
        "kernel": "encode_tiled_kernel<uint32, vec>",
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE 66 019 KB + WRITE_SIZE 65 536 KB,
        # profiles/r01_c_pmc_*.txt; this kernel's sorted gather is tallied exactly, checked against its
        # known 64 MiB read); only valid for the default 256^3 volume
        "traffic": (66019 + 65536) * 1024.0 if args.size == 256 else None,
        "read_frac": (4 * n_vox / (perm_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS,
        "launch_us": perm_ms * 1e3,
        "launch_us_in_concurrent_region": perm_ms_concurrent * 1e3,
        "end_to_end_algorithmic_GBps": args.batch * algo_bytes / (ms_per_step * 1e-3) / 1e9,
    }

    # the kernel with the largest share of device time (profiles/): one outer step of the batched block
    # Jacobi eigen-solver.  Timed live on 16 Gram matrices of order 512 built from the batch's volumes
    # (the shape of the three big eigenproblems of every 256^3 / chi = 64 volume), values phase only:
    # HIP-event time / launches, so the figure includes the per-sweep convergence check.
    eig_step = None
    if not args.skip_single and args.size >= 64:
        import ctypes as C

        lib = _lib.load()
        nb_eig, n_eig = 16, 512
        a = torch.stack([xs[j % len(xs)].reshape(n_eig, -1).to(torch.float64) for j in range(nb_eig)])
        g0 = torch.bmm(a, a.transpose(1, 2)).contiguous()
        del a
        vv = torch.empty_like(g0)
        ww = torch.empty((nb_eig, n_eig), dtype=torch.float64, device=device)
        nbytes = lib.ndmps_syevj_batched_workspace_bytes(n_eig, nb_eig)
        wsb = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        sw = (C.c_int * nb_eig)()
        sizes = _lib.i64_array([n_eig] * nb_eig)
        times = []
        for it in range(3):
            g = g0.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.ndmps_syevj_batched_values_f64(nb_eig, g.data_ptr(), n_eig * n_eig, sizes, vv.data_ptr(),
                                                          n_eig * n_eig, ww.data_ptr(), n_eig, 1e-13, wsb.data_ptr(),
                                                          nbytes, sw, _lib.stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        sweeps = max(sw)
        nblk = n_eig // 16
        launches = sweeps * (nblk - 1) + 1
        step_us = min(times[1:]) * 1e3 / launches
        pairs = nblk // 2
        # per launch and matrix: the block-upper half of G is read once and written once (fp64), every
        # 32x32 rotation block is written once (history) and read by the tiles of its pair row / column
        step_bytes = nb_eig * (n_eig * n_eig * 8 + 2 * pairs * 32 * 32 * 8)
        step_flops = nb_eig * (pairs * (pairs - 1) // 2) * 2 * (2 * 32 ** 3)  # two 32^3 products per upper tile
        eig_step = {
            "kernel": "blk_step_kernel<16>",
            "workload": f"{nb_eig} x ({n_eig} x {n_eig}) fp64 Gram matrices in lockstep, one stream",
            "sweeps": int(sweeps),
            "launches": int(launches),
            "launch_us": step_us,
            "hbm": {"achieved": step_bytes / (step_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": step_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "bytes_per_launch": step_bytes,
                    # rocprofv3 PMC passes on this very workload (tools/role_probe.py 16 512): FETCH_SIZE
                    # 16 596 KB + WRITE_SIZE 20 436 KB per launch (profiles/r01_g_pmc_step_kernel_*.txt)
                    "traffic": (16596 + 20436) * 1024.0},
            "mfma_f64": {"achieved": step_flops / (step_us * 1e-6) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": step_flops / (step_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         "flops_per_launch": step_flops},
            "note": "latency-bound when one group runs alone (a step is a dependent chain of ~25 us); with 4 groups "
                    "in flight the same launches overlap and the aggregate rate is ~2x these figures",
        }
        del g0, vv, ww, wsb

    line = {
        "metric": "Mvoxels/s compress+reconstruct, 256^3 fp32, bond chi=64; SSIM vs ref",
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
            "workload": f"batch of {args.batch} independent {args.size}^3 fp32 synthetic MRI volumes per GPU per step "
                        f"({args.groups} concurrent groups, lockstep inside a group), "
                        f"NDMPS.from_tensors(max_bond={args.chi}, mode={args.mode}) + to_tensor each, "
                        f"device-resident in/out",
            "volumes_per_step": world * args.batch,
            "volume_source": "16 seeded synthetic-MRI volumes per rank from the host generator; further slots "
                             "are rolled / mirrored device-side variants of them",
            "batch_per_gpu": args.batch,
            "groups_per_gpu": args.groups,
            "bonds": obj.bond_sizes(),
            "parallelism": f"{world} independent volume shard(s), no data-path collective",
        },
        "roofline": roofline,
        "roofline_eig_step": eig_step,
        "stages": stages,
        "single_volume": {"ms": single_ms, "Mvoxels_per_s": n_vox / single_ms / 1e3},
        "one_group_of_8": {"ms": group8_ms, "Mvoxels_per_s": min(8, args.batch) * n_vox / group8_ms / 1e3,
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
        ref = OracleNDMPS.from_tensor(x_host, mode=args.mode, max_bond=args.chi, materialise_map=False)
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
            "sample": f"one {args.size}^3 volume, NumPy fp64 oracle (closed-form index permutation, SVD sweep "
                      f"with max_bond={args.chi}, chain contraction), {cpu_s:.1f} s wall",
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
