"""Lists of volumes: the caller side of the hot path, sharded over GPUs.

The reference treats a batch as a Python loop over independent tensors
(src/imgcompressionmps/evaluation/benchmark.py:58-77 ``conv_to_mps``, :80-100
``conv_to_tensors``, :103-118 ``compress_list``).  Independent volumes are the unit of
multi-GPU parallelism: one process per GPU, every rank encodes / truncates /
reconstructs its own shard, and there is NO collective on the data path.  The only
communication is optional and off the critical path: gathering the (small) cores or the
reconstructed volumes to every rank with RCCL all_gather (``backend="nccl"`` on ROCm),
or with gloo on CPU tensors in the tests.

Same function names and argument meaning as the reference where one exists.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Block partition of ``range(n_items)``: rank r owns a contiguous run, sizes differ by
    at most one, concatenating the shards in rank order restores the original order."""
    if world_size < 1 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def conv_to_mps(tensor_list: Sequence, mode: str = "DCT", norm: bool = False, max_bond=None,
                cutoff: float = 1e-10, device=None):
    """benchmark.py:58-77: encode every tensor of the (local) list (default mode "DCT" as there)."""
    from .ndmps import NDMPS

    return [NDMPS.from_tensor(t, norm=norm, mode=mode, max_bond=max_bond, cutoff=cutoff, device=device)
            for t in tensor_list]


def _split(n_items: int, groups: int):
    groups = max(1, min(groups, n_items))
    return [shard_indices(n_items, g, groups) for g in range(groups)]


_group_streams = {}


def group_streams(n: int):
    """n long-lived HIP streams of the current device on different hardware queues
    (``ndmps_streams_create`` measures the binding).  Long-lived for two reasons: torch's caching
    allocator pools memory per stream (a fresh stream per call would hipMalloc ~6 GB of workspace per
    group every time), and the queue binding is decided once, at a stream's first use."""
    import ctypes as C

    import torch

    from .. import _lib

    key = (torch.cuda.current_device(), n)
    if key not in _group_streams:
        lib = _lib.load()
        raw = (C.c_void_p * n)()
        found = C.c_int(0)
        _lib.check(lib.ndmps_streams_create(n, raw, C.byref(found)))
        _group_streams[key] = [torch.cuda.ExternalStream(int(raw[i])) for i in range(n)]
    return _group_streams[key]


def encode_decode_concurrent(tensor_list: Sequence, groups: int = 4, mode: str = "Std", norm: bool = False,
                             max_bond=None, cutoff: float = 1e-10, reconstruct: bool = True, pool=None):
    """Throughput path for a list of same-shape device volumes: the list is cut into ``groups``
    contiguous groups; every group runs on its own host thread and HIP stream and encodes its
    volumes in lockstep (``NDMPS.from_tensors``).  The eigen-solver phases of a group keep only a
    fraction of the chip busy, so several groups in flight overlap them with each other's
    streaming phases (measured on MI355X: 8 volumes in flight 3.1, 16 -> 4.5, 32 -> 5.3 Gvoxel/s).
    Returns (list of NDMPS, list of reconstructions or None) in input order; every group's work has
    completed on the device when the call returns."""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from .ndmps import NDMPS

    tensor_list = list(tensor_list)
    parts = _split(len(tensor_list), groups)
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)

    device_index = torch.cuda.current_device()
    streams = group_streams(len(parts))

    def work(slot):
        idx = parts[slot]
        torch.cuda.set_device(device_index)  # pool threads start on device 0
        stream = streams[slot]
        with torch.cuda.stream(stream):
            stream.wait_event(ready)  # inputs produced on the caller's stream
            objs = NDMPS.from_tensors([tensor_list[i] for i in idx], norm=norm, mode=mode, max_bond=max_bond,
                                      cutoff=cutoff)
            recs = [o.to_tensor(as_torch=True) for o in objs] if reconstruct else None
        # host-side completion: a device-side wait on the caller's stream would sit in whichever
        # hardware queue that stream shares with a group and hold that group's next launches behind it
        stream.synchronize()
        return objs, recs

    own_pool = pool is None
    if own_pool:
        pool = ThreadPoolExecutor(len(parts))
    try:
        results = list(pool.map(work, range(len(parts))))
    finally:
        if own_pool:
            pool.shutdown()
    objs, recs = [], []
    for o, r in results:
        objs.extend(o)
        if reconstruct:
            recs.extend(r)
    return objs, (recs if reconstruct else None)


def conv_to_tensors(mps_list: Sequence, as_torch: bool = False):
    """benchmark.py:80-100: reconstruct every NDMPS of the (local) list."""
    return [m.to_tensor(as_torch=as_torch) for m in mps_list]


def compress_list(mps_list: Sequence, cutoff: float, max_bond=None) -> None:
    """benchmark.py:103-118: in-place truncation of every NDMPS of the (local) list."""
    if cutoff is None:
        raise ValueError("compression_factors must not be None")
    for m in mps_list:
        m.compress(cutoff, max_bond=max_bond)


def benchmark_metric(mps_list, reference_list=None, metric="compression_ratio", dtype=np.uint16):
    """benchmark.py:121-146, metric by metric, quirks included: ``ssim`` / ``psnr`` are called as
    f(reconstruction, original) (so the clip at 0 lands on the original and PSNR's peak is the
    reconstruction's), and ``gzip_ratio`` quantises the cores in place (``replace=True``)."""
    from ..utils.metrics import compute_overlap, compute_psnr, compute_ssim_by_dim

    metric_fn = {
        "compression_ratio": lambda mps, _: mps.compression_ratio(),
        "storage": lambda mps, _: mps.get_storage_space(dtype),
        "gzip_bytes": lambda mps, _: mps.get_bytesize_on_disk(dtype=dtype),
        "gzip_ratio": lambda mps, _: mps.compression_ratio_on_disk(dtype=dtype, replace=True),
        "ssim": lambda mps, ref: compute_ssim_by_dim(mps.to_tensor(as_torch=True), ref),
        "psnr": lambda mps, ref: compute_psnr(mps.to_tensor(as_torch=True), ref),
        "bond_dims": lambda mps, _: mps.bond_sizes(),
        "shape": lambda _, ref: tuple(ref.shape),
        "fidelity": lambda mps, ref: compute_overlap(mps, ref),
    }
    if metric not in metric_fn:
        raise ValueError(f"Unsupported metric: {metric}")
    if reference_list and len(reference_list) != len(mps_list):
        raise IndexError("Length mismatch: reference_list and mps_list must have the same length.")
    results = []
    for i, mps in enumerate(mps_list):
        ref = reference_list[i] if reference_list else None
        results.append(metric_fn[metric](mps, ref))
    return results


def run_benchmark(mps_list, original_tensors_list, cutoff_list, verbose=True):
    """benchmark.py:149-194: all metrics before compression and after every (cumulative) cutoff;
    same result keys, order and array layout ((n_files, 1 + n_cutoffs), ``bond_dims`` left as lists)."""
    from copy import deepcopy

    original_mps_list = deepcopy(mps_list)
    metrics = [
        ("ssim", original_tensors_list),
        ("compression_ratio", None),
        ("bond_dims", None),
        ("psnr", original_tensors_list),
        ("fidelity", original_mps_list),
        ("storage", None),
        ("gzip_bytes", None),
        ("gzip_ratio", None),
    ]
    results = {name: [] for name, _ in metrics}
    for name, ref in metrics:
        results[name].append(benchmark_metric(mps_list, ref, metric=name))
    for i, cutoff in enumerate(cutoff_list):
        if verbose:
            print(f"Status: {100 * (i + 1) / len(cutoff_list):.2f}% - Cutoff: {cutoff}")
        compress_list(mps_list, cutoff)
        for name, ref in metrics:
            results[name].append(benchmark_metric(mps_list, ref, metric=name))
    for key in results:
        if not results[key] or not results[key][0]:
            results[key] = []
        elif key != "bond_dims" and isinstance(results[key][0], (list, np.ndarray)) and np.ndim(results[key][0]) > 0:
            results[key] = np.array(results[key]).T
    return results


def run_full_benchmark(dataset_path, cutoff_list, result_file, datatype="MRI", mode="DCT", start=0, end=-1,
                       ending=".gz", shape=None):
    """benchmark.py:197-242: load every file of a dataset directory, encode, run the cutoff sweep and
    write the result JSON (same keys, same order; relative result paths land under
    ``src/evaluation/results`` like the reference's).  Returns the result dictionary."""
    import json
    from pathlib import Path

    from ..utils.loaders import find_specific_files, get_shapes, load_tensors, mri_to_slices

    dataset_path = Path(dataset_path)
    result_path = Path(result_file)
    if not result_path.is_absolute() and not str(result_path).startswith("src/evaluation/results"):
        result_path = Path("src/evaluation/results") / result_path
    files = find_specific_files(dataset_path, ending)
    files = files[start:] if end == -1 else files[start:end]
    if not files:
        raise FileNotFoundError(f"No files with extension {ending} found in {dataset_path}")
    data_list, bitsize_list = load_tensors(files, ending, shape)
    if datatype == "MRI_Slice":
        data_list, bitsize_list = mri_to_slices(data_list, bitsize_list)
    mps_list = conv_to_mps(data_list, mode)
    print("Starting benchmark...")
    metrics = run_benchmark(mps_list, data_list, cutoff_list)
    print(f"Saving results to {result_path}")
    result_dict = {
        "datatype": datatype,
        "mode": mode,
        "files": files,
        "cutoff_list": cutoff_list.tolist(),
        "bitsize_list": bitsize_list,
        "shapes": get_shapes(data_list),
        **{k: v.tolist() if hasattr(v, "tolist") else v for k, v in metrics.items()},
    }
    result_path.parent.mkdir(parents=True, exist_ok=True)
    with open(result_path, "w") as f:
        json.dump(result_dict, f, indent=2)
    return result_dict


# ------------------------------------------------------------------------ collectives
def _dist():
    import torch.distributed as dist

    return dist


def pack_cores(core_tensors: Sequence):
    """Flatten a list of core tensors into (flat fp32 tensor, int64 shape table (n, 3))."""
    import torch

    shapes = torch.tensor([list(c.shape) for c in core_tensors], dtype=torch.int64).reshape(-1, 3)
    flat = torch.cat([c.reshape(-1).to(torch.float32) for c in core_tensors]) if len(core_tensors) else \
        torch.zeros(0, dtype=torch.float32)
    return flat, shapes


def unpack_cores(flat, shapes):
    out, off = [], 0
    for s in shapes.tolist():
        n = int(np.prod(s))
        out.append(flat[off:off + n].reshape(*s))
        off += n
    return out


def all_gather_cores(local_cores: Sequence[Sequence], group=None):
    """Every rank receives the cores of every volume, in global volume order.

    ``local_cores``: for each local volume, its list of (chi, d, chi') tensors (CPU tensors with
    gloo, device tensors with nccl/RCCL).  Cores are tiny (<= ~1.4 % of a volume), so
    this is two small all_gathers: the shape tables, then the padded flat payloads.
    """
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    dev = local_cores[0][0].device if local_cores and local_cores[0] else torch.device("cpu")
    flat_parts, shape_parts, counts = [], [], []
    for cores in local_cores:
        f, s = pack_cores(cores)
        flat_parts.append(f.to(dev))
        shape_parts.append(s.to(dev))
        counts.append(len(cores))
    flat = torch.cat(flat_parts) if flat_parts else torch.zeros(0, dtype=torch.float32, device=dev)
    shapes = torch.cat(shape_parts) if shape_parts else torch.zeros((0, 3), dtype=torch.int64, device=dev)
    counts_t = torch.tensor(counts, dtype=torch.int64, device=dev)

    # exchange sizes: [n_volumes, n_cores, n_elements]
    meta = torch.tensor([len(counts), shapes.shape[0], flat.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_vol = max(int(m[0]) for m in metas)
    max_cores = max(int(m[1]) for m in metas)
    max_elems = max(int(m[2]) for m in metas)

    def padded(t, n, fill_shape):
        buf = torch.zeros((n,) + fill_shape, dtype=t.dtype, device=dev)
        buf[: t.shape[0]] = t
        return buf

    g_counts = [torch.zeros(max_vol, dtype=torch.int64, device=dev) for _ in range(world)]
    g_shapes = [torch.zeros((max_cores, 3), dtype=torch.int64, device=dev) for _ in range(world)]
    g_flat = [torch.zeros(max_elems, dtype=torch.float32, device=dev) for _ in range(world)]
    dist.all_gather(g_counts, padded(counts_t, max_vol, ()), group=group)
    dist.all_gather(g_shapes, padded(shapes, max_cores, (3,)), group=group)
    dist.all_gather(g_flat, padded(flat, max_elems, ()), group=group)

    result = []
    for r in range(world):
        n_vol, n_cores, n_el = (int(v) for v in metas[r])
        cores = unpack_cores(g_flat[r][:n_el], g_shapes[r][:n_cores])
        off = 0
        for c in g_counts[r][:n_vol].tolist():
            result.append(cores[off:off + c])
            off += c
    return result


def all_gather_volumes(local_volumes: Sequence, n_total: int, group=None):
    """Every rank receives every reconstructed volume (same shape), in global order."""
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per_rank = [len(shard_indices(n_total, r, world)) for r in range(world)]
    if len(local_volumes) != per_rank[rank]:
        raise ValueError("local shard does not match shard_indices")
    width = max(per_rank)
    ref = local_volumes[0] if local_volumes else None
    shape_t = torch.tensor(list(ref.shape) if ref is not None else [], dtype=torch.int64)
    if ref is None:
        raise ValueError("all_gather_volumes needs at least one local volume per rank")
    buf = torch.zeros((width,) + tuple(ref.shape), dtype=ref.dtype, device=ref.device)
    for i, v in enumerate(local_volumes):
        buf[i] = v
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    out = []
    for r in range(world):
        out.extend(gathered[r][i] for i in range(per_rank[r]))
    return out
