"""Lists of volumes: the caller side of the hot path, sharded over GPUs.

The reference treats a batch as a Python loop over independent tensors
(src/imgcompressionmps/evaluation/benchmark.py:58-77 ``conv_to_mps``, :80-100
``conv_to_tensors``, :103-118 ``compress_list``).  Independent volumes are the unit of
multi-GPU parallelism: one process per GPU, every rank encodes / truncates /
reconstructs its own shard, and there is NO collective on the data path.  The only
communication is optional and off the critical path: gathering the (small) cores or the
reconstructed volumes to every rank with RCCL all_gather (``backend="nccl"`` on ROCm),
or with gloo on CPU tensors in the tests.

``conv_to_mps`` / ``conv_to_tensors`` / ``compress_list`` / ``benchmark_metric`` / ``run_benchmark``
keep the reference's names, argument meaning and result layout (they are what a caller of the
reference switches over); their bodies are this repo's own.  The reference's dataset plumbing
(``run_full_benchmark``, file discovery, NIfTI loading through nibabel) is out of scope.
"""
from __future__ import annotations

import os
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
    """benchmark.py:58-77: encode every tensor of the (local) list (default mode "DCT" as there).  The reference loops
    volume by volume; a list of same-shape tensors goes through the lockstep path in chunks of up to 64 volumes, chunk after
    chunk as a stream (``NDMPS.from_tensors_begin`` on three alternating streams: the objects of a chunk are built while the
    next ones run) -- results equal ``from_tensor`` on each up to the rounding of the fp64 eigen-solver, the order of the
    list is kept.  Lists of mixed shapes keep the loop."""
    import numpy as np

    from .ndmps import NDMPS

    from .. import _lib

    _lib.require_device()  # the product never computes on the CPU: the same loud failure as from_tensor
    tensor_list = list(tensor_list)
    shapes = {tuple(np.shape(t)) for t in tensor_list}
    if len(tensor_list) < 2 or len(shapes) != 1 or () in shapes:
        return [NDMPS.from_tensor(t, norm=norm, mode=mode, max_bond=max_bond, cutoff=cutoff, device=device)
                for t in tensor_list]
    import torch

    shape = next(iter(shapes))
    numel = int(np.prod(shape))
    chunk = int(max(1, min(64, (8 << 30) // (4 * numel))))  # at most 8 GiB of fp32 volumes in a chunk
    # ... and a sweep workspace of at most 24 GiB: an exact sweep (no bond cap, the reference's default) solves eigenproblems
    # of the full bond dimensions -- order 4096 in the middle of a 256^3 volume -- and asks for gigabytes per volume
    from .ndmps import _plan_for

    lib = _lib.load()
    dims = [int(q) for q in _plan_for(tuple(int(v) for v in shape), torch.cuda.current_device()).qubit_size]
    per_volume = int(lib.ndmps_tt_sweep_batched_workspace_bytes(1, len(dims), _lib.i64_array(dims), int(max_bond or 0)))
    if per_volume > 0:
        chunk = int(max(1, min(chunk, (24 << 30) // per_volume)))
    _, lanes = default_stream_shape(chunk)
    streams = group_streams(lanes)
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    out, in_flight = [], []
    for k, i0 in enumerate(range(0, len(tensor_list), chunk)):
        stream = streams[k % lanes]
        with torch.cuda.stream(stream):
            stream.wait_event(ready)  # inputs produced on the caller's stream
            in_flight.append((NDMPS.from_tensors_begin(tensor_list[i0:i0 + chunk], norm=norm, mode=mode, max_bond=max_bond,
                                                       cutoff=cutoff, device=device), stream))
        while len(in_flight) > lanes:
            out.extend(in_flight.pop(0)[0].result())
    for pending, stream in in_flight:
        out.extend(pending.result())
        stream.synchronize()  # like the loop, the call returns with everything done on the device
    for s in streams:
        s.synchronize()
    return out


def default_groups(n_items: int) -> int:
    """Concurrent lockstep groups for ``n_items`` same-shape volumes on one GPU: two from 32 volumes on (each group's
    single-workgroup solver kernels then hide under the other's throughput kernels: 64 volumes of 256^3 28 ms against
    32), one below (two groups of 12 take 17.4 ms per step where one of 24 takes 15.6: the resident
    tridiagonalisations of the groups exclude each other and cost the same for 12 matrices as for 24)."""
    return 2 if n_items >= 32 else 1


def default_lanes(groups: int, eig_order: int = 512) -> int:
    """Sets of streams that the consecutive batches of a STREAM of batches alternate between (``encode_decode_begin``):
    batch k runs on lane k mod lanes, so consecutive batches overlap on the GPU the way the groups of one batch do.
    Measured on one MI355X (256^3 chi = 64, ms per batch, one / two / three lanes): 8 volumes 9.7 / 6.8 / 6.0, 16 volumes
    12.7 / 8.2 / 7.3, 31 volumes 16.9 / 13.4 / 12.7; 128^3 chi = 32: 8 volumes 3.2 / 1.8 / 1.3.  Batches of two groups (from
    32 volumes on) gain from a second lane only while the resident tridiagonalisations of a group take half a turn
    (eigenproblems of order <= 256: 64 volumes of 256^3 chi = 32 12.1 -> 10.8 ms, of 128^3 4.9 -> 3.7 ms); four groups of
    32 order-512 matrices in flight -- four full turns, five streams on four hardware queues -- collapse (27 -> 118 ms
    per batch), so those keep one lane."""
    if groups <= 1:
        return 3
    return 2 if eig_order <= 256 else 1


def default_stream_shape(n_items: int, eig_order: int = 512):
    """(groups, lanes) for a STREAM of batches of ``n_items`` same-shape volumes (``encode_decode_begin`` batch after batch):
    ONE lockstep group on three lanes at every size measured -- three batches in different phases overlap better than the two
    groups of one batch, and every launch serves the whole batch.  Measured, 64 volumes per batch, ms per batch (groups,
    lanes): 256^3 chi = 64: (2, 1) 27.0, (1, 3) 25.1 - 25.7, (1, 4) 26.2; 256^3 chi = 32: (2, 1) 12.1, (2, 2) 10.7, (1, 3) 9.9;
    128^3 chi = 32: (2, 1) 4.9, (2, 2) 4.5, (1, 3) 4.1 - 4.5.  One call that waits for its batch (encode_decode_concurrent)
    keeps default_groups."""
    return 1, 3


def _split(n_items: int, groups: int):
    groups = max(1, min(groups, n_items))
    return [shard_indices(n_items, g, groups) for g in range(groups)]


_group_streams = {}
_group_pools = {}


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


def encode_decode_concurrent(tensor_list: Sequence, groups: int = None, mode: str = "Std", norm: bool = False,
                             max_bond=None, cutoff: float = 1e-10, reconstruct: bool = True, pool=None,
                             wait: bool = True):
    """Throughput path for a list of same-shape device volumes: the list is cut into ``groups``
    contiguous groups; every group runs on its own host thread and HIP stream and encodes its
    volumes in lockstep (``NDMPS.from_tensors``).  The eigen-solver phases of a group keep only a
    fraction of the chip busy, so several groups in flight overlap them with each other's
    streaming phases; ``groups=None``: ``default_groups(len(tensor_list))``.
    Returns (list of NDMPS, list of reconstructions or None) in input order; every group's work has
    completed on the device when the call returns -- unless ``wait=False``: then the reconstructions are
    still being written on the groups' streams when the call returns (the NDMPS objects are complete) and the
    caller synchronises the device before reading them; a following call queues behind them on the same
    streams, so consecutive batches run back to back without the host in between."""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from .ndmps import NDMPS

    tensor_list = list(tensor_list)
    if groups is None:
        groups = default_groups(len(tensor_list))
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
            if reconstruct:
                objs, recs = NDMPS.from_tensors([tensor_list[i] for i in idx], norm=norm, mode=mode, max_bond=max_bond,
                                                cutoff=cutoff, reconstruct=True)
            else:
                objs, recs = NDMPS.from_tensors([tensor_list[i] for i in idx], norm=norm, mode=mode,
                                                max_bond=max_bond, cutoff=cutoff), None
        # host-side completion: a device-side wait on the caller's stream would sit in whichever
        # hardware queue that stream shares with a group and hold that group's next launches behind it
        if wait:
            stream.synchronize()
        return objs, recs

    if pool is None:
        # long-lived workers: the solver keeps per-thread pinned flags and events (csrc/eig_block.hip),
        # a fresh pool per call would allocate them again every time
        pool = _group_pools.get(len(parts))
        if pool is None:
            pool = _group_pools.setdefault(len(parts), ThreadPoolExecutor(len(parts), thread_name_prefix="ndmps-group"))
    results = list(pool.map(work, range(len(parts))))
    objs, recs = [], []
    for o, r in results:
        objs.extend(o)
        if reconstruct:
            recs.extend(r)
    return objs, (recs if reconstruct else None)


class PendingBatch:
    """What ``encode_decode_begin`` returns: the groups of one batch, enqueued.  ``result()`` -> (objects, reconstructions)
    in input order, like ``encode_decode_concurrent(wait=False)``: the objects are complete, the reconstructions may still
    be being written on the groups' streams."""

    def __init__(self, groups, reconstruct):
        self._groups, self._reconstruct, self._value = groups, reconstruct, None

    def result(self):
        if self._value is None:
            objs, recs = [], []
            for g in self._groups:
                r = g.result()
                if self._reconstruct:
                    objs.extend(r[0])
                    recs.extend(r[1])
                else:
                    objs.extend(r)
            self._value = (objs, recs if self._reconstruct else None)
            self._groups = None
        return self._value


def encode_decode_begin(tensor_list: Sequence, groups: int = None, mode: str = "Std", norm: bool = False, max_bond=None,
                        cutoff: float = 1e-10, reconstruct: bool = True, pool=None, lane: int = 0, lanes: int = 1):
    """``encode_decode_concurrent(wait=False)`` in two halves, for a stream of batches: this call ENQUEUES the batch
    (every group on its own host thread and stream, ``NDMPS.from_tensors_begin``) and returns a ``PendingBatch``; its
    ``result()`` builds the objects.  A caller that begins batch k + 1 before it asks for the result of batch k has the
    objects of k built while k + 1 runs -- the Python section behind a sweep (ranks back, 32 objects per group) and the
    set-up in front of the next one otherwise leave the GPU idle at every batch boundary (1.2 ms of a 27 ms step at 64
    volumes of 256^3, tools/gap_report.py).  Same kernels in the same order as the one-call form: results are
    bit-identical -- except for batches of five to eight volumes on MORE than one lane, whose resident tridiagonalisations
    take 32-column blocks then (ndmps_syevd_topk_set_streamed): equal up to the last bits of the fp64 eigen-solver.
    ``lane`` / ``lanes``: a stream of SMALL batches (one group each) may alternate between ``lanes`` sets of streams --
    batch k on lane k mod lanes -- so that consecutive batches overlap on the GPU the way two groups of one batch do."""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from .. import _lib
    from .ndmps import NDMPS

    tensor_list = list(tensor_list)
    if groups is None:
        groups = default_groups(len(tensor_list))
    parts = _split(len(tensor_list), groups)
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    device_index = torch.cuda.current_device()
    lanes = max(1, int(lanes))
    streams = group_streams(len(parts) * lanes)[(int(lane) % lanes) * len(parts):]

    def work(slot):
        idx = parts[slot]
        torch.cuda.set_device(device_index)  # pool threads start on device 0
        lib = _lib.load()
        # several lanes: this batch shares the GPU with its neighbours -- the solver then sizes small batches' resident
        # launches for overlap, not for the latency of one batch (per host thread; restored behind the call)
        was = lib.ndmps_syevd_topk_set_streamed(1 if lanes > 1 else 0)
        try:
            with torch.cuda.stream(streams[slot]):
                streams[slot].wait_event(ready)  # inputs produced on the caller's stream
                return NDMPS.from_tensors_begin([tensor_list[i] for i in idx], norm=norm, mode=mode, max_bond=max_bond,
                                                cutoff=cutoff, reconstruct=reconstruct)
        finally:
            lib.ndmps_syevd_topk_set_streamed(was)

    if pool is None:
        pool = _group_pools.get(len(parts))
        if pool is None:
            pool = _group_pools.setdefault(len(parts), ThreadPoolExecutor(len(parts), thread_name_prefix="ndmps-group"))
    return PendingBatch(list(pool.map(work, range(len(parts)))), reconstruct)


def encode_decode_stream(batches, groups: int = None, lanes: int = None, mode: str = "Std", norm: bool = False, max_bond=None,
                         cutoff: float = 1e-10, reconstruct: bool = True, pool=None):
    """Generator over an iterable of batches (lists of same-shape device volumes; what a dataset driver's loop over chunks
    of its list produces, evaluation/benchmark.py:58-100): yields ``(objects, reconstructions)`` per batch, in input order.
    Batch k is enqueued on lane k mod ``lanes`` (``encode_decode_begin``) before the result of batch k - lanes is asked for,
    so object construction runs under the following batches' kernels and consecutive batches overlap on the GPU;
    ``groups`` / ``lanes`` default to ``default_stream_shape`` of the first batch.  The reconstructions of a yielded batch may
    still be being written on its streams: synchronise the device (or the streams) before reading them on the host."""
    in_flight = []
    for k, chunk in enumerate(batches):
        chunk = list(chunk)
        if groups is None or lanes is None:
            g0, l0 = default_stream_shape(len(chunk))
            groups = g0 if groups is None else groups
            lanes = l0 if lanes is None else lanes
        in_flight.append(encode_decode_begin(chunk, groups=groups, mode=mode, norm=norm, max_bond=max_bond, cutoff=cutoff,
                                             reconstruct=reconstruct, pool=pool, lane=k, lanes=lanes))
        while len(in_flight) > lanes:
            yield in_flight.pop(0).result()
    while in_flight:
        yield in_flight.pop(0).result()


def conv_to_tensors(mps_list: Sequence, as_torch: bool = False):
    """benchmark.py:80-100: reconstruct every NDMPS of the (local) list."""
    from .ndmps import NDMPS

    return NDMPS.to_tensors(mps_list, as_torch=as_torch)


def _map_on_lanes(fn, items, lanes: int = 3):
    """``[fn(item) for item in items]`` with the items dealt round robin to ``lanes`` host threads that have a HIP stream each
    (the group streams): what one item's calls leave idle -- host reads of ranks or metrics, small launches, gzip on the CPU --
    is filled by the others'.  Fewer than two items, or NDMPS_LIST_SERIAL=1: the plain loop on the caller's stream."""
    items = list(items)
    lanes = min(int(lanes), len(items))
    if lanes < 2 or os.environ.get("NDMPS_LIST_SERIAL") or os.environ.get("NDMPS_COMPRESS_LIST_SERIAL"):
        return [fn(item) for item in items]
    import torch
    from concurrent.futures import ThreadPoolExecutor

    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    device_index = torch.cuda.current_device()
    streams = group_streams(lanes)
    results = [None] * len(items)

    def work(slot):
        torch.cuda.set_device(device_index)  # pool threads start on device 0
        with torch.cuda.stream(streams[slot]):
            streams[slot].wait_event(ready)  # operands produced on the caller's stream
            for idx in range(slot, len(items), lanes):
                results[idx] = fn(items[idx])
        streams[slot].synchronize()

    pool = _group_pools.get(lanes)
    if pool is None:
        pool = _group_pools.setdefault(lanes, ThreadPoolExecutor(lanes, thread_name_prefix="ndmps-group"))
    list(pool.map(work, range(lanes)))
    return results


def compress_list(mps_list: Sequence, cutoff: float, max_bond=None) -> None:
    """benchmark.py:103-118: in-place truncation of every NDMPS of the (local) list.  A compress reads its ranks back bond by
    bond (a host round trip per bond), but the objects of a list are independent: from two objects on they are dealt to
    three host threads with a stream each (``_map_on_lanes``), so one object's eigen-decompositions run under another's
    products and host reads (same calls on the same operands per object: the result of each is what the loop gives)."""
    if cutoff is None:
        raise ValueError("compression_factors must not be None")
    _map_on_lanes(lambda m: m.compress(cutoff, max_bond=max_bond), mps_list)


# ------------------------------------------------------------------ quality-vs-ratio sweep
# Result keys of the reference's metric loop (evaluation/benchmark.py:149-194), in its order.  The keys
# and the (n_volumes, 1 + n_cutoffs) layout are the data format of the caller side; the evaluation
# below is this repo's own: one record per (volume, stage), the reconstruction computed once and shared
# by SSIM and PSNR, everything on the device.
QUALITY_KEYS = ("ssim", "compression_ratio", "bond_dims", "psnr", "fidelity", "storage", "gzip_bytes",
                "gzip_ratio")
_NEEDS_ORIGINAL = {"ssim", "psnr", "shape"}
_NEEDS_ORIGINAL_MPS = {"fidelity"}


def quality_record(obj, original=None, original_mps=None, dtype=np.uint16, keys=QUALITY_KEYS):
    """All requested figures of one NDMPS at its current truncation, as a dict.

    Conventions inherited from the reference's caller (benchmark.py:125-133), because they change
    the numbers: SSIM and PSNR take the RECONSTRUCTION as their first argument (the clip at 0 and the
    PSNR peak therefore refer to it), and ``gzip_ratio`` leaves the cores de-quantised in place
    (``replace=True``) -- it is evaluated last so the other figures of the stage see the cores as
    ``compress`` left them."""
    from ..utils import metrics as M

    rec = None
    out = {}
    for key in keys:
        if key in _NEEDS_ORIGINAL and original is None:
            raise ValueError(f"metric {key!r} needs the original tensor")
        if key in _NEEDS_ORIGINAL_MPS and original_mps is None:
            raise ValueError(f"metric {key!r} needs the untruncated NDMPS")
        if key in ("ssim", "psnr") and rec is None:
            rec = obj.to_tensor(as_torch=True)
    for key in sorted(keys, key=lambda k: k == "gzip_ratio"):  # stable: gzip_ratio moves to the end
        if key == "ssim":
            out[key] = M.compute_ssim_by_dim(rec, original)
        elif key == "psnr":
            out[key] = M.compute_psnr(rec, original)
        elif key == "fidelity":
            out[key] = M.compute_overlap(obj, original_mps)
        elif key == "compression_ratio":
            out[key] = obj.compression_ratio()
        elif key == "bond_dims":
            out[key] = obj.bond_sizes()
        elif key == "storage":
            out[key] = obj.get_storage_space(dtype)
        elif key == "gzip_bytes":
            out[key] = obj.get_bytesize_on_disk(dtype=dtype)
        elif key == "gzip_ratio":
            out[key] = obj.compression_ratio_on_disk(dtype=dtype, replace=True)
        elif key == "shape":
            out[key] = tuple(original.shape)
        else:
            raise ValueError(f"Unsupported metric: {key}")
    return out


def benchmark_metric(mps_list, reference_list=None, metric="compression_ratio", dtype=np.uint16):
    """One figure for every NDMPS of the list (caller-side counterpart of benchmark.py:121-146).
    ``reference_list`` holds the originals (ssim / psnr / shape) or the untruncated NDMPS (fidelity)."""
    if metric not in QUALITY_KEYS + ("shape",):
        raise ValueError(f"Unsupported metric: {metric}")
    if reference_list and len(reference_list) != len(mps_list):
        raise IndexError("Length mismatch: reference_list and mps_list must have the same length.")
    refs = reference_list if reference_list else [None] * len(mps_list)
    as_mps = metric in _NEEDS_ORIGINAL_MPS
    return _map_on_lanes(lambda mr: quality_record(mr[0], None if as_mps else mr[1], mr[1] if as_mps else None, dtype,
                                                   (metric,))[metric], list(zip(mps_list, refs)))


def run_benchmark(mps_list, original_tensors_list, cutoff_list, verbose=True, dtype=np.uint16):
    """Quality figures before truncation and after every cutoff of ``cutoff_list`` (applied
    cumulatively, in place), as ``{key: array (n_volumes, 1 + n_cutoffs)}`` with ``bond_dims`` kept as
    nested lists [stage][volume] -- the layout of benchmark.py:149-194."""
    from copy import deepcopy

    untouched = deepcopy(list(mps_list))
    stages = []

    def take_stage():
        # the volumes of a stage are independent: three at a time (reconstruction, SSIM / PSNR with their host reads, gzip)
        stages.append(_map_on_lanes(lambda t: quality_record(t[0], t[1], t[2], dtype),
                                    list(zip(mps_list, original_tensors_list, untouched))))

    take_stage()
    for n, cutoff in enumerate(cutoff_list, 1):
        if verbose:
            print(f"Status: {100 * n / len(cutoff_list):.2f}% - Cutoff: {cutoff}")
        compress_list(mps_list, cutoff)
        take_stage()
    results = {}
    for key in QUALITY_KEYS:
        table = [[rec[key] for rec in stage] for stage in stages]
        if not table[0]:
            results[key] = []
        elif key == "bond_dims":
            results[key] = table
        else:
            results[key] = np.array(table).T
    return results


# ------------------------------------------------------------------------ collectives
def _dist():
    import torch.distributed as dist

    return dist


def broadcast_job(job=None, src: int = 0, group=None, device=None):
    """Rank ``src`` sends the job descriptor (a JSON-serialisable dict: shape, chi, mode, batch, seeds, ...)
    to every rank; returns the dict on all ranks.  Two broadcasts (length, payload) of tensors that live on
    ``device`` (a HIP device for the nccl = RCCL backend, CPU for gloo).  This is the only collective a
    sharded run needs before it starts; the data path has none."""
    import json

    import torch

    dist = _dist()
    rank = dist.get_rank(group)
    dev = torch.device("cpu") if device is None else torch.device(device)
    if rank == src:
        if job is None:
            raise ValueError("the source rank must provide the job descriptor")
        payload = torch.tensor(list(json.dumps(job, sort_keys=True).encode("utf-8")), dtype=torch.uint8, device=dev)
        length = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    else:
        length = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(length, src=src, group=group)
    if rank != src:
        payload = torch.zeros(int(length.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(payload, src=src, group=group)
    return json.loads(bytes(payload.cpu().tolist()).decode("utf-8"))


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
