"""Multi-GPU row: independent volumes shard over ranks with no data-path
collective; the optional gathers are exercised here with world_size-2 gloo on CPU tensors
(no kernels run: the product has no CPU compute path, so the 'cores' below are synthetic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imgcompressionmps_amd.core import batch


def test_shard_indices_partition():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            shards = [batch.shard_indices(n, r, world) for r in range(world)]
            assert sum(shards, []) == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert len(batch.shard_indices(64, 3, 8)) == 8  # BASELINE config 4: 64 volumes over 8 GPUs
    with pytest.raises(ValueError):
        batch.shard_indices(4, 2, 2)


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    cores = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)) for s in [(1, 8, 8), (8, 8, 5), (5, 8, 1)]]
    flat, shapes = batch.pack_cores(cores)
    assert flat.numel() == 64 + 320 + 40 and shapes.tolist() == [[1, 8, 8], [8, 8, 5], [5, 8, 1]]
    for a, b in zip(cores, batch.unpack_cores(flat, shapes)):
        assert torch.equal(a, b)


def _fake_cores(volume_id):
    g = torch.Generator().manual_seed(volume_id)
    bond = 2 + volume_id % 3  # ragged bonds across volumes
    return [torch.rand((1, 4, bond), generator=g), torch.rand((bond, 4, bond), generator=g),
            torch.rand((bond, 4, 1), generator=g)]


def _worker(rank, world, port, n_volumes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = batch.shard_indices(n_volumes, rank, world)
        local = [_fake_cores(i) for i in mine]
        gathered = batch.all_gather_cores(local)
        ok = len(gathered) == n_volumes
        for i, cores in enumerate(gathered):
            ok = ok and all(torch.equal(a, b) for a, b in zip(cores, _fake_cores(i)))
        vols = [torch.full((3, 2), float(i)) for i in mine]
        allv = batch.all_gather_volumes(vols, n_volumes)
        ok = ok and [float(v[0, 0]) for v in allv] == [float(i) for i in range(n_volumes)]
        # weak-scaling bookkeeping used by bench.py: MAX over ranks of the elapsed time
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_volumes", [5, 8])
def test_all_gather_world_size_2_gloo(n_volumes):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_volumes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
