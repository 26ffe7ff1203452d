"""One tensor sharded by rows over ranks (core/sharded.py): the bond-capped sweep with an all-reduce of the Gram
matrices, against the single-GPU sweep of the same tensor.  World size 1 in-process; world size 2 as two processes
that share cuda:0 and talk through gloo (RCCL needs one device per rank: the 8-GPU run is the driver's)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

pytestmark = pytest.mark.gpu

from imgcompressionmps_amd import NDMPS, _lib  # noqa: E402
from imgcompressionmps_amd.core import sharded  # noqa: E402
from imgcompressionmps_amd.core.ndmps import _plan_for  # noqa: E402
from oracle import mps as omps  # noqa: E402
from oracle.metrics import synthetic_mri  # noqa: E402

DEV = "cuda:0"


def _site_order(x):
    """The site-order tensor of a C-order volume (the library's own reshape stage) and its site dims."""
    lib = _lib.load()
    plan = _plan_for(tuple(x.shape), 0)
    src = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV)
    dense = torch.empty(plan.numel, dtype=torch.float32, device=DEV)
    _lib.check(lib.ndmps_encode_permute(plan.handle, src.data_ptr(), dense.data_ptr(), 4, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return dense, [int(q) for q in plan.qubit_size]


def _check_against_single_gpu(x, chi, rank, world, group=None):
    dense, dims = _site_order(x)
    n_local = dense.numel() // world
    mine = dense[rank * n_local:(rank + 1) * n_local].clone()
    mps = sharded.from_dense_sharded(mine, dims, max_bond=chi, group=group)
    ref = NDMPS.from_tensor(x, max_bond=chi)
    assert mps.bond_sizes() == ref.bond_sizes()
    ref_dense = ref.mps.to_dense()
    rec = sharded.local_dense(mps, rank, world)
    want = ref_dense[rank * n_local:(rank + 1) * n_local]
    rel = float((rec - want).norm() / want.norm())
    assert rel <= 2e-5, rel
    # the replicated MPS stands for the whole tensor
    full = mps.to_dense()
    assert float((full - ref_dense).norm() / ref_dense.norm()) <= 2e-5
    err = float((full - dense).norm() / dense.norm())
    err_ref = float((ref_dense - dense).norm() / dense.norm())
    assert abs(err - err_ref) <= 1e-5 * max(err_ref, 1e-3)
    # and against the ORACLE's sweep of the same site-order tensor, not only against the single-GPU HIP sweep
    cores, _ = omps.mps_from_dense(dense.double().cpu().numpy(), dims, max_bond=chi)
    assert mps.bond_sizes() == [int(c.shape[2]) for c in cores[:-1]]
    want64 = omps.mps_to_dense(cores).reshape(-1)
    got64 = full.double().cpu().numpy().reshape(-1)
    assert np.linalg.norm(got64 - want64) <= 2e-5 * np.linalg.norm(want64)
    return True


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device; the product has no CPU path")


@pytest.mark.parametrize("shape,chi", [((64, 64, 64), 16), ((128, 128, 128), 32), ((32, 32, 16, 24), 12)])
def test_sharded_sweep_world_size_1_equals_the_ordinary_sweep(shape, chi):
    assert _check_against_single_gpu(synthetic_mri(shape, seed=5), chi, 0, 1)


@pytest.mark.parametrize("shape,chi,why", [
    ((512, 680), 16, "dims [34, 20, 8, 8, 8]: the first unfolding is already too short to shard (break at i = L - 1)"),
    ((8, 512, 680), 64, "dims [160, 128, 136]: sharded down to i = 0, the replicated rest is a single-site tensor"),
], ids=["breaks_at_once", "runs_to_site_0"])
def test_sharded_sweep_edge_exits(shape, chi, why):
    """The two ends of the loop in from_dense_sharded: no sharded site at all, and every site but the first sharded
    (the gathered head then has one site and no bond of its own)."""
    x = np.random.default_rng(2025).random(shape).astype(np.float32)
    dense, dims = _site_order(x)
    mps = sharded.from_dense_sharded(dense.clone(), dims, max_bond=chi)
    cores, _ = omps.mps_from_dense(dense.double().cpu().numpy(), dims, max_bond=chi)
    assert mps.bond_sizes() == [int(c.shape[2]) for c in cores[:-1]], why
    want = omps.mps_to_dense(cores).reshape(-1)
    got = mps.to_dense().double().cpu().numpy().reshape(-1)
    assert np.linalg.norm(got - want) <= 2e-5 * np.linalg.norm(want), why


def test_sharded_sweep_rejects_bad_arguments():
    dense, dims = _site_order(synthetic_mri((32, 32, 32), seed=1))
    with pytest.raises(ValueError):
        sharded.from_dense_sharded(dense, dims, max_bond=None)
    with pytest.raises(ValueError):
        sharded.from_dense_sharded(dense[:100], dims, max_bond=8)
    with pytest.raises(ValueError):
        sharded.from_dense_sharded(dense.double(), dims, max_bond=8)


def _worker(rank, world, port, shape, chi, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, bool(_check_against_single_gpu(synthetic_mri(shape, seed=5), chi, rank, world))))
    except Exception as exc:  # the assertion text travels back to the parent
        q.put((rank, f"{type(exc).__name__}: {exc}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,chi", [((64, 64, 64), 16), ((128, 128, 128), 32)])
def test_sharded_sweep_two_ranks_all_reduce_the_gram_matrices(shape, chi):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape, chi, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)], results


def _worker_with_aborts(rank, world, port, shape, chi, inject_on, count, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = _lib.load()
        before = int(lib.ndmps_syevd_topk_team_fallbacks())
        if rank == inject_on:
            lib.ndmps_debug_inject_team_abort(count)
        ok = bool(_check_against_single_gpu(synthetic_mri(shape, seed=5), chi, rank, world))
        lib.ndmps_debug_inject_team_abort(0)
        q.put((rank, ok, int(lib.ndmps_syevd_topk_team_fallbacks()) - before))
    except Exception as exc:  # the assertion text travels back to the parent
        q.put((rank, f"{type(exc).__name__}: {exc}", -1))
    finally:
        dist.destroy_process_group()


def test_sharded_sweep_a_resident_launch_that_gives_up_on_one_rank_sends_every_rank_to_the_column_launches():
    """128^3 at chi = 32: the sharded sites 4 and 3 (order 256) and the replicated rest take the resident
    tridiagonalisation (the replicated rest, an 8 x 8 x 256 tensor, has none).  On rank 1 the next two resident launches
    are replaced by what an aborted one leaves behind (status 2).  Every decision is collective -- rank 0, whose
    launches were fine, redoes the same sites on the column launches -- so nobody hangs in an unmatched collective,
    both ranks count the same two fall-backs, and the result is the single-GPU sweep's and the oracle's."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_with_aborts, args=(r, 2, port, (128, 128, 128), 32, 1, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, 2), (1, True, 2)], results


def _worker_tail_abort(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = _lib.load()
        # site dims chosen so that the REPLICATED rest holds a resident launch: site 3 (order 32) is sharded, site 2
        # (order 8 * 32 = 256, 400 rows in all, 200 per rank) is not, and its 400 x 256 unfolding is tall
        dims, chi = [2, 200, 8, 32], 32
        rng = np.random.default_rng(7)
        dense = np.einsum("ia,ajb,bkc,cl->ijkl", rng.standard_normal((2, 6)), rng.standard_normal((6, 200, 9)),
                          rng.standard_normal((9, 8, 40)), rng.standard_normal((40, 32)))
        dense = (dense + 1e-3 * rng.standard_normal(dense.shape)).astype(np.float32)
        flat = torch.from_numpy(dense.reshape(-1)).to(DEV)
        n_local = flat.numel() // world
        before = int(lib.ndmps_syevd_topk_team_fallbacks())
        if rank == 1:
            lib.ndmps_debug_inject_team_abort(1)
        mps = sharded.from_dense_sharded(flat[rank * n_local:(rank + 1) * n_local].clone(), dims, max_bond=chi)
        lib.ndmps_debug_inject_team_abort(0)
        counted = int(lib.ndmps_syevd_topk_team_fallbacks()) - before
        cores, _ = omps.mps_from_dense(dense.astype(np.float64), dims, max_bond=chi)
        want = omps.mps_to_dense(cores).reshape(-1)
        got = mps.to_dense().double().cpu().numpy().reshape(-1)
        ok = mps.bond_sizes() == [int(c.shape[2]) for c in cores[:-1]] and \
            float(np.linalg.norm(got - want)) <= 2e-5 * float(np.linalg.norm(want))
        q.put((rank, bool(ok), counted))
    except Exception as exc:  # the assertion text travels back to the parent
        q.put((rank, f"{type(exc).__name__}: {exc}", -1))
    finally:
        dist.destroy_process_group()


def test_sharded_sweep_an_abort_in_the_replicated_rest_is_repeated_by_every_rank():
    """The replicated rest of the sweep overwrites its gathered input; when its resident launch gives up on ONE rank,
    EVERY rank gathers again (a collective) and repeats it on the column launches -- decided by an all-reduce every
    rank joins, not in an except branch only the aborting rank would take."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_tail_abort, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, 1), (1, True, 1)], results


@pytest.mark.parametrize("shape", [(64, 64, 64), (48, 40, 36), (32, 32, 16, 24), (512, 680), (128, 128, 128)])
def test_top_level_blocks_are_the_rows_of_the_site_order_tensor(shape):
    """Bit-exact: encoding a rank's top-level blocks on their own gives its rows of the full site-order tensor."""
    x = synthetic_mri(shape, seed=9) if len(shape) > 2 else np.random.default_rng(9).random(shape).astype(np.float32)
    dense, dims = _site_order(x)
    d0 = dims[0]
    for world in [w for w in (1, 2, d0) if d0 % w == 0]:
        step = d0 // world
        for rank in range(world):
            blocks = [x[sharded.top_block_slices(shape, d)] for d in range(rank * step, (rank + 1) * step)]
            mine = sharded.shard_from_blocks(blocks, shape)
            n_local = dense.numel() // world
            assert torch.equal(mine, dense[rank * n_local:(rank + 1) * n_local]), (world, rank)
    with pytest.raises(ValueError):
        sharded.top_block_slices(shape, d0)


def test_from_volume_sharded_equals_from_tensor():
    shape, chi = (64, 64, 64), 16
    x = synthetic_mri(shape, seed=2)
    blocks = [x[sharded.top_block_slices(shape, d)] for d in range(8)]
    mps = sharded.from_volume_sharded(blocks, shape, max_bond=chi)
    ref = NDMPS.from_tensor(x, max_bond=chi)
    assert mps.bond_sizes() == ref.bond_sizes()
    a, b = mps.to_dense(), ref.mps.to_dense()
    assert float((a - b).norm() / b.norm()) <= 2e-5


def test_bench_config_5_on_two_ranks_sharing_the_gpu():
    """The driver's multi-GPU line for BASELINE config 5 (strong scaling: ONE tensor, its top-level blocks dealt to the
    ranks, the Gram matrices all-reduced), rehearsed the way this box allows: two ranks over gloo, both on cuda:0
    (bench.py --share-gpu).  Rank 0 prints one JSON line; its reconstruction check against the single-GPU sweep is
    part of the run (bench.py raises if the sharded result is off)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--config", "5", "--gpus", "2", "--backend", "gloo",
           "--share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs", "--skip-single"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["workload"].startswith("ONE 128x128x64x256")


@pytest.mark.parametrize("flags", [[], ["--no-pipeline"], ["--total-volumes", "8"]], ids=["stream", "one-call", "strong"])
def test_bench_cube_config_on_two_ranks_sharing_the_gpu(flags):
    """The driver's multi-GPU command for the headline path (independent volumes sharded over the ranks, no data-path
    collective), rehearsed the way this box allows: two ranks over gloo, both on cuda:0, small volumes.  Batches issued as
    a stream (the default: encode_decode_begin on three lanes, objects built two batches later), one call per batch
    (--no-pipeline) and strong scaling (8 volumes in total, 4 per rank); rank 0 prints one JSON line with the whole job's
    throughput."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
           "--size", "64", "--chi", "16", "--batch", "6", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-configs",
           "--skip-single"] + flags
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    strong = "--total-volumes" in flags
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["steps"] == 4
    assert line["scaling"] == ("strong" if strong else "weak")
    assert line["config"]["volumes_per_step"] == (8 if strong else 12)
    assert line["team_fallbacks"] == 0
    assert ("none" in line["config"]["host_pipeline"]) == ("--no-pipeline" in flags)
    # throughput counts the job's volumes once: volumes x voxels / time
    assert line["value"] == pytest.approx(line["config"]["volumes_per_step"] * 64 ** 3 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
