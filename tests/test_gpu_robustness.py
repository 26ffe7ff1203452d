"""GPU tests of the ways out of the latency-bound kernels (no reference counterpart: the reference is single
threaded NumPy): the resident tridiagonalisation that gives up (status 2) and is redone on the per-column launches,
and turn-taking between streams that share a hardware queue.

The abort itself (a team waiting 3 s for a workgroup that never comes) is not provoked on the GPU: the host-side
hook ``ndmps_debug_inject_team_abort`` replaces the next resident launches by what an aborted one leaves behind
(status 2 in every descriptor, reduction not done).
"""
import ctypes as C
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from imgcompressionmps_amd import NDMPS, _lib  # noqa: E402
from imgcompressionmps_amd.core import batch as batch_mod  # noqa: E402
from oracle.metrics import synthetic_mri  # noqa: E402
from oracle.ndmps_oracle import OracleNDMPS  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device; the product has no CPU path")
    _lib.load()


@pytest.fixture()
def lib():
    lib = _lib.load()
    yield lib
    lib.ndmps_debug_inject_team_abort(0)
    lib.ndmps_syevd_topk_set_team(1)


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((2 * n, n))
    return a.T @ a


@pytest.mark.parametrize("n,batch", [(256, 3), (900, 2), (1600, 1)])
def test_solver_recovers_from_an_aborted_resident_launch(lib, n, batch):
    """values -> recover -> vectors: with status 2 injected, recover redoes phase 1 on the column launches, says so,
    and the eigenpairs are LAPACK's.  Orders up to 2048 take the resident launch whole (4 and 8 rows per thread above
    512) when their teams fit the chip together."""
    k = 32
    g = np.stack([_spd(n, 10 + b) for b in range(batch)])
    dg = torch.from_numpy(g).to(DEV)
    v = torch.empty_like(dg)
    w = torch.empty((batch, n), dtype=torch.float64, device=DEV)
    nbytes = int(lib.ndmps_syevd_topk_workspace_bytes(n, batch, k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    sizes = _lib.i64_array([n] * batch)
    before = int(lib.ndmps_syevd_topk_team_fallbacks())
    for inject, expect in ((1, 1), (0, 0)):
        lib.ndmps_debug_inject_team_abort(inject)
        _lib.check(lib.ndmps_syevd_topk_values_f64(batch, dg.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n,
                                                   k, ws.data_ptr(), nbytes, _lib.stream_ptr()))
        rec = C.c_int(-1)
        _lib.check(lib.ndmps_syevd_topk_recover_f64(batch, sizes, k, ws.data_ptr(), nbytes, C.byref(rec), _lib.stream_ptr()))
        assert rec.value == expect
        status = (C.c_int * batch)()
        _lib.check(lib.ndmps_syevd_topk_vectors_f64(batch, sizes, _lib.i64_array([k] * batch), k, ws.data_ptr(), nbytes,
                                                    status, _lib.stream_ptr()))
        assert list(status) == [0] * batch
        wv, vv = w.cpu().numpy(), v.cpu().numpy()
        for b in range(batch):
            ref = np.linalg.eigvalsh(g[b])[::-1]
            assert np.allclose(wv[b, :k], ref[:k], rtol=0, atol=1e-12 * ref[0])
            vk = vv[b][:, :k]
            assert np.abs(vk.T @ vk - np.eye(k)).max() < 1e-12
            assert np.abs(g[b] @ vk - vk * wv[b, :k]).max() < 1e-11 * ref[0]
    assert int(lib.ndmps_syevd_topk_team_fallbacks()) == before + 1


def test_solver_recovers_when_the_resident_end_of_a_panel_reduction_aborts(lib, monkeypatch):
    """Even uniform orders above 512 hand the last 512 columns of the panel-blocked reduction to the resident kernel
    (csrc/eig_tridiag.hip, `hybrid`).  With that launch replaced by an aborted one, recover redoes phase 1 on the
    panel launches alone; with NDMPS_TRD_NO_HYBRID the panel launches go to the end by themselves.  All three give
    LAPACK's eigenpairs, the first two agree to rounding."""
    n, k, batch = 1280, 96, 2
    monkeypatch.setenv("NDMPS_TRD_PANEL_MIN", "513")
    rng = np.random.default_rng(12)
    g = np.stack([(lambda a: a.T @ a)(rng.standard_normal((n + 64, n)) * np.logspace(0, -5, n)[None, :]) for _ in range(batch)])
    dg = torch.from_numpy(g).to(DEV)
    v = torch.empty_like(dg)
    w = torch.empty((batch, n), dtype=torch.float64, device=DEV)
    nbytes = int(lib.ndmps_syevd_topk_workspace_bytes(n, batch, k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    sizes = _lib.i64_array([n] * batch)
    before = int(lib.ndmps_syevd_topk_team_fallbacks())
    got = {}
    for label, inject, expect, env in (("hybrid", 0, 0, None), ("aborted", 1, 1, None), ("panel", 1, 0, "NDMPS_TRD_NO_HYBRID")):
        if env:
            monkeypatch.setenv(env, "1")
        lib.ndmps_debug_inject_team_abort(inject)
        _lib.check(lib.ndmps_syevd_topk_values_f64(batch, dg.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n,
                                                   k, ws.data_ptr(), nbytes, _lib.stream_ptr()))
        rec = C.c_int(-1)
        _lib.check(lib.ndmps_syevd_topk_recover_f64(batch, sizes, k, ws.data_ptr(), nbytes, C.byref(rec), _lib.stream_ptr()))
        assert rec.value == expect, label
        status = (C.c_int * batch)()
        _lib.check(lib.ndmps_syevd_topk_vectors_f64(batch, sizes, _lib.i64_array([k] * batch), k, ws.data_ptr(), nbytes,
                                                    status, _lib.stream_ptr()))
        assert list(status) == [0] * batch, label
        wv, vv = w.cpu().numpy().copy(), v.cpu().numpy().copy()
        got[label] = (wv, vv)
        for b in range(batch):
            ref = np.linalg.eigvalsh(g[b])[::-1]
            assert np.allclose(wv[b, :k], ref[:k], rtol=0, atol=1e-12 * ref[0]), label
            vk = vv[b][:, :k]
            assert np.abs(vk.T @ vk - np.eye(k)).max() < 1e-12, label
            assert np.abs(g[b] @ vk - vk * wv[b, :k]).max() < 1e-11 * ref[0], label
    lib.ndmps_debug_inject_team_abort(0)  # the panel-only pass has no resident launch to consume the injection
    assert int(lib.ndmps_syevd_topk_team_fallbacks()) == before + 1
    # the recovery IS the panel-only route: same kernels, same order, same bits
    assert np.array_equal(got["aborted"][0][:, :k], got["panel"][0][:, :k])
    assert np.array_equal(got["aborted"][1][:, :, :k], got["panel"][1][:, :, :k])
    assert np.abs(got["hybrid"][0][:, :k] - got["panel"][0][:, :k]).max() <= 1e-13 * got["panel"][0].max()


def test_an_unrecovered_abort_is_reported_not_hidden(lib):
    """Without the recover call the status stays 2 through phase 2 (a later Cholesky breakdown does not mask it)."""
    n, k = 256, 16
    dg = torch.from_numpy(_spd(n, 3)).to(DEV)
    v = torch.empty_like(dg)
    w = torch.empty(n, dtype=torch.float64, device=DEV)
    nbytes = int(lib.ndmps_syevd_topk_workspace_bytes(n, 1, k))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    sizes = _lib.i64_array([n])
    lib.ndmps_debug_inject_team_abort(1)
    _lib.check(lib.ndmps_syevd_topk_values_f64(1, dg.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k,
                                               ws.data_ptr(), nbytes, _lib.stream_ptr()))
    status = (C.c_int * 1)()
    _lib.check(lib.ndmps_syevd_topk_vectors_f64(1, sizes, _lib.i64_array([k]), k, ws.data_ptr(), nbytes, status,
                                                _lib.stream_ptr()))
    assert status[0] == 2


@pytest.mark.parametrize("dtype", [None, torch.float64], ids=["fused_f32", "f64_storage"])
def test_sweep_redoes_itself_on_the_column_launches(lib, dtype):
    """The bond-capped sweep decides ranks on the device and learns of an abort only at its end: the fused fp32 sweep
    (input intact) repeats itself inside the library, the others through NDMPS.from_tensors.  Either way the caller
    gets the result of the column launches and the event is counted."""
    xs = [synthetic_mri((64, 64, 64), seed=2025 + j) for j in range(2)]
    chi = 32  # eigenproblems of order 256 at the middle sites: the resident launch is on their path
    clean = NDMPS.from_tensors(xs, max_bond=chi, device=DEV, dtype=dtype)
    before = int(lib.ndmps_syevd_topk_team_fallbacks())
    lib.ndmps_debug_inject_team_abort(1)
    again = NDMPS.from_tensors(xs, max_bond=chi, device=DEV, dtype=dtype)
    assert int(lib.ndmps_syevd_topk_team_fallbacks()) == before + 1
    assert lib.ndmps_syevd_topk_set_team(1) == 1  # the switch was put back
    for a, b, x in zip(clean, again, xs):
        assert a.bond_sizes() == b.bond_sizes()
        ra, rb = a.to_tensor(), b.to_tensor()
        assert np.linalg.norm(ra - rb) <= 2e-6 * np.linalg.norm(ra)
        ref = OracleNDMPS.from_tensor(x, max_bond=chi)
        assert b.bond_sizes() == ref.bond_sizes()
        assert np.linalg.norm(rb - ref.to_tensor()) <= 2e-5 * np.linalg.norm(ref.to_tensor())


def test_more_groups_than_hardware_queues_take_turns_without_stalling(lib):
    """Streams beyond the runtime's hardware queues (4 by default) share one: a turn-taking spinner that landed between
    another stream's acquire and release in a shared queue would wait for a release queued behind itself until the
    3 s bound.  Turns are submitted atomically (util.hip), so no step may take anything like that long."""
    raw = (C.c_void_p * 8)()
    found = C.c_int(0)
    _lib.check(lib.ndmps_streams_create(8, raw, C.byref(found)))
    _lib.check(lib.ndmps_streams_destroy(8, raw))
    groups = 8
    assert found.value < groups, "every stream got a queue of its own: nothing is shared in this test"
    # 128^3 at chi = 32: order-256 eigenproblems (resident launch + its turn) and, with 4 volumes per group, Gram
    # launches big enough to take the Gram turn
    xs = [torch.from_numpy(synthetic_mri((128, 128, 128), seed=2025 + j)).to(DEV) for j in range(4)]
    xs = xs * groups
    batch_mod.encode_decode_concurrent(xs, groups=groups, max_bond=32)  # warm-up: streams, plans, allocator
    torch.cuda.synchronize()
    worst = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        objs, recs = batch_mod.encode_decode_concurrent(xs, groups=groups, max_bond=32)
        torch.cuda.synchronize()
        worst = max(worst, time.perf_counter() - t0)
    assert worst < 1.0, f"a step of 32 volumes of 128^3 took {worst:.2f} s: a turn waited for its time-out"
    ref = recs[0]
    for j in range(0, len(recs), 4):  # every group encoded the same four volumes
        assert float((recs[j] - ref).norm() / ref.norm()) < 1e-6
    assert objs[0].bond_sizes() == [8, 32, 32, 32, 32, 8]


@pytest.mark.parametrize("bw", ["2", "4"])
def test_two_stage_reduction_gives_lapacks_eigenpairs(lib, bw, monkeypatch):
    """The optional two-stage tridiagonalisation (csrc/eig_band.inc, NDMPS_TRD_BAND=2|4: dense -> band with one
    exchange per panel, bulge chase, two back-transformations) on graded Gram matrices, mixed orders in one batch,
    against LAPACK.  Not the default path (measured slower than the one-stage kernel at order 512)."""
    monkeypatch.setenv("NDMPS_TRD_BAND", bw)
    rng = np.random.default_rng(11)
    orders = [512, 384, 200, 137]
    n_max, k = max(orders), 24
    g = np.zeros((len(orders), n_max, n_max))
    mats = []
    for b, n in enumerate(orders):
        x = rng.standard_normal((2 * n, n)) * np.logspace(0, -5, n)
        mats.append(x.T @ x)
    flat = torch.zeros((len(orders), n_max * n_max), dtype=torch.float64, device=DEV)
    for b, (n, m) in enumerate(zip(orders, mats)):
        flat[b, : n * n] = torch.from_numpy(m.reshape(-1)).to(DEV)
    v = torch.zeros_like(flat)
    w = torch.zeros((len(orders), n_max), dtype=torch.float64, device=DEV)
    nbytes = int(lib.ndmps_syevd_topk_workspace_bytes(n_max, len(orders), k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    sizes = _lib.i64_array(orders)
    _lib.check(lib.ndmps_syevd_topk_values_f64(len(orders), flat.data_ptr(), n_max * n_max, sizes, v.data_ptr(), n_max * n_max,
                                               w.data_ptr(), n_max, k, ws.data_ptr(), nbytes, _lib.stream_ptr()))
    status = (C.c_int * len(orders))()
    _lib.check(lib.ndmps_syevd_topk_vectors_f64(len(orders), sizes, _lib.i64_array([k] * len(orders)), k, ws.data_ptr(),
                                                nbytes, status, _lib.stream_ptr()))
    assert list(status) == [0] * len(orders)
    wv, vv = w.cpu().numpy(), v.cpu().numpy()
    for b, (n, m) in enumerate(zip(orders, mats)):
        ref = np.linalg.eigvalsh(m)[::-1]
        assert np.allclose(wv[b, :k], ref[:k], rtol=0, atol=2e-14 * ref[0])
        vk = vv[b, : n * n].reshape(n, n)[:, :k]
        assert np.abs(vk.T @ vk - np.eye(k)).max() < 1e-12
        assert np.abs(m @ vk - vk * wv[b, :k]).max() < 1e-12 * ref[0]


def test_cross_lane_sums_of_the_tridiagonalisation_kernels(lib):
    """csrc/lanes.h (DPP moves inside a row of 16 lanes, permlane swaps between rows) against plain sums: every
    lane must hold the sum of its group -- 2 ... 64 adjacent lanes, and the lanes with equal lane % 1 ... 32."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(64) * np.logspace(0, 3, 64)
    d_in = torch.from_numpy(x).to(DEV)
    d_out = torch.zeros((12, 64), dtype=torch.float64, device=DEV)
    _lib.check(lib.ndmps_debug_lane_sums_f64(d_in.data_ptr(), d_out.data_ptr(), _lib.stream_ptr()))
    out = d_out.cpu().numpy()
    scale = np.abs(x).sum()
    lanes = np.arange(64)
    for row, n in enumerate([2, 4, 8, 16, 32, 64]):
        want = x.reshape(-1, n).sum(axis=1)[lanes // n]
        assert np.abs(out[row] - want).max() < 1e-15 * scale, f"{n} adjacent lanes"
    for row, stride in enumerate([1, 2, 4, 8, 16, 32], start=6):
        want = x.reshape(-1, stride).sum(axis=0)[lanes % stride]
        assert np.abs(out[row] - want).max() < 1e-15 * scale, f"lanes with equal lane % {stride}"


def test_half_storage_reduction_gives_lapacks_eigenpairs(lib, monkeypatch):
    """The resident tridiagonalisation on half the matrix (csrc/eig_sym.inc; batches that fill more than half the
    GPU's workgroup slots) against LAPACK: mixed orders in one launch -- full 16 block-columns, a ragged last block,
    an odd number of block-columns (the middle workgroup holds one), orders the tail kernel reduces alone."""
    monkeypatch.setenv("NDMPS_TRD_SYM", "1")
    rng = np.random.default_rng(17)
    orders = [512] * 6 + [500] * 4 + [480, 449, 384, 320, 257, 200, 137, 129, 96, 33]
    n_max, k = max(orders), 24
    mats = []
    for n in orders:
        x = rng.standard_normal((2 * n, n)) * np.logspace(0, -5, n)
        mats.append(x.T @ x)
    flat = torch.zeros((len(orders), n_max * n_max), dtype=torch.float64, device=DEV)
    for b, (n, m) in enumerate(zip(orders, mats)):
        flat[b, : n * n] = torch.from_numpy(m.reshape(-1)).to(DEV)
    v = torch.zeros_like(flat)
    w = torch.zeros((len(orders), n_max), dtype=torch.float64, device=DEV)
    nbytes = int(lib.ndmps_syevd_topk_workspace_bytes(n_max, len(orders), k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    sizes = _lib.i64_array(orders)
    for _ in range(2):  # twice on one workspace: nothing of the first run may leak into the second
        _lib.check(lib.ndmps_syevd_topk_values_f64(len(orders), flat.data_ptr(), n_max * n_max, sizes, v.data_ptr(),
                                                   n_max * n_max, w.data_ptr(), n_max, k, ws.data_ptr(), nbytes, _lib.stream_ptr()))
        status = (C.c_int * len(orders))()
        _lib.check(lib.ndmps_syevd_topk_vectors_f64(len(orders), sizes, _lib.i64_array([min(k, n) for n in orders]), k,
                                                    ws.data_ptr(), nbytes, status, _lib.stream_ptr()))
        assert list(status) == [0] * len(orders)
        wv, vv = w.cpu().numpy(), v.cpu().numpy()
        for b, (n, m) in enumerate(zip(orders, mats)):
            kb = min(k, n)
            ref = np.linalg.eigvalsh(m)[::-1]
            assert np.allclose(wv[b, :kb], ref[:kb], rtol=0, atol=2e-14 * ref[0]), n
            vk = vv[b, : n * n].reshape(n, n)[:, :kb]
            assert np.abs(vk.T @ vk - np.eye(kb)).max() < 1e-12, n
            assert np.abs(m @ vk - vk * wv[b, :kb]).max() < 1e-12 * ref[0], n


def test_two_streams_of_narrow_teams_share_the_gpu_without_giving_up(lib):
    """Eight order-512 matrices run with 8-column blocks: 512 workgroups, every resident slot of the GPU.  Two such
    launches from two streams must not be in flight together (each would hold slots the other's teams wait for, both
    would spin out their 3 s and fall back): a launch takes half a turn only if it fits half the slots.  Both streams
    finish, nobody gives up, and the eigenvalues are LAPACK's."""
    import threading

    n, k, batch = 512, 64, 8
    before = int(lib.ndmps_syevd_topk_team_fallbacks())
    mats = [np.stack([_spd(n, 100 * t + b) for b in range(batch)]) for t in range(2)]
    out, errs = [None, None], []

    def run(t):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                dg = torch.from_numpy(mats[t]).to(DEV)
                v = torch.empty_like(dg)
                w = torch.empty((batch, n), dtype=torch.float64, device=DEV)
                nbytes = int(lib.ndmps_syevd_topk_workspace_bytes(n, batch, k))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
                sizes = _lib.i64_array([n] * batch)
                for _ in range(3):
                    _lib.check(lib.ndmps_syevd_topk_values_f64(batch, dg.data_ptr(), n * n, sizes, v.data_ptr(), n * n,
                                                               w.data_ptr(), n, k, ws.data_ptr(), nbytes, _lib.stream_ptr()))
                torch.cuda.current_stream().synchronize()
                out[t] = w.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    took = time.perf_counter() - t0
    assert not errs, errs
    assert int(lib.ndmps_syevd_topk_team_fallbacks()) == before
    assert took < 2.5, f"two streams of eight order-512 matrices took {took:.2f} s: a team waited for its time-out"
    for t in range(2):
        for b in range(batch):
            ref = np.linalg.eigvalsh(mats[t][b])[::-1]
            assert np.abs(out[t][b, :k] - ref[:k]).max() <= 2e-13 * ref[0]


def test_the_solver_switch_is_per_host_thread(lib):
    """ndmps_syevd_topk_set_team(0) -- what a retry after an aborted resident launch sets around its second attempt --
    acts on the calling host thread only (a thread_local in the library): while one thread holds the switch off, a
    lockstep group encoded on another thread still takes the resident launches and gets, bit for bit, what it gets
    with nobody else around; the thread that holds the switch gets the column launches' result, bit for bit."""
    import threading

    xs = [synthetic_mri((64, 64, 64), seed=2025 + j) for j in range(2)]
    chi = 32  # eigenproblems of order 256 at the middle sites: the resident launch is on their path

    def encode():
        return [o.to_tensor() for o in NDMPS.from_tensors(xs, max_bond=chi, device=DEV, dtype=torch.float64)]

    clean = encode()
    assert lib.ndmps_syevd_topk_set_team(0) == 1
    columns = encode()
    assert lib.ndmps_syevd_topk_set_team(1) == 0
    assert any(not np.array_equal(a, b) for a, b in zip(clean, columns)), "the two routes agree bit for bit: no test"
    off, done = threading.Event(), threading.Event()
    seen = {}

    def holder():
        seen["was"] = lib.ndmps_syevd_topk_set_team(0)
        off.set()
        done.wait(timeout=120)
        seen["columns"] = encode()
        seen["back"] = lib.ndmps_syevd_topk_set_team(1)

    th = threading.Thread(target=holder)
    th.start()
    assert off.wait(timeout=60)
    beside = encode()  # this thread's switch is untouched
    assert lib.ndmps_syevd_topk_set_team(1) == 1
    done.set()
    th.join(timeout=180)
    assert seen["was"] == 1 and seen["back"] == 0
    for a, b in zip(clean, beside):
        assert np.array_equal(a, b)
    for a, b in zip(columns, seen["columns"]):
        assert np.array_equal(a, b)


def test_a_pending_group_redoes_itself_when_a_resident_launch_gave_up(monkeypatch):
    """from_tensors_begin enqueues a sweep whose solver status is only read in result(): when that status says a resident
    tridiagonalisation gave up (NDMPS_ETEAM from ndmps_tt_sweep_finish -- injected here on the host side, no 3-s spin), the
    volumes are intact and the group is encoded again through the one-call form; the caller gets the same objects and
    reconstructions as from from_tensors, and a later group is not affected."""
    from oracle.metrics import synthetic_mri

    lib = _lib.load()
    vols = [torch.from_numpy(synthetic_mri((64, 64, 64), seed=300 + i)).to(DEV) for i in range(4)]
    want_objs, want_recs = NDMPS.from_tensors(vols, max_bond=16, reconstruct=True)
    real = lib.ndmps_tt_sweep_finish
    calls = []

    def gave_up(*args):
        calls.append(1)
        return _lib.ETEAM if len(calls) == 1 else real(*args)

    monkeypatch.setattr(lib, "ndmps_tt_sweep_finish", gave_up)
    first = NDMPS.from_tensors_begin(vols, max_bond=16, reconstruct=True)
    second = NDMPS.from_tensors_begin(vols, max_bond=16, reconstruct=True)
    assert first.asynchronous and second.asynchronous
    for pend in (first, second):
        objs, recs = pend.result()
        for a, b, ra, rb in zip(want_objs, objs, want_recs, recs):
            assert a.bond_sizes() == b.bond_sizes()
            assert all(torch.equal(x, y) for x, y in zip(a.mps.cores, b.mps.cores))
            assert torch.equal(ra, rb)
    assert len(calls) == 2  # the redone group went through the one-call sweep, not through finish again
