"""One tensor too large for one GPU: the bond-capped sweep with the ROWS of every unfolding sharded over ranks
(BASELINE config 5 "1 and 8 GPUs").

The right-to-left sweep (``MatrixProductState.from_dense``, core/ndmps.py:74 of the reference) needs, per site, the
Gram matrix of the unfolding ``A_i`` (rows = the sites to the left, ``n_i = d_i chi_{i+1}`` columns).  Rows split
over ranks cleanly: rank r holds a contiguous block of the rows of the SITE-ORDER tensor (for 2^k cubes and R = d_0
ranks: one octant of the volume, in its own site order), and

    G_i = sum_r A_i[r]^T A_i[r]                      -- the one real exchange step: an all-reduce of n_i x n_i fp64

after which every rank solves the same eigenproblem (same bits in, same bits out: the solver is deterministic),
keeps the same core and projects its own rows.  When the rows to the left are no longer more than the columns, the
carried matrix (at most a few MB) is all-gathered and every rank finishes the remaining sites with the ordinary
single-GPU sweep: the carried matrix of site i is a tensor over the sites 0..i-1 with ``d_i chi_{i+1}`` as its last
"site", whose core is core i.  The result is a replicated ``DeviceMPS``; ``local_dense`` reconstructs a rank's own
rows only.

Every product, Gram matrix and eigen-decomposition is a kernel of libndmps_hip.so (ndmps_gram_f32,
ndmps_syevd_topk_*, ndmps_sgemm, ndmps_tt_sweep_f32, ndmps_chain_contract_f32); torch.distributed carries the two
collectives (backend "nccl" = RCCL over xGMI on a node; "gloo" in the tests).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from .mps import DeviceMPS

_CUTOFF_FLOOR = 1e-6  # csrc/tt.hip kCutoffFloor: fp32 data


def _all_max(flag: int, device, world, group) -> int:
    """The largest of the ranks' flags (a collective every rank joins; world size 1: the flag itself)."""
    import torch
    import torch.distributed as dist

    if world <= 1:
        return int(flag)
    t = torch.tensor([int(flag)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def from_dense_sharded(local_dense, dims, max_bond, cutoff: float = 1e-10, group=None) -> DeviceMPS:
    """``local_dense``: this rank's rows of the site-order tensor, a device fp32 tensor of
    ``prod(dims) / world`` elements (rank r holds rows ``[r, r + 1) * prod(dims) / world`` of the flattened tensor;
    ``world`` must divide ``dims[0]``).  ``dims``: the site dimensions of the FULL tensor.  Returns the MPS of the full
    tensor, replicated on every rank.  ``local_dense`` is left untouched.

    Per sharded site: ONE collective on the data path (the all-reduce of the Gram matrix) and no host read -- the rank
    is decided on the device from the eigenvalues, as in the single-GPU sweep (cores and carried matrices keep their
    cap shapes, zero beyond the rank; the ranks come back in one copy behind the loop).  Only sites whose
    eigenproblem may take the resident tridiagonalisation (orders 129 .. 512) add a one-word all-reduce and one host
    read: a launch that gave up on ANY rank sends EVERY rank to the column launches for that site (the two routes
    differ in the last bits and the ranks must keep identical cores).  Every decision that leads to a collective or
    to an exception is itself taken collectively."""
    import torch
    import torch.distributed as dist

    lib = _lib.load()
    _lib.require_device()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dims = [int(d) for d in dims]
    L = len(dims)
    if not max_bond or int(max_bond) > int(lib.ndmps_syevd_topk_max_k()):
        raise ValueError(f"the sharded sweep needs 1 <= max_bond <= {int(lib.ndmps_syevd_topk_max_k())}")
    chi = int(max_bond)
    numel = int(np.prod(dims, dtype=np.int64))
    if dims[0] % world != 0:
        raise ValueError(f"the number of ranks ({world}) must divide the first site dimension ({dims[0]})")
    if local_dense.numel() * world != numel or local_dense.dtype != torch.float32 or not local_dense.is_cuda:
        raise ValueError("local_dense must be this rank's prod(dims) / world fp32 elements on the device")
    device = local_dense.device
    stream = _lib.stream_ptr()
    max_n = int(lib.ndmps_syevd_topk_max_n())
    # ---- the sharded sites, known up front: the shapes follow the caps, not the ranks
    sites, chi_r, i = [], 1, L - 1
    while i >= 1:
        n = dims[i] * chi_r
        rows_local = int(np.prod(dims[:i], dtype=np.int64)) // world
        # sharded while the unfolding is tall on every rank and the eigenproblem fits the direct solver
        if rows_local < max(n, 256) or n > max_n:
            break
        sites.append((i, n, rows_local, chi_r, min(chi, n)))
        chi_r, i = min(chi, n), i - 1
    i_head = i
    cores = [None] * L
    cut = max(float(cutoff), _CUTOFF_FLOOR)
    with torch.cuda.device(device):
        # ---- one arena for every site
        n_big = max([s[1] for s in sites], default=1)
        g = torch.empty(n_big * n_big, dtype=torch.float64, device=device)
        v = torch.empty(n_big * n_big, dtype=torch.float64, device=device)
        w = torch.empty(n_big, dtype=torch.float64, device=device)
        gbytes = max([int(lib.ndmps_gram_workspace_bytes(s[2], s[1])) for s in sites], default=0)
        ebytes = max([int(lib.ndmps_syevd_topk_workspace_bytes(s[1], 1, s[4])) for s in sites], default=0)
        gws = torch.empty(max(gbytes, 1), dtype=torch.uint8, device=device)
        ews = torch.empty(max(ebytes, 1), dtype=torch.uint8, device=device)
        pong = [torch.empty(max([s[2] * s[4] for s in sites[k::2]], default=1), dtype=torch.float32, device=device)
                for k in range(2)]
        ranks_dev = torch.zeros(max(len(sites), 1), dtype=torch.int32, device=device)
        status_dev = torch.zeros(max(len(sites), 1), dtype=torch.int32, device=device)
        carried = local_dense.reshape(-1)          # (rows_local, n_i) row-major, rows_local = rows_global / world
        for t, (i, n, rows_local, chi_r, k_cap) in enumerate(sites):
            a = carried
            gm = g[: n * n]
            _lib.check(lib.ndmps_gram_f32(a.data_ptr(), rows_local, n, n, gm.data_ptr(), gws.data_ptr(), gbytes, stream))
            if world > 1:
                dist.all_reduce(gm, group=group)  # the exchange step: Gram matrices of the row blocks add up
            sizes = _lib.i64_array([n])

            def solve():
                _lib.check(lib.ndmps_syevd_topk_values_f64(1, gm.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(),
                                                           n, k_cap, ews.data_ptr(), ebytes, stream))
                _lib.check(lib.ndmps_syevd_topk_vectors_auto_f64(1, sizes, k_cap, cut, ranks_dev[t:].data_ptr(), None, 0,
                                                                 status_dev[t:].data_ptr(), ews.data_ptr(), ebytes, stream))

            solve()
            if 128 < n <= 512:  # the resident tridiagonalisation may have run: did it give up anywhere?
                gave_up = int(status_dev[t].item()) == 2
                if _all_max(gave_up, device, world, group):
                    _lib.check(lib.ndmps_syevd_topk_note_team_fallback())
                    was = lib.ndmps_syevd_topk_set_team(0)
                    try:
                        solve()
                    finally:
                        lib.ndmps_syevd_topk_set_team(was)
            basis = v[: n * n].view(n, n)[:, :k_cap].to(torch.float32).contiguous()  # (n, k_cap), zero beyond the rank
            cores[i] = basis.t().contiguous().view(k_cap, dims[i], chi_r)             # rows of V^T, like the single-GPU sweep
            nxt = pong[t & 1][: rows_local * k_cap]
            _lib.check(lib.ndmps_sgemm(0, 0, rows_local, k_cap, n, a.data_ptr(), n, basis.data_ptr(), k_cap, nxt.data_ptr(),
                                       k_cap, stream))
            carried = nxt
        # ---- ranks and status of every sharded site in one copy; cores and the carried matrix cut to the ranks
        chi_r = 1
        if sites:
            ranks = [int(r) for r in ranks_dev.cpu().numpy()[: len(sites)]]
            bad = [(sites[t][0], int(st)) for t, st in enumerate(status_dev.cpu().numpy()[: len(sites)]) if st != 0]
            if bad:
                raise _lib.NdmpsHipError(f"the eigen-solver reported (site, status) {bad}")
            right = 1
            for t, (i, n, rows_local, _c, k_cap) in enumerate(sites):
                cores[i] = cores[i][: ranks[t], :, :right].contiguous()
                right = ranks[t]
            last = sites[-1]
            carried = carried.view(last[2], last[4])[:, : ranks[-1]].contiguous().view(-1)
            chi_r = ranks[-1]
        i = i_head
        # ---- the rest on every rank: gather the carried matrix (rows of sites 0..i, chi_r columns)
        # what is left runs replicated: the gathered carried matrix plus the ordinary sweep's workspace must fit
        head = dims[:i] + [dims[i] * chi_r]     # the carried matrix as a tensor whose last "site" is (d_i, chi_{i+1})
        wsb = C.c_int64()
        _lib.check(lib.ndmps_tt_layout(len(head), _lib.i64_array(head), chi, (C.c_int64 * (len(head) + 1))(),
                                       (C.c_int64 * (len(head) + 1))(), (C.c_int64 * (len(head) + 1))(), C.byref(wsb)))
        need = 2 * carried.numel() * world * 4 + int(wsb.value)
        free, _total = torch.cuda.mem_get_info(device)
        if _all_max(need > free, device, world, group):  # a rank that cannot hold it stops every rank, before the gather
            raise MemoryError(
                f"the replicated part of the sharded sweep (sites 0..{i}: {carried.numel() * world} carried elements "
                f"gathered on every rank + {int(wsb.value)} bytes of sweep workspace = {need} bytes) does not fit the "
                f"free memory of every rank ({free} bytes free on this rank's {device}); use more ranks only if the "
                f"left sites stay sharded (the loop stops sharding once a rank holds fewer than max(n_i, 256) rows)")

        def gathered():
            if world > 1:
                parts = [torch.empty_like(carried) for _ in range(world)]
                dist.all_gather(parts, carried.contiguous(), group=group)
                return torch.cat(parts)
            return carried.clone()  # the ordinary sweep works in place

        full = gathered()
        Lh = len(head)
        cdims = _lib.i64_array(head)
        max_bonds = (C.c_int64 * (Lh + 1))()
        core_off = (C.c_int64 * (Lh + 1))()
        spec_off = (C.c_int64 * (Lh + 1))()
        _lib.check(lib.ndmps_tt_layout(Lh, cdims, chi, max_bonds, core_off, spec_off, C.byref(wsb)))
        arena = torch.zeros(int(core_off[Lh]) + 64, dtype=torch.float32, device=device)
        ws = torch.empty(int(wsb.value), dtype=torch.uint8, device=device)
        bonds = (C.c_int64 * (Lh + 1))()
        spectra = (C.c_double * max(int(spec_off[Lh]), 1))()

        def sweep():
            _lib.check(lib.ndmps_tt_sweep_f32(full.data_ptr(), Lh, cdims, float(cutoff), chi, arena.data_ptr(), core_off,
                                              bonds, spectra, spec_off, ws.data_ptr(), int(wsb.value), stream))

        gave_up = 0
        try:
            sweep()
        except _lib.NdmpsTeamAbort:
            gave_up = 1
        # the sweep overwrote its input, and the two routes differ in the last bits: if ANY rank's resident launch gave
        # up, EVERY rank gathers again and repeats the sweep on the column launches
        if _all_max(gave_up, device, world, group):
            _lib.check(lib.ndmps_syevd_topk_note_team_fallback())
            full = gathered()
            was = lib.ndmps_syevd_topk_set_team(0)
            try:
                sweep()
            finally:
                lib.ndmps_syevd_topk_set_team(was)
        padded = bool(lib.ndmps_tt_sweep_pads_cores(Lh, cdims, chi))
        for j in range(Lh):
            k0, k1 = int(bonds[j]), int(bonds[j + 1])
            if padded:
                c0, c1 = int(max_bonds[j]), int(max_bonds[j + 1])
                blk = arena[int(core_off[j]): int(core_off[j]) + c0 * head[j] * c1].view(c0, head[j], c1)[:k0, :, :k1]
            else:
                blk = arena[int(core_off[j]): int(core_off[j]) + k0 * head[j] * k1].view(k0, head[j], k1)
            blk = blk.contiguous().clone()
            cores[j] = blk.view(k0, dims[i], chi_r) if j == Lh - 1 else blk
    return DeviceMPS(cores)


def local_dense(mps: DeviceMPS, rank: int, world: int):
    """This rank's rows of the site-order tensor the MPS stands for (``prod(dims) / world`` elements): the chain
    contraction with the first core restricted to the rank's block of the first site.  The restricted first core
    is multiplied into its right neighbours until it has at least as many rows as columns, so that what is handed
    to the chain kernel is again an MPS whose bonds respect the unfolding ranks."""
    import torch

    lib = _lib.load()
    dims, L = mps.dims, len(mps.cores)
    if dims[0] % world != 0:
        raise ValueError("world must divide the first site dimension")
    step = dims[0] // world
    device = mps.device
    with torch.cuda.device(device):
        cur = mps.cores[0][0, rank * step:(rank + 1) * step, :].to(torch.float32).contiguous()  # (rows, chi_1)
        rows, j = step, 1
        while j < L - 1 and rows < cur.shape[1]:
            core = mps.cores[j].to(torch.float32).contiguous()
            chi_j, d_j, chi_n = (int(x) for x in core.shape)
            out = torch.empty((rows, d_j * chi_n), dtype=torch.float32, device=device)
            _lib.check(lib.ndmps_sgemm(0, 0, rows, d_j * chi_n, chi_j, cur.data_ptr(), chi_j, core.data_ptr(), d_j * chi_n,
                                       out.data_ptr(), d_j * chi_n, _lib.stream_ptr()))
            rows *= d_j
            cur = out.view(rows, chi_n)
            j += 1
        head = cur.reshape(1, rows, cur.shape[1]).contiguous()
        return DeviceMPS([head] + [c.to(torch.float32) for c in mps.cores[j:]]).to_dense()


# ---------------------------------------------------------------- from blocks of the C-order volume
# Site 0 of the encoding (utils/core.py: hierarchical_block_indexing) numbers the TOP-LEVEL blocks of the volume:
# axis a is cut into f_0[a] pieces, the block (b_0, .., b_{D-1}) has digit ravel_multi_index(b, f_0), and inside a
# block the remaining sites encode the block's own voxels with the factor lists of levels 1 .. L-1.  A rank that
# owns the digits [r, r + 1) * d_0 / world therefore needs exactly those sub-boxes of the volume, and its rows of
# the site-order tensor are the blocks' own site-order tensors, one after the other.
def top_block_slices(shape, digit: int):
    """Slices of the C-order volume that make up top-level block ``digit`` (0 <= digit < site_dims(shape)[0])."""
    from ..utils import core as _core

    fa, _ = _core.get_factorlist(tuple(int(s) for s in shape))
    f0 = [int(f) for f in fa[0]]
    if not 0 <= int(digit) < int(np.prod(f0)):
        raise ValueError(f"top-level block {digit} outside [0, {int(np.prod(f0))})")
    b = np.unravel_index(int(digit), f0)
    sizes = [int(s) // f for s, f in zip(shape, f0)]
    return tuple(slice(int(bi) * sz, (int(bi) + 1) * sz) for bi, sz in zip(b, sizes))


def shard_from_blocks(blocks, shape, device=None):
    """``blocks``: the top-level blocks a rank owns, in digit order (each a C-order sub-box
    ``volume[top_block_slices(shape, digit)]``, NumPy or torch).  Returns the rank's rows of the site-order tensor
    (fp32, device): every block goes through the library's reshape stage with the factor lists of levels 1 .. L-1."""
    import torch

    from ..utils import core as _core

    lib = _lib.load()
    _lib.require_device()
    shape = tuple(int(s) for s in shape)
    fa, _ = _core.get_factorlist(shape)
    if fa.shape[0] < 2:
        raise ValueError("a single-site encoding has no rows to shard")
    sub = np.ascontiguousarray(fa[1:], dtype=np.int64)
    sub_shape = tuple(int(v) for v in np.prod(sub, axis=0))
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    handle = C.c_void_p()
    _lib.check(lib.ndmps_plan_create(C.byref(handle), len(sub_shape), _lib.i64_array(sub_shape), sub.shape[0],
                                     sub.ctypes.data_as(_lib.p_i64)))
    try:
        n_sub = int(np.prod(sub_shape, dtype=np.int64))
        out = torch.empty(len(blocks) * n_sub, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            for t, blk in enumerate(blocks):
                src = blk if isinstance(blk, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(blk, dtype=np.float32))
                if tuple(src.shape) != sub_shape:
                    raise ValueError(f"block {t} has shape {tuple(src.shape)}, a top-level block of {shape} is {sub_shape}")
                src = src.to(device=device, dtype=torch.float32).contiguous()
                _lib.check(lib.ndmps_encode_permute(handle, src.data_ptr(), out[t * n_sub:].data_ptr(), 4, _lib.stream_ptr()))
            torch.cuda.current_stream().synchronize()  # the plan's tables are freed below
    finally:
        lib.ndmps_plan_destroy(handle)
    return out


def from_volume_sharded(blocks, shape, max_bond, cutoff: float = 1e-10, group=None, device=None) -> DeviceMPS:
    """The bond-capped MPS of ONE volume of ``shape`` held as top-level blocks across the ranks of ``group``: this
    rank passes the blocks with digits ``[rank, rank + 1) * d_0 / world`` in order.  Replicated result."""
    from ..utils import core as _core

    dims = [int(q) for q in _core.site_dims(tuple(int(s) for s in shape))]
    return from_dense_sharded(shard_from_blocks(blocks, shape, device), dims, max_bond, cutoff, group)
