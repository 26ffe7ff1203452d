"""ctypes binding of libndmps_hip.so (C ABI declared in include/ndmps_hip.h).

The library is looked up in-tree only (next to this file); it is built by
``__graft_entry__.build()`` / ``make -C img-compression-mps_amd/csrc``.  A missing library
is a hard error -- there is no fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libndmps_hip.so")

OK, EINVAL, EHIP, ENOCONV, EWORKSPACE, ETEAM = 0, -1, -2, -3, -4, -5

i64 = C.c_int64
p_i64 = C.POINTER(C.c_int64)
p_f64 = C.POINTER(C.c_double)
p_f32 = C.POINTER(C.c_float)
p_int = C.POINTER(C.c_int)
vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/ndmps_hip.h declares
SIGNATURES = {
    "ndmps_version": (C.c_int, []),
    "ndmps_last_error": (C.c_char_p, []),
    "ndmps_device_count": (C.c_int, []),
    "ndmps_streams_create": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(C.c_int)]),
    "ndmps_streams_destroy": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "ndmps_profile_enable": (C.c_int, [C.c_int]),
    "ndmps_profile_collect": (C.c_int, [C.c_int, p_f64, p_i64, p_i64]),
    "ndmps_plan_create": (C.c_int, [C.POINTER(vp), C.c_int, p_i64, C.c_int, p_i64]),
    "ndmps_plan_create_reversed": (C.c_int, [C.POINTER(vp), C.c_int, p_i64, C.c_int, p_i64]),
    "ndmps_plan_destroy": (C.c_int, [vp]),
    "ndmps_plan_numel": (i64, [vp]),
    "ndmps_plan_is_tiled": (C.c_int, [vp]),
    "ndmps_plan_emulate": (C.c_int, [C.c_int, p_i64, C.c_int, p_i64, C.c_int, p_i64]),
    "ndmps_plan_emulate_reversed": (C.c_int, [C.c_int, p_i64, C.c_int, p_i64, C.c_int, p_i64]),
    "ndmps_encode_permute": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "ndmps_decode_permute": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "ndmps_encode_permute_many": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, vp]),
    "ndmps_decode_permute_many": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, vp]),
    "ndmps_encode_permute_generic": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "ndmps_decode_permute_generic": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "ndmps_dct_basis_f32": (C.c_int, [vp, i64, vp]),
    "ndmps_dct_last_f32": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "ndmps_idct_last_f32": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "ndmps_dct_last_many_f32": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(vp), i64, i64, vp, vp]),
    "ndmps_idct_last_many_f32": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(vp), i64, i64, vp, vp]),
    "ndmps_sumsq_f32": (C.c_int, [vp, i64, p_f64, vp, i64, vp]),
    "ndmps_minmax_f32": (C.c_int, [vp, i64, p_f32, p_f32, vp, i64, vp]),
    "ndmps_minmax_many_workspace_bytes": (i64, [C.c_int]),
    "ndmps_minmax_partials_bytes": (i64, [C.c_int]),
    "ndmps_minmax_arena_launch_f32": (C.c_int, [vp, i64, C.c_int, C.c_int, p_i64, p_i64, vp, vp]),
    "ndmps_minmax_collect": (C.c_int, [C.c_int, vp, C.POINTER(C.c_float), C.POINTER(C.c_double), vp]),
    "ndmps_minmax_many_f32": (C.c_int, [C.c_int, C.POINTER(vp), p_i64, p_f32, p_f64, vp, i64, vp]),
    "ndmps_scale_f32": (C.c_int, [vp, i64, C.c_double, vp]),
    "ndmps_scale_many_f32": (C.c_int, [C.c_int, C.POINTER(vp), i64, p_f64, vp]),
    "ndmps_reduce_workspace_bytes": (i64, []),
    "ndmps_sgemm": (C.c_int, [C.c_int, C.c_int, i64, i64, i64, vp, i64, vp, i64, vp, i64, vp]),
    "ndmps_dgemm": (C.c_int, [C.c_int, C.c_int, i64, i64, i64, vp, i64, vp, i64, vp, i64, vp]),
    "ndmps_gram_workspace_bytes": (i64, [i64, i64]),
    "ndmps_gram_f32": (C.c_int, [vp, i64, i64, i64, vp, vp, i64, vp]),
    "ndmps_gemm_bf16_workspace_bytes": (i64, [C.c_int, i64, i64]),
    "ndmps_gemm_bf16": (C.c_int, [C.c_int, i64, i64, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp]),
    "ndmps_gram_batched_workspace_bytes": (i64, [C.c_int, i64, i64]),
    "ndmps_gram_batched_f32": (C.c_int, [C.c_int, C.POINTER(vp), i64, i64, i64, vp, i64, vp, i64, vp]),
    "ndmps_gram_batched_bf16": (C.c_int, [C.c_int, C.POINTER(vp), i64, i64, i64, vp, i64, vp, i64, vp]),
    "ndmps_gram_batched_indexed_f32": (C.c_int, [C.c_int, C.POINTER(vp), i64, i64, vp, vp, vp, vp, i64, vp, i64, vp]),
    "ndmps_gram_bf16": (C.c_int, [vp, i64, i64, i64, vp, vp, i64, vp]),
    "ndmps_convert_bf16_to_f32": (C.c_int, [vp, i64, vp, vp]),
    "ndmps_convert_f32_to_bf16": (C.c_int, [vp, i64, vp, vp]),
    "ndmps_tt_merge_columns": (i64, [C.c_int, p_i64, i64]),
    "ndmps_tt_sweep_batched_fused_f32": (C.c_int, [C.c_int, C.POINTER(vp), C.c_int, p_i64, C.c_double, i64,
                                                   C.POINTER(vp), p_i64, p_i64, p_f64, p_i64, vp, vp, vp, vp, vp, i64,
                                                   vp, i64, vp]),
    "ndmps_sgemm_gathered64_stream_batched": (C.c_int, [C.c_int, i64, i64, C.POINTER(vp), vp, vp, vp, C.POINTER(vp), i64,
                                                        C.POINTER(vp), i64, vp]),
    "ndmps_gram_indexed_f32": (C.c_int, [vp, i64, i64, vp, vp, vp, vp, vp, i64, vp]),
    "ndmps_tt_sweep_batched_bf16": (C.c_int, [C.c_int, C.POINTER(vp), C.c_int, p_i64, C.c_double, i64,
                                              C.POINTER(vp), p_i64, p_i64, p_f64, p_i64, vp, i64, vp]),
    "ndmps_chain_tail_columns": (i64, [C.c_int, p_i64]),
    "ndmps_plan_split_offsets": (C.c_int, [vp, i64, p_i64, p_i64]),
    "ndmps_chain_contract_scatter_f32": (C.c_int, [C.c_int, p_i64, p_i64, C.POINTER(vp), vp, vp, vp, vp, i64, vp, i64, vp]),
    "ndmps_chain_batched_workspace_bytes": (i64, [C.c_int, C.c_int, p_i64, p_i64]),
    "ndmps_chain_contract_scatter_batched_f32": (C.c_int, [C.c_int, C.c_int, p_i64, p_i64, C.POINTER(vp), C.POINTER(vp), vp, vp, vp,
                                                           i64, vp, i64, vp]),
    "ndmps_gemm_batched_max": (C.c_int, []),
    "ndmps_sgemm_batched": (C.c_int, [C.c_int, C.c_int, C.c_int, i64, i64, i64, C.POINTER(vp), i64, C.POINTER(vp), i64,
                                      C.POINTER(vp), i64, vp]),
    "ndmps_dgemm_batched": (C.c_int, [C.c_int, C.c_int, C.c_int, i64, i64, i64, C.POINTER(vp), i64, C.POINTER(vp), i64,
                                      C.POINTER(vp), i64, vp]),
    "ndmps_sgemm_indexed_batched": (C.c_int, [C.c_int, i64, i64, i64, C.POINTER(vp), i64, vp, vp, C.c_int, C.POINTER(vp), i64,
                                              C.POINTER(vp), i64, vp, vp, vp]),
    "ndmps_sgemm_indexed": (C.c_int, [i64, i64, i64, vp, i64, vp, vp, C.c_int, vp, i64, vp, i64, vp, vp, vp]),
    "ndmps_chain_contract_bf16": (C.c_int, [C.c_int, p_i64, p_i64, C.POINTER(vp), vp, vp, i64, vp]),
    "ndmps_syevj_workspace_bytes": (i64, [i64]),
    "ndmps_syevj_f64": (C.c_int, [vp, i64, vp, vp, vp, i64, p_int, vp]),
    "ndmps_syevj_batched_workspace_bytes": (i64, [i64, C.c_int]),
    "ndmps_syevj_batched_f64": (C.c_int, [C.c_int, vp, i64, p_i64, vp, i64, vp, i64, vp, i64, p_int, vp]),
    "ndmps_syevj_batched_tol_f64": (C.c_int, [C.c_int, vp, i64, p_i64, vp, i64, vp, i64, C.c_double, vp, i64,
                                              p_int, vp]),
    "ndmps_syevj_batched_values_f64": (C.c_int, [C.c_int, vp, i64, p_i64, vp, i64, vp, i64, C.c_double, vp, i64,
                                                 p_int, vp]),
    "ndmps_syevj_batched_vectors_f64": (C.c_int, [C.c_int, vp, i64, p_i64, vp, i64, vp, i64, p_i64, vp, i64, vp]),
    "ndmps_syevd_topk_max_n": (i64, []),
    "ndmps_syevd_topk_max_k": (i64, []),
    "ndmps_syevd_topk_max_k_wide": (i64, []),
    "ndmps_potrf_scratch_elems": (i64, [i64]),
    "ndmps_potrf_lower_f64": (C.c_int, [vp, i64, vp, C.POINTER(C.c_int), vp]),
    "ndmps_syevd_topk_workspace_bytes": (i64, [i64, C.c_int, i64]),
    "ndmps_syevd_topk_stamps_offset": (i64, [i64, C.c_int, i64]),
    "ndmps_syevd_topk_values_f64": (C.c_int, [C.c_int, vp, i64, p_i64, vp, i64, vp, i64, i64, vp, i64, vp]),
    "ndmps_syevd_topk_vectors_f64": (C.c_int, [C.c_int, p_i64, p_i64, i64, vp, i64, p_int, vp]),
    "ndmps_syevd_topk_vectors_auto_f64": (C.c_int, [C.c_int, p_i64, i64, C.c_double, vp, vp, i64, vp, vp, i64, vp]),
    "ndmps_syevd_topk_recover_f64": (C.c_int, [C.c_int, p_i64, i64, vp, i64, p_int, vp]),
    "ndmps_syevd_topk_set_team": (C.c_int, [C.c_int]),
    "ndmps_syevd_topk_set_streamed": (C.c_int, [C.c_int]),
    "ndmps_syevd_topk_team_fallbacks": (i64, []),
    "ndmps_syevd_topk_team_slots": (C.c_int, [i64]),
    "ndmps_syevd_topk_note_team_fallback": (C.c_int, []),
    "ndmps_debug_inject_team_abort": (C.c_int, [C.c_int]),
    "ndmps_debug_lane_sums_f64": (C.c_int, [vp, vp, vp]),
    "ndmps_tt_sweep_pads_cores": (C.c_int, [C.c_int, p_i64, i64]),
    "ndmps_tt_sweep_async_ints": (i64, [C.c_int, C.c_int]),
    "ndmps_tt_sweep_async_doubles": (i64, [C.c_int, C.c_int, p_i64, i64]),
    "ndmps_tt_sweep_batched_fused_begin_f32": (C.c_int, [C.c_int, C.POINTER(vp), C.c_int, p_i64, C.c_double, i64,
                                                         C.POINTER(vp), p_i64, p_i64, vp, vp, vp, vp, vp, i64,
                                                         vp, i64, vp, vp, vp]),
    "ndmps_tt_sweep_finish": (C.c_int, [C.c_int, C.c_int, p_i64, i64, vp, vp, p_i64, p_f64, p_i64]),
    "ndmps_tt_sweep_batched_workspace_bytes": (i64, [C.c_int, C.c_int, p_i64, i64]),
    "ndmps_tt_sweep_batched_f32": (C.c_int, [C.c_int, C.POINTER(vp), C.c_int, p_i64, C.c_double, i64,
                                             C.POINTER(vp), p_i64, p_i64, p_f64, p_i64, vp, i64, vp]),
    "ndmps_tt_layout": (C.c_int, [C.c_int, p_i64, i64, p_i64, p_i64, p_i64, p_i64]),
    "ndmps_tt_sweep_f32": (C.c_int, [vp, C.c_int, p_i64, C.c_double, i64, vp, p_i64, p_i64, p_f64,
                                     p_i64, vp, i64, vp]),
    "ndmps_compress_bond_workspace_bytes": (i64, [i64, i64, i64, i64, i64]),
    "ndmps_compress_bond_f32": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, C.c_double, i64, vp, vp,
                                          p_i64, p_f64, vp, i64, vp]),
    "ndmps_chain_workspace_bytes": (i64, [C.c_int, p_i64, p_i64]),
    "ndmps_chain_contract_f32": (C.c_int, [C.c_int, p_i64, p_i64, C.POINTER(vp), vp, vp, i64, vp]),
    "ndmps_overlap_workspace_bytes": (i64, [C.c_int, p_i64, p_i64, p_i64]),
    "ndmps_overlap_f32": (C.c_int, [C.c_int, p_i64, p_i64, C.POINTER(vp), p_i64, C.POINTER(vp), p_f64,
                                    vp, i64, vp]),
    "ndmps_ssim_workspace_bytes": (i64, [C.c_int, p_i64]),
    "ndmps_ssim_f32": (C.c_int, [vp, vp, C.c_int, p_i64, p_f64, vp, i64, vp]),
    "ndmps_ssim_slices_workspace_bytes": (i64, [p_i64, C.c_int]),
    "ndmps_ssim_slices_f32": (C.c_int, [vp, vp, p_i64, C.c_int, p_f64, vp, i64, vp]),
    "ndmps_psnr_workspace_bytes": (i64, []),
    "ndmps_psnr_f32": (C.c_int, [vp, vp, i64, p_f64, vp, i64, vp]),
    "ndmps_quantize_f32": (C.c_int, [vp, i64, C.c_float, C.c_float, C.c_int, vp, vp]),
    "ndmps_dequantize_f32": (C.c_int, [vp, i64, C.c_float, C.c_float, C.c_int, vp, vp]),
    # fp64 storage (the reference's own element type)
    "ndmps_gram_f64_workspace_bytes": (i64, [i64, i64]),
    "ndmps_gram_f64": (C.c_int, [vp, i64, i64, i64, vp, vp, i64, vp]),
    "ndmps_tt_sweep_batched_workspace_bytes_f64": (i64, [C.c_int, C.c_int, p_i64, i64]),
    "ndmps_tt_sweep_batched_f64": (C.c_int, [C.c_int, C.POINTER(vp), C.c_int, p_i64, C.c_double, i64,
                                             C.POINTER(vp), p_i64, p_i64, p_f64, p_i64, vp, i64, vp]),
    "ndmps_compress_bond_f64": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, C.c_double, i64, vp, vp,
                                          p_i64, p_f64, vp, i64, vp]),
    "ndmps_chain_workspace_bytes_f64": (i64, [C.c_int, p_i64, p_i64]),
    "ndmps_chain_contract_f64": (C.c_int, [C.c_int, p_i64, p_i64, C.POINTER(vp), vp, vp, i64, vp]),
    "ndmps_overlap_f64": (C.c_int, [C.c_int, p_i64, p_i64, C.POINTER(vp), p_i64, C.POINTER(vp), p_f64,
                                    vp, i64, vp]),
    "ndmps_sumsq_f64": (C.c_int, [vp, i64, p_f64, vp, i64, vp]),
    "ndmps_scale_f64": (C.c_int, [vp, i64, C.c_double, vp]),
    "ndmps_minmax_many_f64": (C.c_int, [C.c_int, C.POINTER(vp), p_i64, p_f64, p_f64, vp, i64, vp]),
    "ndmps_dct_basis_f64": (C.c_int, [vp, i64, vp]),
    "ndmps_dct_last_f64": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "ndmps_idct_last_f64": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "ndmps_quantize_f64": (C.c_int, [vp, i64, C.c_double, C.c_double, C.c_int, vp, vp]),
    "ndmps_dequantize_f64": (C.c_int, [vp, i64, C.c_double, C.c_double, C.c_int, vp, vp]),
}

_lib = None


class NdmpsHipError(RuntimeError):
    pass


class NdmpsTeamAbort(NdmpsHipError):
    """A resident tridiagonalisation gave up waiting for its workgroups (NDMPS_ETEAM): the call's outputs are
    invalid; the caller repeats it with the resident launch switched off (ndmps_syevd_topk_set_team(0))."""


def load():
    """Load libndmps_hip.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C img-compression-mps_amd/csrc`.  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the binding drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Map a return code to the exception the reference would raise (Errors)."""
    if rc >= 0:
        return rc
    msg = load().ndmps_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ETEAM:
        raise NdmpsTeamAbort(f"libndmps_hip error {rc}: {msg}")
    raise NdmpsHipError(f"libndmps_hip error {rc}: {msg}")


def i64_array(values):
    arr = (C.c_int64 * len(values))(*[int(v) for v in values])
    return arr


def f64_array(values):
    return (C.c_double * len(values))(*[float(v) for v in values])


def require_device():
    """Fail loudly when there is no HIP device (the product never computes on the CPU)."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "imgcompressionmps_amd needs an AMD GPU (HIP device); none is visible and there is no CPU path."
        )
    load()


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
