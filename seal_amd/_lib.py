"""ctypes binding of libsealfm.so (C ABI declared in include/sealfm.h).

There is no Python/CPU fallback: if the shared library is missing this module
raises, and every query entry point fails with FMI_ERR_NO_DEVICE unless the
index is resident on a gfx950 GPU."""
import ctypes
import os

from . import _build

_u64 = ctypes.c_uint64
_i64 = ctypes.c_int64
_p64 = ctypes.POINTER(ctypes.c_uint64)
_pi64 = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
_int = ctypes.c_int

class FmiBeamStep(ctypes.Structure):
    """``fmi_beam_step_t`` of include/sealfm.h (one decode step of the beam loop, fmi_dev_beam_step), field for field"""
    _fields_ = [
        ("struct_bytes", _u64), ("n_groups", _u64), ("group_batch", _u64 * 3), ("group_eos", _i64 * 3), ("group_force", (_i64 * 8) * 3),
        ("group_n_force", _u64 * 3), ("group_stop", _i64 * 3), ("beams", _u64), ("cur_len", _u64), ("vocab", _u64), ("shift", _i64), ("pad_id", _i64),
        ("always_allow_eos", ctypes.c_int32), ("chain_next", ctypes.c_int32), ("d_ids", _vp), ("ids_stride", _u64), ("d_logits", _vp),
        ("d_beam_scores", _vp), ("d_first_bits", _vp), ("d_scratch", _vp), ("scratch_bytes", _u64), ("d_top_idx", _vp), ("d_top_con", _vp),
        ("d_top_unc", _vp), ("d_beam_idx", _vp), ("d_tokens_out", _vp), ("d_anc", _vp), ("anc_rows", _u64), ("anc_positions", _u64),
        ("d_hist_tok", _vp * 3), ("d_hist_sc", _vp * 3), ("hist_H", _u64 * 3), ("hist_L", _u64 * 3), ("hist_off", _u64),
        ("state_tag", _u64), ("dropped_rows", _u64)]


class _DeviceWords:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def last_constraint_bits(handle, device):
    """the allowed-token bitmap the handle's last constraint call filled, as an int32 tensor [rows, words_per_row] that ALIASES the
    handle's workspace (clone it before the next call), or None (after a cur_len == 1 step: the constant first-step mask)"""
    import torch
    rows, wpr = _u64(), _u64()
    p = lib().fmi_dev_last_constraint_bits(handle, ctypes.byref(rows), ctypes.byref(wpr))
    if not p or not rows.value:
        return None
    return torch.as_tensor(_DeviceWords(p, rows.value * wpr.value), device=device).view(rows.value, wpr.value)


# name -> (restype, argtypes); the list mirrors include/sealfm.h one to one
SIGNATURES = {
    "fmi_last_error": (ctypes.c_char_p, []),
    "fmi_abi_version": (ctypes.c_uint32, []),
    "fmi_source_digest": (ctypes.c_char_p, []),
    "fmi_create": (_int, [ctypes.POINTER(_vp)]),
    "fmi_free": (None, [_vp]),
    "fmi_view_create": (_int, [_vp, ctypes.POINTER(_vp)]),
    "fmi_build": (_int, [_vp, _p64, _u64, _int]),
    "fmi_build_from_file": (_int, [_vp, ctypes.c_char_p, _int, _int]),
    "fmi_build_device": (_int, [_vp, _vp, _u64, _int, _int]),
    "fmi_build_from_bwt_device": (_int, [_vp, _vp, _u64, _int, _u64, _int]),
    "fmi_build_device_sliced": (_int, [_vp, _vp, _u64, _int, _int, _u64]),
    "fmi_save": (_int, [_vp, ctypes.c_char_p]),
    "fmi_load": (_int, [ctypes.POINTER(_vp), ctypes.c_char_p, _int]),
    "fmi_load_sdsl": (_int, [ctypes.POINTER(_vp), ctypes.c_char_p, _int]),
    "fmi_to_device": (_int, [_vp, _int]),
    "fmi_set_doc_beginnings": (_int, [_vp, _p64, _u64]),
    "fmi_size": (_u64, [_vp]),
    "fmi_sigma": (_u64, [_vp]),
    "fmi_max_symbol": (_u64, [_vp]),
    "fmi_levels": (ctypes.c_uint32, [_vp]),
    "fmi_device": (_int, [_vp]),
    "fmi_device_bytes": (_u64, [_vp]),
    "fmi_host_array": (_vp, [_vp, ctypes.c_char_p, _p64, ctypes.POINTER(ctypes.c_uint32)]),
    "fmi_backward_search_step": (_int, [_vp, _u64, _u64, _u64, _p64]),
    "fmi_backward_search_multi": (_int, [_vp, _p64, _u64, _p64]),
    "fmi_backward_search_multi_batch": (_int, [_vp, _u64, _p64, _p64, _p64, _p64]),
    "fmi_distinct_count_multi": (_int, [_vp, _u64, _p64, _p64, _p64, _p64, _p64, _u64]),
    "fmi_locate": (_int, [_vp, _u64, _p64, _p64, _p64]),
    "fmi_extract_text": (_int, [_vp, _u64, _u64, _p64]),
    "fmi_dev_reserve": (_int, [_vp, _u64]),
    "fmi_dev_bs_step": (_int, [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "fmi_dev_get_range": (_int, [_vp, _vp, _u64, _vp, _vp, _i64, _vp, _vp]),
    "fmi_dev_constrain_scores": (_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _u64, _i64, _i64, _i64, _pi64, _u64, _i64, _int]),
    "fmi_dev_allowed_bits": (_int, [_vp, _vp, _u64, _u64, _vp, _vp, _u64, _i64, _i64, _i64, _pi64, _u64, _i64, _int]),
    "fmi_dev_allowed_bits_step": (_int, [_vp, _vp, _u64, _u64, _vp, _vp, _u64, _i64, _i64, _i64, _pi64, _u64, _i64, _int, _u64, _vp, _vp]),
    "fmi_dev_debug_timestamps": (_int, [_vp, _vp, _u64]),
    "fmi_dev_set_option": (_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    "fmi_dev_prefix_table_stats": (_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "fmi_dev_debug_marks": (_int, [_vp, _vp]),
    "fmi_dev_mark": (_int, [_vp, _vp, ctypes.c_uint32]),
    "fmi_dev_constrained_topk": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp, _u64, _i64, _i64, _i64, _pi64, _u64, _i64, _int,
                                        _vp, _vp, _u64, _vp, _vp, _vp]),
    "fmi_dev_constrained_topk_step": (_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp, _u64, _i64, _i64, _i64, _pi64, _u64, _i64, _int,
                                             _vp, _vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    "fmi_dev_constrained_topk_groups": (_int, [_vp, _vp, _u64, _p64, _pi64, _pi64, _p64, _u64, _u64, _vp, _vp, _vp, _u64, _i64, _i64, _i64, _int,
                                               _vp, _vp, _u64, _vp, _vp, _vp, _u64, _vp, _pi64]),
    "fmi_dev_locate": (_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "fmi_dev_locate_ranges": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _vp, _u64, _vp, _vp]),
    "fmi_dev_get_docs": (_int, [_vp, _vp, _u64, _vp, _vp, _i64, _vp]),
    "fmi_dev_enable_probe_count": (_int, [_vp, _int]),
    "fmi_dev_read_probe_count": (_int, [_vp, _p64]),
    "fmi_dev_read_expand_stats": (_int, [_vp, _p64]),
    "fmi_dev_enable_timing": (_int, [_vp, _int]),
    "fmi_dev_read_timing": (_int, [_vp, _p64, ctypes.POINTER(ctypes.c_double)]),
    "fmi_dev_beam_step": (_int, [_vp, _vp, ctypes.POINTER(FmiBeamStep)]),
    "fmi_dev_last_constraint_bits": (_vp, [_vp, _p64, _p64]),
    "fmi_dev_agg_timing": (_int, [_vp, _int]),
    "fmi_dev_kernel_copy": (_int, [_vp, _vp, _vp, _u64]),
    "fmi_dev_read_agg_timing": (_int, [_vp, ctypes.POINTER(ctypes.c_double), _p64, _p64]),
    "fmi_dev_call_log": (_int, [_vp, _int]),
    "fmi_dev_read_call_log": (_int, [_vp, _u64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                     ctypes.POINTER(ctypes.c_float), _p64, _p64]),
    "fmi_dev_array": (_vp, [_vp, ctypes.c_char_p, _p64, ctypes.POINTER(ctypes.c_uint32)]),
    "fmi_first_stage": (_int, [_u64, _vp, _vp, _vp, _vp, _vp, _vp, _int, ctypes.c_double, ctypes.c_double, _u64,
                               ctypes.POINTER(_vp)]),
    "fmi_evidence_docs": (_u64, [_vp]),
    "fmi_evidence_entries": (_u64, [_vp]),
    "fmi_evidence_read": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fmi_evidence_free": (None, [_vp]),
    "fmi_full_score": (_int, [_u64, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _vp, _int, ctypes.c_double, ctypes.c_double, _int, _int,
                              ctypes.POINTER(_vp)]),
    "fmi_fullscore_docs": (_u64, [_vp]),
    "fmi_fullscore_entries": (_u64, [_vp]),
    "fmi_fullscore_read": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fmi_fullscore_free": (None, [_vp]),
    "fmi_agg_pack": (_int, [_u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _u64, ctypes.POINTER(_vp)]),
    "fmi_agg_score_pack": (_int, [_u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _u64, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double, ctypes.c_double, _int, _int, _i64, _u64, _u64, _u64, ctypes.POINTER(_vp)]),
    "fmi_agg_plan_ngrams": (_u64, [_vp, _u64, _vp, _vp, _vp]),
    "fmi_agg_plan_table_src": (_u64, [_vp, _vp]),
    "fmi_agg_plan_blob": (_vp, [_vp, _p64]),
    "fmi_agg_plan_occurrences": (_u64, [_vp]),
    "fmi_agg_plan_free": (None, [_vp]),
    "fmi_dev_aggregate_sizes": (_int, [_vp, _vp, _u64, _u64, _int, _p64, _p64]),
    "fmi_dev_aggregate": (_int, [_vp, _vp, _vp, _vp, _u64, _u64, _int, ctypes.c_double, ctypes.c_double, _int, _int, _i64,
                                 _vp, _u64, _vp, _u64]),
    "fmi_log_odds_batch": (_int, [_u64, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp]),
}

_f32 = ctypes.c_float
_u32 = ctypes.c_uint32
# include/sealnn.h (fused decoder-step kernels, same shared library)
NN_SIGNATURES = {
    "sealnn_self_attn_step": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp]),
    "sealnn_cross_attn_step": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_add_layernorm": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp]),
    "sealnn_causal_self_attn": (_int, [_vp, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_tree_self_attn": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_cross_attn_rows": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_cross_attn_runs": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_self_attn_step_bf16": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp]),
    "sealnn_cross_attn_step_bf16": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_add_layernorm_bf16": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp]),
    "sealnn_causal_self_attn_bf16": (_int, [_vp, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_tree_self_attn_bf16": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_cross_attn_rows_bf16": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_cross_attn_runs_bf16": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_split_planes": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "sealnn_add_layernorm_planes": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sealnn_gelu_planes": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "sealnn_self_attn_step_acc": (_int, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp]),
    "sealnn_tree_self_attn_acc": (_int, [_vp, _vp, _vp, _f32, _vp, _u32, _u32, _u32, _f32, _vp]),
    "sealnn_add_layernorm_acc": (_int, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sealnn_gelu_planes_acc": (_int, [_vp, _vp, _vp, _f32, _u32, _u32, _vp, _vp]),
    "sealnn_split_planes_pairs": (_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "sealnn_add_layernorm_acc_slabs_pairs": (_int, [_vp, _vp, _vp, _u32, _u64, _vp, _f32, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sealnn_gelu_planes_acc_slabs_pairs": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _u32, _u32, _vp, _vp]),
    "sealnn_self_attn_step_x_pairs": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp, _vp, _vp]),
    "sealnn_cross_attn_step_x_pairs": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sealnn_finish_product": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _u32, _u32, _vp]),
    "sealnn_gelu_planes_acc_slabs": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _u32, _u32, _vp, _vp]),
    "sealnn_self_attn_step_x": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _vp, _vp, _vp, _u32, _u32, _u32, _f32, _vp, _vp, _vp, _vp]),
    "sealnn_cross_attn_step_x": (_int, [_vp, _vp, _u32, _u64, _vp, _f32, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sealnn_add_layernorm_acc_slabs": (_int, [_vp, _vp, _vp, _u32, _u64, _vp, _f32, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp]),
    "sealnn_hgemm_nt": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32]),
    "sealnn_hgemm_nt_ep": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _u32, _vp, _u32, _f32]),
}

_lib = None


class SealFMError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsealfm error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.LIB):
            raise ImportError(
                f"{_build.LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  seal_amd has no pure-Python or CPU fallback.")
        if _build.stale() and os.environ.get("SEALFM_ALLOW_STALE_LIB") != "1":
            # a binary built from other sources than the ones present (it is git-ignored and travels with snapshots): never run it silently
            raise ImportError(
                f"{_build.LIB} was built from other sources (its digest {_build.built_digest()[:16] or 'missing'}, sources "
                f"{_build.source_digest()[:16]}): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(_build.LIB)
        for name, (res, args) in list(SIGNATURES.items()) + list(NN_SIGNATURES.items()):
            fn = getattr(L, name)   # AttributeError here = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        # the digest the LOADED library reports about itself (the check above read the file; this is the mapped image)
        inside = (L.fmi_source_digest() or b"").decode("ascii", "replace")
        if inside != _build.source_digest() and os.environ.get("SEALFM_ALLOW_STALE_LIB") != "1":
            raise ImportError(f"{_build.LIB}: fmi_source_digest() = {inside[:16]}, the sources present are {_build.source_digest()[:16]}: rebuild")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise SealFMError(rc, lib().fmi_last_error().decode("utf-8", "replace"))
