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

# name -> (restype, argtypes); the list mirrors include/sealfm.h one to one
SIGNATURES = {
    "fmi_last_error": (ctypes.c_char_p, []),
    "fmi_abi_version": (ctypes.c_uint32, []),
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
    "fmi_dev_agg_timing": (_int, [_vp, _int]),
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
                f"{_build.LIB} was built from other sources (stamp {_build.built_digest()[:16] or 'missing'}, sources "
                f"{_build.source_digest()[:16]}): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(_build.LIB)
        for name, (res, args) in list(SIGNATURES.items()) + list(NN_SIGNATURES.items()):
            fn = getattr(L, name)   # AttributeError here = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise SealFMError(rc, lib().fmi_last_error().decode("utf-8", "replace"))
