"""FM-index constrained decoding on the MI355X (reference seal/beam_search.py).

``IndexBasedLogitsProcessor`` keeps the HF ``LogitsProcessor`` protocol of the
reference (beam_search.py:33-140): ``__call__(input_ids[R, t], scores[R, V]) ->
scores[R, V]`` with every token that is not a continuation of the row's prefix
in the corpus set to ``-inf``.  The reference does this with ``.tolist()``,
``get_range``/``get_count`` per row from scratch over SWIG, one ``std::async``
per row and R small H2D copies; here it is three kernels on the caller's stream
(prefix ranges, wavelet-matrix expansion into an allowed-token bitmap, masked
copy), no host round trip.
"""
import ctypes
from typing import List, Optional

import torch

from ._lib import check, lib
from .index import SHIFT, FMIndex


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class IndexBasedLogitsProcessor:
    """Drop-in for ``seal.beam_search.IndexBasedLogitsProcessor`` (beam_search.py:33-140)."""

    def __init__(
            self,
            index: FMIndex,
            num_beams: int,
            pad_token_id: int = 0,
            eos_token_id: int = 2,
            force_decoding_from: Optional[List[int]] = None,
            stop_at_count: int = 0,
            always_allow_eos: bool = False,
            forced_bos_token_id: Optional[int] = None,
    ):
        self.index = index
        self.pad_token_id = pad_token_id
        self.eos_token_id = eos_token_id
        self._num_beams = num_beams
        self.log_odds_weight = 0.0
        self.force_decoding_from = force_decoding_from
        self.force_decoding_second_token = None
        self.block_initial_stopwords = False
        self.stop_at_count = stop_at_count
        self.always_allow_eos = always_allow_eos
        self.forced_bos_token_id = forced_bos_token_id
        self._first_mask = {}

    def _first_step_mask(self, scores: torch.Tensor) -> torch.Tensor:
        """cur_len == 1: the constant ``occurring_distinct`` mask (beam_search.py:73-77)."""
        key = (scores.device, scores.shape[-1], scores.dtype)
        m = self._first_mask.get(key)
        if m is None:
            m = torch.full((scores.shape[-1],), float("-inf"), dtype=scores.dtype, device=scores.device)
            distinct = torch.as_tensor(self.index.occurring_distinct, dtype=torch.long, device=scores.device)
            m[distinct] = 0.0
            self._first_mask[key] = m
        return m

    def __call__(self, input_ids: torch.LongTensor, scores: torch.FloatTensor) -> torch.FloatTensor:
        if self.forced_bos_token_id is not None:   # beam_search.py:66-71
            if input_ids.size(1) == 1:
                mask = torch.full_like(scores, float("-inf"))
                mask[:, self.forced_bos_token_id] = 0.0
                return scores + mask
            input_ids = input_ids[:, 1:]

        if input_ids.size(1) == 1:
            out = scores + self._first_step_mask(scores)
            if self.always_allow_eos:              # beam_search.py:137-138
                out[:, self.eos_token_id] = scores[:, self.eos_token_id]
            return out

        if not scores.is_cuda:
            raise RuntimeError("IndexBasedLogitsProcessor: scores must live on the GPU that holds the FM-index "
                               "(seal_amd has no CPU path)")
        ids = input_ids.contiguous()
        if ids.dtype != torch.long:
            ids = ids.long()
        src = scores.contiguous()
        if src.dtype != torch.float32:
            src = src.float()
        out = torch.empty_like(src)
        rows, cur_len = ids.shape
        ff = self.force_decoding_from or []
        ff_arr = (ctypes.c_int64 * max(len(ff), 1))(*ff)
        check(lib().fmi_dev_constrain_scores(
            self.index.handle, _stream_ptr(scores.device), rows, cur_len, ids.data_ptr(), src.data_ptr(), out.data_ptr(),
            src.shape[-1], SHIFT, self.pad_token_id, self.eos_token_id, ff_arr, len(ff),
            int(self.stop_at_count), int(bool(self.always_allow_eos))))
        return out if out.dtype == scores.dtype else out.to(scores.dtype)
