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
import itertools
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

    # -- fused decode step ----------------------------------------------------
    def supports_fused_topk(self, logits: torch.Tensor, num_beams: int) -> bool:
        return (logits.is_cuda and logits.dtype == torch.float32 and self.forced_bos_token_id is None
                and 2 * num_beams <= 64 and len(self.force_decoding_from or []) <= MAX_FORCE)    # longer forced prefixes: the unfused path

    def _first_bits(self, vocab: int, device) -> torch.Tensor:
        """allowed-token bitmap of the first step (``occurring_distinct``, beam_search.py:73-77), cached on the index"""
        cache = self.index.__dict__.setdefault("_first_bits_cache", {})
        key = (str(device), vocab, bool(self.always_allow_eos), self.eos_token_id if self.always_allow_eos else -1)
        b = cache.get(key)
        if b is None:
            allowed = torch.zeros(((vocab + 31) // 32) * 32, dtype=torch.bool)
            allowed[torch.as_tensor(self.index.occurring_distinct, dtype=torch.long)] = True
            if self.always_allow_eos:
                allowed[self.eos_token_id] = True
            w = allowed.view(-1, 32).to(torch.int64) << torch.arange(32, dtype=torch.int64)
            b = w.sum(1).to(torch.int64)
            b = torch.where(b >= 2 ** 31, b - 2 ** 32, b).to(torch.int32).to(device)
            cache[key] = b
        return b

    def fused_topk(self, input_ids: torch.LongTensor, logits: torch.FloatTensor, beam_scores: torch.FloatTensor,
                   batch: int, num_beams: int, parent_rows: Optional[torch.LongTensor] = None, tag: Optional[int] = None):
        """log_softmax + InfNanRemove + this constraint + beam scores + top-2K per query, fused
        (``fmi_dev_constrained_topk_step``): returns (flat indices [B, 2K], unconstrained scores [B, 2K]) -- what
        reference beam_search.py:244-307 computes through five [rows, vocab] intermediates.

        ``parent_rows`` (the previous step's ``beam_idx``) lets the index advance every row's prefix
        range by one backward-search step instead of re-searching the prefix (same ranges)."""
        return fused_topk_groups([self], [batch], input_ids, logits, beam_scores, num_beams, parent_rows=parent_rows,
                                 tag=tag if tag is not None else ((id(self) & 0x7FFFFFFFFFFFFFFF) or 1))

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
        if getattr(self.index, "_trace", None) is not None:
            self.index._trace.append(("mask", ids.clone(), list(self.force_decoding_from or []),
                                      dict(pad=self.pad_token_id, eos=self.eos_token_id, stop_at_count=int(self.stop_at_count),
                                           always_allow_eos=bool(self.always_allow_eos))))
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


# the beam loop's bookkeeping between two model steps as ONE library call (fmi_dev_beam_step: _BeamStepper); ``FUSED_BEAM_STEP = False`` keeps
# round 4's form (fmi_dev_constrained_topk_groups + ~15 torch launches per step): same histories, the GPU tests run both
FUSED_BEAM_STEP = True
MAX_FORCE = 8          # tokens of force_decoding_from the constraint kernel takes (fmi_kernels.hip)
MAX_ROW_GROUPS = 3     # decodes one constraint call can serve in lockstep (fmi_kernels.hip)


def can_fuse_groups(procs, logits: torch.Tensor, num_beams: int) -> bool:
    """one fused constraint call can serve these processors' rows together: the HIP path applies to each, they constrain
    against the same index handle and share everything but the end-of-sequence token and the forced prefix"""
    p0 = procs[0]
    return (0 < len(procs) <= MAX_ROW_GROUPS
            and all(isinstance(p, IndexBasedLogitsProcessor) and p.supports_fused_topk(logits, num_beams) for p in procs)
            and all(p.index is p0.index and p.pad_token_id == p0.pad_token_id
                    and bool(p.always_allow_eos) == bool(p0.always_allow_eos) and len(p.force_decoding_from or []) <= MAX_FORCE for p in procs)
            and (len(procs) == 1 or not p0.always_allow_eos))      # (the first-step bitmap folds ONE eos in)


def fused_topk_groups(procs, batches, input_ids: torch.LongTensor, logits: torch.FloatTensor, beam_scores: torch.FloatTensor,
                      num_beams: int, parent_rows: Optional[torch.LongTensor] = None, tag: int = 0):
    """``IndexBasedLogitsProcessor.fused_topk`` for the stacked rows of several decodes that advance in lockstep (group g =
    the next ``batches[g]`` queries, constrained by ``procs[g]``): ONE ``fmi_dev_constrained_topk_groups`` call -- one
    constraint launch, one log-softmax/top-2K launch, one merge launch -- for all of them.  Same per-row masks and per-query
    picks as one call per processor."""
    p0 = procs[0]
    dev = logits.device
    V = logits.shape[-1]
    want = 2 * num_beams
    batch = int(sum(batches))
    rows = batch * num_beams
    ids = input_ids.contiguous()
    cur_len = ids.shape[1]
    scratch = p0._first_mask.get(("scratch", dev, rows, want))
    if scratch is None:
        scratch = torch.empty(rows * (3 + 2 * want) + 64, dtype=torch.float32, device=dev)
        p0._first_mask[("scratch", dev, rows, want)] = scratch
    top_idx = torch.empty(batch, want, dtype=torch.int64, device=dev)
    top_con = torch.empty(batch, want, dtype=torch.float32, device=dev)
    top_unc = torch.empty(batch, want, dtype=torch.float32, device=dev)
    n = len(procs)
    g_batch = (ctypes.c_uint64 * n)(*[int(b) for b in batches])
    g_eos = (ctypes.c_int64 * n)(*[int(p.eos_token_id) for p in procs])
    g_nff = (ctypes.c_uint64 * n)(*[len(p.force_decoding_from or []) for p in procs])
    g_stop = (ctypes.c_int64 * n)(*[max(0, int(p.stop_at_count)) for p in procs])      # per decode: the reference gives it to the body decode only
    g_ff = (ctypes.c_int64 * (n * MAX_FORCE))()
    for g, p in enumerate(procs):
        for j, t in enumerate(p.force_decoding_from or []):
            g_ff[(g * MAX_FORCE if n > 1 else 0) + j] = int(t)
    first = p0._first_bits(V, dev) if cur_len == 1 else None
    if getattr(p0.index, "_trace", None) is not None and cur_len >= 2:
        a = 0
        for p, b in zip(procs, batches):         # one recorded operation per decode: its rows, its arguments
            p0.index._trace.append(("mask", ids[a:a + b * num_beams].clone(), list(p.force_decoding_from or []),
                                    dict(pad=p.pad_token_id, eos=p.eos_token_id, stop_at_count=int(p.stop_at_count),
                                         always_allow_eos=bool(p.always_allow_eos))))
            a += b * num_beams
    lg = logits.contiguous()
    bs = beam_scores.contiguous()
    parent = parent_rows.contiguous() if parent_rows is not None else None
    check(lib().fmi_dev_constrained_topk_groups(
        p0.index.handle, _stream_ptr(dev), n, g_batch, g_eos, g_ff, g_nff, num_beams, cur_len, ids.data_ptr(), lg.data_ptr(), bs.data_ptr(),
        V, SHIFT, p0.pad_token_id, int(p0.stop_at_count), int(bool(p0.always_allow_eos)),
        first.data_ptr() if first is not None else None, scratch.data_ptr(), scratch.numel() * 4,
        top_idx.data_ptr(), top_con.data_ptr(), top_unc.data_ptr(), int(tag), parent.data_ptr() if parent is not None else None, g_stop))
    return top_idx, top_unc


# ---------------------------------------------------------------------------
# beam loop (reference beam_search.py:143-389) + keep-history scorer
# (reference beam_search.py:559-758), tensorised: no .item()/.tolist() per step,
# one D2H transfer of the recorded history at the end.
# ---------------------------------------------------------------------------
def _inf_nan_remove(scores: torch.Tensor) -> torch.Tensor:
    """HF 4.13 ``InfNanRemoveLogitsProcessor`` (the only stock processor that
    ``_get_logits_processor`` adds here, beam_search.py:430-445): nan -> 0,
    +inf -> finfo.max; -inf is left alone."""
    scores = torch.where(scores != scores, torch.zeros_like(scores), scores)
    return torch.where(scores == float("inf"), torch.full_like(scores, torch.finfo(scores.dtype).max), scores)


class _Steps(list):
    """the recorded steps of one decode; ``packed`` = (tokens [batch, H, L], scores [batch, H]) when the steps are views of the packed
    history that ``fmi_dev_beam_step`` wrote in place (``PendingGenerate`` then has nothing to copy together)"""
    packed = None


class _BeamStepper:
    """The beam loop's state as device buffers that ``fmi_dev_beam_step`` advances in place -- ONE library call per decode step:
    constraint (started from the chains the previous step's ``k_beam_advance`` ran) + log-softmax + top-2K + merge, then
    ``k_beam_advance``: the keep-history scorer's bookkeeping (reference beam_search.py:658-685), ``input_ids`` = cat(input_ids[beam_idx],
    tokens) (:313), the decoder's ancestry table, and the chains of the NEXT constraint call.  Replaces ~15 torch launches per step.
    Rows are the stacked rows of the live groups; a group that ends leaves at the front (``drop``)."""

    def __init__(self, specs, num_beams: int, decoder_start_token_id: int, device):
        from ._lib import FmiBeamStep
        K = self.K = num_beams
        self.specs = specs
        B = sum(sp["batch"] for sp in specs)
        R = B * K
        t_max = max(sp["max_length"] for sp in specs)
        p0 = specs[0]["processor"]
        self.index = p0.index
        self.ids = torch.full((R, t_max + 1), int(p0.pad_token_id), dtype=torch.long, device=device)
        self.ids[:, 0] = decoder_start_token_id
        bs = torch.zeros(B, K, dtype=torch.float32, device=device)
        bs[:, 1:] = -1e9
        self.beam_scores = bs.view(R)
        self.beam_idx = torch.empty(R, dtype=torch.long, device=device)
        self.tokens = torch.empty(R, dtype=torch.long, device=device)
        want = 2 * K
        self.top_idx = torch.empty(B, want, dtype=torch.int64, device=device)
        self.top_con = torch.empty(B, want, dtype=torch.float32, device=device)
        self.top_unc = torch.empty(B, want, dtype=torch.float32, device=device)
        self.scratch = torch.empty(R * (3 + 2 * want) + 64, dtype=torch.float32, device=device)
        # the decodes' histories, packed as PendingGenerate hands them to the host: tokens [batch, H, L] (-1 beyond a hypothesis), scores [batch, H]
        self.hist = []
        for sp in specs:
            L = sp["max_length"]
            H = (L - 1) * want + K
            self.hist.append((torch.full((sp["batch"], H, L), -1, dtype=torch.int64, device=device), torch.empty(sp["batch"], H, dtype=torch.float32, device=device)))
        self.row0 = 0                 # rows of the groups that have left
        self.dropped = 0              # ... since the previous step
        self.step_no = 0
        self.st = FmiBeamStep()
        self.st.struct_bytes = ctypes.sizeof(FmiBeamStep)

    def input_ids(self, cur_len: int) -> torch.Tensor:
        return self.ids[self.row0:, :cur_len]

    def drop(self, rows: int) -> None:
        self.row0 += rows
        self.dropped += rows

    def step(self, live, logits: torch.Tensor, cur_len: int, tag: int, chain_next: bool, tokens_out, anc):
        """one decode step for the live groups; returns the (prefix, tokens, scores) views of each live group's history"""
        K, st, dev = self.K, self.st, logits.device
        want = 2 * K
        procs = [self.specs[g]["processor"] for g in live]
        p0 = procs[0]
        V = logits.shape[-1]
        n = len(live)
        batch = 0
        for i, g in enumerate(live):
            p, sp = procs[i], self.specs[g]
            st.group_batch[i] = sp["batch"]
            st.group_eos[i] = int(p.eos_token_id)
            ff = p.force_decoding_from or []
            st.group_n_force[i] = len(ff)
            for j, tk in enumerate(ff):
                st.group_force[i][j] = int(tk)
            st.group_stop[i] = max(0, int(p.stop_at_count))
            tok, sc = self.hist[g]
            st.d_hist_tok[i], st.d_hist_sc[i] = tok.data_ptr(), sc.data_ptr()
            st.hist_H[i], st.hist_L[i] = tok.shape[1], tok.shape[2]
            batch += sp["batch"]
        for i in range(n, 3):
            st.group_batch[i] = 0
            st.d_hist_tok[i] = st.d_hist_sc[i] = None
        rows = batch * K
        st.n_groups, st.beams, st.cur_len, st.vocab = n, K, cur_len, V
        st.shift, st.pad_id = SHIFT, int(p0.pad_token_id)
        st.always_allow_eos, st.chain_next = int(bool(p0.always_allow_eos)), int(bool(chain_next))
        ids = self.ids[self.row0:]
        st.d_ids, st.ids_stride = ids.data_ptr(), self.ids.stride(0)
        lg = logits.contiguous()
        st.d_logits = lg.data_ptr()
        st.d_beam_scores = self.beam_scores[self.row0:].data_ptr()
        first = p0._first_bits(V, dev) if cur_len == 1 else None
        st.d_first_bits = first.data_ptr() if first is not None else None
        st.d_scratch, st.scratch_bytes = self.scratch.data_ptr(), self.scratch.numel() * 4
        q0 = self.row0 // K
        st.d_top_idx, st.d_top_con, st.d_top_unc = self.top_idx[q0:].data_ptr(), self.top_con[q0:].data_ptr(), self.top_unc[q0:].data_ptr()
        st.d_beam_idx = self.beam_idx[self.row0:].data_ptr()
        out_tok = tokens_out if tokens_out is not None else self.tokens[self.row0:]
        assert out_tok.numel() == rows and out_tok.is_contiguous() and out_tok.dtype == torch.long
        st.d_tokens_out = out_tok.data_ptr()
        if anc is not None:
            assert anc.dtype == torch.int32 and anc.is_contiguous() and anc.shape[1] == rows
            st.d_anc, st.anc_rows, st.anc_positions = anc.data_ptr(), anc.shape[1], anc.shape[0]
        else:
            st.d_anc, st.anc_rows, st.anc_positions = None, 0, 0
        off = self.step_no * want
        st.hist_off = off
        st.state_tag, st.dropped_rows = int(tag), self.dropped
        trace = getattr(self.index, "_trace", None)
        ids_before = self.ids[self.row0:, :cur_len].clone() if (trace is not None and cur_len >= 2) else None
        check(lib().fmi_dev_beam_step(self.index.handle, _stream_ptr(dev), ctypes.byref(st)))
        if ids_before is not None:
            # one recorded operation per decode: its rows, its arguments, and the bitmap the step ACTUALLY applied (table / chained /
            # generic form alike): bench.py's parity check holds that bitmap, not a recomputation, to the CPU oracle
            from ._lib import last_constraint_bits
            bits = last_constraint_bits(self.index.handle, dev)
            bits = bits[:rows].clone() if bits is not None else None
            a = 0
            for p, g in zip(procs, live):
                b = self.specs[g]["batch"] * K
                trace.append(("mask", ids_before[a:a + b], list(p.force_decoding_from or []),
                              dict(pad=p.pad_token_id, eos=p.eos_token_id, stop_at_count=int(p.stop_at_count), always_allow_eos=bool(p.always_allow_eos)),
                              bits[a:a + b] if bits is not None else None))
                a += b
        self.dropped = 0
        self.step_no += 1
        views = []
        for g in live:
            tok, sc = self.hist[g]
            views.append((tok[:, off:off + want, :cur_len], tok[:, off:off + want, cur_len], sc[:, off:off + want]))
        return views, out_tok


_LOOP_TAGS = itertools.count(1)      # one continuity tag per decode loop (fmi_dev_constrained_topk_step's state_tag)
_DEBUG_MARK = None                   # tools/soak.py: callable(code) that writes a progress word in stream order (which launch of a stalled decode never completed)


@torch.no_grad()
def constrained_beam_search(decoder, batch_size: int, num_beams: int, max_length: int, decoder_start_token_id: int,
                            eos_token_id: int, constrained_decoding_processor=None, device=None, fused: bool = True):
    """Runs the loop of reference ``constrained_beam_search`` with the
    ``BeamSearchScorerWithMemory`` bookkeeping and returns the raw history:

    ``steps``: list over decode steps of (prefix_ids [B, 2K, t], tokens [B, 2K],
    sum_logprobs [B, 2K]) -- every ranked candidate of every step, in rank order
    (scorer.process, beam_search.py:658-668) -- and ``final``: (input_ids [R, T],
    beam_scores [R]) re-added by ``finalize`` (beam_search.py:717-725).

    Semantics kept: beam scores start at [0, -1e9, ...] (214-216); fp32
    log_softmax (251); top-2K is taken on the CONSTRAINED scores but the value
    carried on is the UNCONSTRAINED one (302-307); the first K non-eos candidates
    continue (673-685); stop when the sequence length reaches max_length (340).
    """
    spec = dict(batch=batch_size, max_length=max_length, eos_token_id=eos_token_id, processor=constrained_decoding_processor)
    return constrained_beam_search_groups(decoder, [spec], num_beams, decoder_start_token_id, device=device, fused=fused)[0]


@torch.no_grad()
def constrained_beam_search_groups(decoder, specs, num_beams: int, decoder_start_token_id: int, device=None, fused: bool = True):
    """The loop above for SEVERAL decodes in lockstep, their rows stacked (group g = ``specs[g]["batch"]`` queries with its
    own ``max_length`` / ``eos_token_id`` / ``processor``, max lengths ascending in row order): one model step, one
    constraint call and one top-2K per step for all of them -- the searcher's body and title decodes of a batch are 2 x
    batch x beams rows per step instead of two loops (reference retrieval.py:70-83 and 162-176 run one ``generate`` after
    the other).  A group that reaches its max_length is finalised and its rows leave the loop (``decoder.narrow``); every
    group's history is what its own loop records.  Returns [(steps, final)] per group."""
    K = num_beams
    assert all(a["max_length"] <= b["max_length"] for a, b in zip(specs, specs[1:])), "groups must come in ascending max_length"
    live = list(range(len(specs)))
    B = sum(sp["batch"] for sp in specs)
    R = B * K
    input_ids = torch.full((R, 1), decoder_start_token_id, dtype=torch.long, device=device)
    beam_scores = torch.zeros(B, K, dtype=torch.float32, device=device)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(R)
    row_base = (torch.arange(B, device=device) * K).unsqueeze(1)
    eos_q = torch.cat([torch.full((sp["batch"],), int(sp["eos_token_id"]), dtype=torch.long, device=device) for sp in specs]).unsqueeze(1)
    out_steps = [_Steps() for _ in specs]
    finals = [None] * len(specs)
    beam_idx = None     # rows of the previous step that this step's rows extend (incremental constraint state)
    first_logits = None
    tag = next(_LOOP_TAGS)
    shared_first = bool(getattr(decoder, "shared_first_step", False))
    stepper = None                 # the fused beam step (fmi_dev_beam_step): the loop's state lives in its buffers
    next_tokens_in = None          # ... and the decoder's next input is already where the decoder reads it
    while True:
        if first_logits is None and shared_first:
            logits = decoder.step(input_ids[:, -1], beams_identical=True)     # every beam starts from decoder_start_token_id
        elif next_tokens_in is not None:
            logits = decoder.step(next_tokens_in)
        else:
            logits = decoder.step(input_ids[:, -1])
        if _DEBUG_MARK is not None:
            _DEBUG_MARK(1000 * tag + 10 * input_ids.shape[-1] + 1)       # the model step of this position is behind us
        V = logits.shape[-1]
        if first_logits is None:
            # every beam of a query sees the same first step: the model's next-token logits after the start token,
            # i.e. what compute_unigram_scores (keys.py:145-176) runs the whole model for
            first_logits = logits.view(B, K, V)[:, 0].clone()
        procs = [specs[g]["processor"] for g in live]
        batches = [specs[g]["batch"] for g in live]
        if (fused and FUSED_BEAM_STEP and K <= 32 and all(type(p) is IndexBasedLogitsProcessor for p in procs) and can_fuse_groups(procs, logits, K)
                and max(sp["max_length"] for sp in specs) < 62 and (stepper is not None or input_ids.shape[-1] == 1)):
            # ---- ONE library call for everything between two model steps (the torch ops below are its specification) ----
            if stepper is None:
                stepper = _BeamStepper(specs, K, decoder_start_token_id, device)
                for g in range(len(specs)):
                    out_steps[g].packed = stepper.hist[g]
            cur = input_ids.shape[-1]
            tok_buf, anc = decoder.step_buffers() if hasattr(decoder, "step_buffers") else (None, None)
            n_live_after = [g for g in live if cur + 1 < specs[g]["max_length"]]
            views, next_tokens_in = stepper.step(live, logits, cur, tag, chain_next=bool(n_live_after) and cur >= 2, tokens_out=tok_buf, anc=anc)
            for g, v in zip(live, views):
                out_steps[g].append(v)
            if anc is None:
                decoder.reorder(stepper.beam_idx[stepper.row0:])
            if _DEBUG_MARK is not None:
                _DEBUG_MARK(1000 * tag + 10 * cur + 3)
            cur += 1
            input_ids = stepper.input_ids(cur)
            done = [g for g in live if cur >= specs[g]["max_length"]]
            if done:
                a = 0
                for g in done:
                    b = a + specs[g]["batch"] * K
                    finals[g] = (input_ids[a:b].contiguous(), stepper.beam_scores[stepper.row0 + a:stepper.row0 + b].clone())
                    a = b
                live = live[len(done):]
                if not live:
                    break
                nq = a // K
                stepper.drop(a)
                input_ids = stepper.input_ids(cur)
                B -= nq
                R = B * K
                decoder.narrow(nq)
                next_tokens_in = None if next_tokens_in is None else next_tokens_in[a:]
                if hasattr(decoder, "step_buffers") and decoder.step_buffers()[0] is not None:
                    # the narrower decode state has its own token buffer: the survivors' tokens move there (once per dropped group)
                    decoder.step_buffers()[0].copy_(next_tokens_in)
                    next_tokens_in = decoder.step_buffers()[0]
            continue
        if fused and all(p is not None and hasattr(p, "fused_topk") for p in procs) and can_fuse_groups(procs, logits, K):
            if len(procs) == 1:
                flat, next_scores = procs[0].fused_topk(input_ids, logits, beam_scores, B, K, parent_rows=beam_idx, tag=tag)
            else:
                flat, next_scores = fused_topk_groups(procs, batches, input_ids, logits, beam_scores, K, parent_rows=beam_idx, tag=tag)
        else:
            logp = torch.log_softmax(logits.float(), dim=-1)
            processed = _inf_nan_remove(logp)
            unconstrained = processed + beam_scores[:, None]
            if any(p is not None for p in procs):
                constrained = unconstrained.clone()
                a = 0
                for p, b in zip(procs, batches):
                    if p is not None:
                        constrained[a:a + b * K] = p(input_ids[a:a + b * K], processed[a:a + b * K]) + beam_scores[a:a + b * K, None]
                    a += b * K
            else:
                constrained = unconstrained
            _, flat = torch.topk(constrained.view(B, K * V), 2 * K, dim=1, largest=True, sorted=True)
            next_scores = unconstrained.view(B, K * V).gather(-1, flat)
        if _DEBUG_MARK is not None:
            _DEBUG_MARK(1000 * tag + 10 * input_ids.shape[-1] + 2)       # constraint + top-2K
        next_indices = flat // V                    # (next_tokens / V).long(), exact for K*V < 2^24 (309)
        next_tokens = flat % V
        src_rows = row_base + next_indices          # batch_beam_idx (661)
        prefix = input_ids[src_rows]
        a = 0
        for g in live:
            b = a + specs[g]["batch"]
            out_steps[g].append((prefix[a:b], next_tokens[a:b], next_scores[a:b]))
            a = b
        # first K non-eos candidates, in rank order, become the next beams (670-685)
        keep = next_tokens != eos_q
        order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)[:, :K]
        # (the reference's ValueError at 687-690 cannot trigger: each of the K rows holds
        # exactly one eos entry, so at most K of the 2K picks are eos)
        beam_scores = next_scores.gather(1, order).view(R)
        beam_tokens = next_tokens.gather(1, order).view(R)
        beam_idx = src_rows.gather(1, order).view(R)
        input_ids = torch.cat([input_ids[beam_idx], beam_tokens.unsqueeze(-1)], dim=-1)
        decoder.reorder(beam_idx)
        if _DEBUG_MARK is not None:
            _DEBUG_MARK(1000 * tag + 10 * (input_ids.shape[-1] - 1) + 3)  # the loop's own torch ops + the ancestry table
        cur = input_ids.shape[-1]
        done = [g for g in live if cur >= specs[g]["max_length"]]       # MaxLengthCriteria (340); a prefix of `live`
        if done:
            a = 0
            for g in done:
                b = a + specs[g]["batch"] * K
                finals[g] = (input_ids[a:b], beam_scores[a:b])
                a = b
            live = live[len(done):]
            if not live:
                break
            nq = a // K
            # the finished decodes' rows leave the loop; beam_idx keeps naming rows of the call before the cut, which is how
            # the index's kept prefix ranges are addressed by the next constraint call
            input_ids, beam_scores, beam_idx, eos_q = input_ids[a:], beam_scores[a:], beam_idx[a:], eos_q[nq:]
            B -= nq
            R = B * K
            row_base = (torch.arange(B, device=device) * K).unsqueeze(1)
            decoder.narrow(nq)
    try:
        decoder.first_logits = first_logits          # picked up by fm_index_generate(pending=True)
    except AttributeError:
        pass
    return [(out_steps[g], finals[g]) for g in range(len(specs))]


def _history_to_hypotheses(steps, final, batch_size: int, num_beams: int, length_penalty: float):
    """BeamHypothesesWithMemory.add (752-755) + the output comprehension of
    fm_index_generate (555), done on the host after the transfer.  ``add`` computes
    ``score = sum_logprobs / size**lp``, keeps the hypothesis iff ``score > -inf`` and the output
    multiplies the ``size**lp`` back: the same python-float operations here, per element, with the
    token lists built by one ``tolist`` per step instead of a list concatenation per hypothesis."""
    B, K = batch_size, num_beams
    out = [[] for _ in range(B)]
    ninf = float("-inf")

    def extend(scores, seqs, size):
        norm = size ** length_penalty                    # python pow, as .item() floats gave the reference
        if norm == 1.0:                                   # x / 1.0 * 1.0 == x exactly
            return [(s, q) for s, q in zip(scores, seqs) if s > ninf]
        return [((s / norm) * norm, q) for s, q in zip(scores, seqs) if (s / norm) > ninf]

    for prefix, tokens, scores in steps:
        size = prefix.shape[-1] + 1
        seqs = torch.cat([prefix, tokens.unsqueeze(-1)], dim=-1).tolist()      # [B][2K][size]
        sc = scores.tolist()
        for b in range(B):
            out[b] += extend(sc[b], seqs[b], size)
    ids, fscores = final
    size = ids.shape[-1]
    ids, fscores = ids.tolist(), fscores.tolist()
    for b in range(B):
        out[b] += extend(fscores[b * K:(b + 1) * K], ids[b * K:(b + 1) * K], size)
    return out


@torch.no_grad()
def fm_index_generate(
        model,
        index: FMIndex,
        input_ids: torch.LongTensor,
        attention_mask: torch.LongTensor,
        min_length: int = 3,
        max_length: int = 25,
        length_penalty: float = 1.0,
        num_beams: int = 3,
        diverse_bs_groups: int = 1,
        diverse_bs_penalty: float = 0.0,
        eos_token_id: Optional[int] = None,
        force_decoding_from: Optional[List[int]] = None,
        always_allow_eos: bool = False,
        keep_history: bool = False,
        disable_fm_index: bool = False,
        sample: bool = False,
        stop_at_count: int = 0,
        topk: int = 0,
        transformers_output: bool = False,
        **kwargs,
):
    """Drop-in for ``seal.beam_search.fm_index_generate`` (beam_search.py:391-557)
    on the configuration the SEAL searcher uses (retrieval.py:70-83,162-176):
    ``keep_history=True``, one beam group, no sampling, no top-k warper.
    Returns ``List[List[(score, token_list)]]``, one list per query.

    ``kwargs``: ``forced_bos_token_id`` as in the reference; ``decoder`` (a
    ``BartStepDecoder``-like object) and ``constrained_decoding_processor`` to
    inject pre-built pieces.
    """
    if diverse_bs_groups != 1 or sample or topk or transformers_output or not keep_history:
        raise NotImplementedError(
            "seal_amd.fm_index_generate implements the SEAL search path: keep_history=True, diverse_bs_groups=1, "
            "sample=False, topk=0 (reference retrieval.py:70-83); other modes of the reference are out of scope")
    from .bart_decoder import BartStepDecoder

    forced_bos_token_id = kwargs.pop("forced_bos_token_id", getattr(model.config, "forced_bos_token_id", None))
    decoder = kwargs.pop("decoder", None)
    if decoder is None:   # fused projection weights are built once per model
        decoder = getattr(model, "_seal_step_decoder", None)
        if decoder is None or decoder.lm_w.device != input_ids.device:
            decoder = BartStepDecoder(model)
            model._seal_step_decoder = decoder
    processor = kwargs.pop("constrained_decoding_processor", None)
    logit_bias = kwargs.pop("logit_bias", None)     # extension: [batch, vocab] additive bias on the next-token logits
    if eos_token_id is None:
        eos_token_id = model.config.eos_token_id
    if processor is None and not disable_fm_index:
        processor = IndexBasedLogitsProcessor(
            num_beams=num_beams, index=index, pad_token_id=model.config.pad_token_id,
            eos_token_id=eos_token_id or model.config.eos_token_id, force_decoding_from=force_decoding_from,
            stop_at_count=stop_at_count, always_allow_eos=always_allow_eos, forced_bos_token_id=forced_bos_token_id)
    if disable_fm_index:
        processor = None
    pending = kwargs.pop("pending", False)
    enc = decoder.encode(input_ids, attention_mask)
    decoder.start(enc, attention_mask, num_beams, max_length)
    saved_bias = decoder.logit_bias
    if logit_bias is not None:
        decoder.logit_bias = logit_bias
    try:
        steps, final = constrained_beam_search(
            decoder, input_ids.shape[0], num_beams, max_length, model.config.decoder_start_token_id, eos_token_id,
            constrained_decoding_processor=processor, device=input_ids.device)
    finally:
        decoder.logit_bias = saved_bias
    if pending:
        # the whole decode is enqueued by now and nothing has waited for the GPU: hand back a handle, so that the caller
        # can put more work behind it (the next decode) before it asks for the hypotheses
        pg = PendingGenerate(steps, final, input_ids.shape[0], num_beams, length_penalty, enc=enc, attention_mask=attention_mask)
        pg.first_logits = getattr(decoder, "first_logits", None)
        return pg
    return _history_to_hypotheses(steps, final, input_ids.shape[0], num_beams, length_penalty)


@torch.no_grad()
def fm_index_generate_joint(model, index: FMIndex, input_ids: torch.LongTensor, attention_mask: torch.LongTensor, jobs,
                            num_beams: int = 3, length_penalty: float = 1.0, stop_at_count: int = 0, always_allow_eos: bool = False,
                            disable_fm_index: bool = False, logit_bias=None, decoder=None, forced_bos_token_id=None, extra_inputs=None):
    """Several ``fm_index_generate(keep_history=True)`` calls of ONE model as one decode loop: ``jobs`` = dicts with
    ``batch`` (the next ``batch`` rows of ``input_ids`` are this job's encoder inputs), ``max_length``, ``eos_token_id``,
    ``force_decoding_from``, optionally its own ``stop_at_count`` (default: the call's).  The searcher's body and title decodes (reference retrieval.py:70-83, 162-176: two
    ``generate`` calls one after the other over the same queries) become 2 x batch x beams rows per model step: one encoder
    pass, GEMMs at twice the height, one constraint launch per step for both (``constrained_beam_search_groups``).  Every job
    gets the hypotheses its own call would produce.  ``logit_bias`` [sum of batches, vocab].  Returns one ``PendingGenerate``
    per job, in the order given (nothing has waited for the GPU).  ``extra_inputs`` = (input_ids, attention_mask) of further encoder
    inputs of the same width that ride along in the ONE encoder pass (the searcher's bare queries, which its rescoring of the body
    keys encodes, reference retrieval.py:93-100); their states come back as ``out[0].extra_encoded``."""
    from .bart_decoder import BartStepDecoder
    if forced_bos_token_id is None:
        forced_bos_token_id = getattr(model.config, "forced_bos_token_id", None)
    if decoder is None:
        decoder = getattr(model, "_seal_step_decoder", None)
        if decoder is None or decoder.lm_w.device != input_ids.device:
            decoder = BartStepDecoder(model)
            model._seal_step_decoder = decoder
    assert sum(j["batch"] for j in jobs) == input_ids.shape[0]
    # rows in ascending max_length: the decodes that end first sit in front and are cut off when they end
    order = sorted(range(len(jobs)), key=lambda i: jobs[i]["max_length"])
    starts = [sum(j["batch"] for j in jobs[:i]) for i in range(len(jobs))]
    if order != list(range(len(jobs))):
        perm = torch.cat([torch.arange(starts[i], starts[i] + jobs[i]["batch"], device=input_ids.device) for i in order])
        input_ids, attention_mask = input_ids[perm], attention_mask[perm]
        if logit_bias is not None:
            logit_bias = logit_bias[perm]
    specs = []
    for i in order:
        j = jobs[i]
        eos = j.get("eos_token_id")
        if eos is None:
            eos = model.config.eos_token_id
        proc = None if disable_fm_index else IndexBasedLogitsProcessor(
            num_beams=num_beams, index=index, pad_token_id=model.config.pad_token_id, eos_token_id=eos,
            force_decoding_from=j.get("force_decoding_from"), stop_at_count=j.get("stop_at_count", stop_at_count), always_allow_eos=always_allow_eos,
            forced_bos_token_id=forced_bos_token_id)
        specs.append(dict(batch=j["batch"], max_length=j["max_length"], eos_token_id=eos, processor=proc))
    extra_encoded = None
    if extra_inputs is not None and extra_inputs[0].shape[1] == input_ids.shape[1]:
        n_own = input_ids.shape[0]
        enc_all = decoder.encode(torch.cat([input_ids, extra_inputs[0]]), torch.cat([attention_mask, extra_inputs[1]]))
        enc, extra_encoded = enc_all[:n_own], (enc_all[n_own:], extra_inputs[1])
    else:
        enc = decoder.encode(input_ids, attention_mask)
    cuts, gone = [], 0                       # queries that have left the loop after each group but the last
    for sp in specs[:-1]:
        gone += sp["batch"]
        cuts.append(gone)
    decoder.start(enc, attention_mask, num_beams, specs[-1]["max_length"], narrow_plan=cuts)
    saved_bias = decoder.logit_bias
    if logit_bias is not None:
        decoder.logit_bias = logit_bias
    try:
        results = constrained_beam_search_groups(decoder, specs, num_beams, model.config.decoder_start_token_id, device=input_ids.device)
    finally:
        decoder.logit_bias = saved_bias
    first_logits = getattr(decoder, "first_logits", None)
    out = [None] * len(jobs)
    a = 0
    for i, sp, (steps, final) in zip(order, specs, results):
        b = a + sp["batch"]
        pg = PendingGenerate(steps, final, sp["batch"], num_beams, length_penalty, enc=enc[a:b], attention_mask=attention_mask[a:b])
        pg.first_logits = first_logits[a:b] if first_logits is not None else None
        out[i] = pg
        a = b
    out[0].extra_encoded = extra_encoded
    return out


class PendingGenerate:
    """an enqueued ``fm_index_generate``: ``result()`` waits for the GPU and builds the hypothesis lists.
    ``enc`` = the encoder states of the call (the searcher reuses them where the reference re-encodes the same input).
    ``arrays()`` is the same history as arrays (what the searcher filters before any python list exists)."""

    def __init__(self, steps, final, batch, beams, length_penalty, enc=None, attention_mask=None):
        self._args = (steps, final, batch, beams, length_penalty)
        self.enc, self.attention_mask = enc, attention_mask
        self.first_logits = None          # [batch, vocab]: next-token logits of the first decoder position (logit bias included)
        self.extra_encoded = None         # fm_index_generate_joint(extra_inputs=...): (encoder states, attention mask) of the inputs that rode along
        self._packed = None
        self._pack(steps, final, batch, beams)
        self._event = torch.cuda.Event() if final[0].is_cuda else None
        if self._event is not None:
            self._event.record(torch.cuda.current_stream(final[0].device))

    def _pack(self, steps, final, batch, beams):
        """the whole history in two tensors, copied to pinned host memory behind the decode (nothing waits): tokens
        [batch, H, L] (-1 beyond a hypothesis' length; H = steps x 2K + K hypotheses per query in the reference's recording
        order, beam_search.py:658-668 then 717-725) and their summed log-probs [batch, H]"""
        B, K = batch, beams
        dev = final[0].device
        L = final[0].shape[-1]
        H = sum(t.shape[1] for _, t, _ in steps) + K
        packed = getattr(steps, "packed", None)
        if packed is not None and tuple(packed[0].shape) == (B, H, L):
            tok, sc = packed                      # written in place, step by step, by fmi_dev_beam_step: only the final beams are missing
            a = H - K
        else:
            tok = torch.full((B, H, L), -1, dtype=torch.int64, device=dev)
            sc = torch.empty(B, H, dtype=torch.float32, device=dev)
            a = 0
            for prefix, tokens, scores in steps:
                n, t = tokens.shape[1], prefix.shape[-1]
                tok[:, a:a + n, :t] = prefix
                tok[:, a:a + n, t] = tokens
                sc[:, a:a + n] = scores
                a += n
        tok[:, a:a + K, :] = final[0].view(B, K, L)
        sc[:, a:a + K] = final[1].view(B, K)
        if dev.type == "cuda":
            host_tok = torch.empty(tok.shape, dtype=tok.dtype, pin_memory=True)
            host_sc = torch.empty(sc.shape, dtype=sc.dtype, pin_memory=True)
            host_tok.copy_(tok, non_blocking=True)
            host_sc.copy_(sc, non_blocking=True)
            from . import split_gemm
            self._split_flag = split_gemm.flag_snapshot(dev)          # activations beyond fp16's range in this decode's split GEMMs?
        else:
            host_tok, host_sc = tok, sc
        lens = [p.shape[-1] + 1 for p, _, _ in steps for _ in range(p.shape[1])] + [L] * K
        self._packed = (host_tok, host_sc, lens)

    def arrays(self):
        """(tok [B, H, L] int64, length [H] int64, score [B, H] float64, valid [B, H] bool) as numpy: hypothesis h of query b
        is ``tok[b, h, :length[h]]`` with the score ``fm_index_generate`` would return for it (``BeamHypothesesWithMemory.add``
        divides by ``size ** length_penalty``, the output comprehension multiplies it back, reference beam_search.py:555,752-755);
        ``valid`` = kept by ``add`` (score > -inf)."""
        import numpy as np
        self._wait()
        host_tok, host_sc, lens = self._packed
        lp = self._args[4] if self._args is not None else self._lp
        tok = host_tok.numpy()
        length = np.asarray(lens, dtype=np.int64)
        s = host_sc.numpy().astype(np.float64)                      # what .tolist() / .item() hand the reference: the fp32 value as a double
        norm = np.asarray([float(n) ** lp for n in lens], dtype=np.float64)[None, :]
        with np.errstate(invalid="ignore"):
            normed = s / norm
            score = np.where(norm == 1.0, s, normed * norm)
        return tok, length, score, normed > float("-inf")

    def _wait(self):
        if self._event is not None:
            self._event.synchronize()
            flag, self._split_flag = getattr(self, "_split_flag", None), None
            if flag is not None:
                from . import split_gemm
                split_gemm.check_snapshot(flag, self.enc.device if self.enc is not None else None)

    def result(self):
        self._wait()
        out = _history_to_hypotheses(*self._args)
        self._lp = self._args[4]
        self._args = None
        return out
