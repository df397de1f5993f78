"""TEST INFRASTRUCTURE ONLY -- plain restatements of the reference's decode-side
Python semantics, run on the CPU against an ``OracleFMIndex``.  PINNED against the reference's
own code: tests/golden/ref_index_and_mask.json (IndexBasedLogitsProcessor.__call__) and
ref_beam_search.json (the whole fm_index_generate with keep_history) were produced by running
seal/beam_search.py itself (tests/golden/make_reference_golden.py); tests/test_reference_golden.py
checks both functions below against them.

* ``oracle_logits_mask``: IndexBasedLogitsProcessor.__call__
  (reference seal/beam_search.py:62-140), returning the boolean "allowed" matrix.
"""
from typing import List, Optional

import numpy as np


def oracle_logits_mask(index, input_ids: List[List[int]], vocab: int, num_beams: int, pad_token_id: int = 0,
                       eos_token_id: int = 2, force_decoding_from: Optional[List[int]] = None,
                       stop_at_count: int = 0, always_allow_eos: bool = False,
                       forced_bos_token_id: Optional[int] = None) -> np.ndarray:
    rows = len(input_ids)
    allowed = np.zeros((rows, vocab), dtype=bool)
    ids = [list(r) for r in input_ids]
    if forced_bos_token_id is not None:             # beam_search.py:66-71
        if len(ids[0]) == 1:
            allowed[:, forced_bos_token_id] = True
            return allowed
        ids = [r[1:] for r in ids]
    if len(ids[0]) == 1:                            # beam_search.py:73-77
        allowed[:, index.occurring_distinct] = True
    else:
        lows, highs, counts = [], [], []
        for sent in ids:                            # beam_search.py:87-105
            if sent[-1] in (eos_token_id, pad_token_id):
                low = high = count = 0
            elif force_decoding_from is not None:
                low, high = index.get_range(force_decoding_from + sent[1:])
                count = index.get_count(force_decoding_from + sent[1:-1])
            else:
                low, high = index.get_range(sent[1:])
                count = index.get_count(sent[1:-1])
            lows.append(low); highs.append(high); counts.append(count)
        results = index.get_distinct_count_multi(lows, highs)   # beam_search.py:107
        for r, sent in enumerate(ids):              # beam_search.py:111-135
            if stop_at_count > 0 and counts[r] <= stop_at_count:
                distinct = [eos_token_id]
            elif sent[-1] == eos_token_id:
                distinct = [pad_token_id]
            elif sent[-1] == pad_token_id:
                distinct = [pad_token_id]
            else:
                distinct, _ = results[r]
            allowed[r, distinct] = True
    if always_allow_eos:                            # beam_search.py:137-138
        allowed[:, eos_token_id] = True
    return allowed


# ---------------------------------------------------------------------------
# constrained_beam_search + BeamSearchScorerWithMemory (reference
# seal/beam_search.py:143-389, 559-758), restated with python loops and
# .item() exactly where the reference has them.  `logits_fn(input_ids)` returns
# next-token logits [rows, vocab] for the decoder prefixes `input_ids`.
# ---------------------------------------------------------------------------
def oracle_fm_index_generate(logits_fn, index, batch_size: int, num_beams: int, max_length: int, vocab: int,
                             decoder_start_token_id: int = 2, pad_token_id: int = 1, eos_token_id: int = 2,
                             length_penalty: float = 1.0, force_decoding_from=None, stop_at_count: int = 0,
                             always_allow_eos: bool = False, forced_bos_token_id=None, disable_fm_index: bool = False):
    import torch

    hyps = [[] for _ in range(batch_size)]                      # BeamHypothesesWithMemory.beams

    def hyp_add(b, tokens, sum_logprobs):                       # beam_search.py:752-755
        size = len(tokens)
        hyps[b].append((sum_logprobs / (size ** length_penalty), list(tokens)))

    input_ids = torch.full((batch_size * num_beams, 1), decoder_start_token_id, dtype=torch.long)
    beam_scores = torch.zeros((batch_size, num_beams), dtype=torch.float)     # 214-216
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    while True:
        logits = logits_fn(input_ids).float().cpu()
        next_token_scores = torch.nn.functional.log_softmax(logits, dim=-1)   # 251
        # logits_processor == [InfNanRemoveLogitsProcessor] (430-445 with eos_token_id=None)
        proc = next_token_scores.clone()
        proc[proc != proc] = 0.0
        proc[proc == float("inf")] = torch.finfo(proc.dtype).max
        unconstrained = proc + beam_scores[:, None]                            # 258
        if disable_fm_index:
            constrained = unconstrained
        else:
            allowed = oracle_logits_mask(index, input_ids.tolist(), vocab, num_beams, pad_token_id=pad_token_id,
                                         eos_token_id=eos_token_id, force_decoding_from=force_decoding_from,
                                         stop_at_count=stop_at_count, always_allow_eos=always_allow_eos,
                                         forced_bos_token_id=forced_bos_token_id)
            mask = torch.full_like(proc, float("-inf"))
            mask[torch.from_numpy(allowed)] = 0.0
            constrained = (proc + mask) + beam_scores[:, None]                 # 261-262
        flat_c = constrained.view(batch_size, num_beams * vocab)
        _, next_tokens = torch.topk(flat_c, 2 * num_beams, dim=1, largest=True, sorted=True)   # 304-306
        next_scores = unconstrained.view(batch_size, num_beams * vocab).gather(-1, next_tokens)  # 307
        next_indices = (next_tokens / vocab).long()                            # 309
        next_tokens = next_tokens % vocab                                      # 310
        # BeamSearchScorerWithMemory.process, 614-703
        nb_scores = torch.zeros((batch_size, num_beams))
        nb_tokens = torch.zeros((batch_size, num_beams), dtype=torch.long)
        nb_indices = torch.zeros((batch_size, num_beams), dtype=torch.long)
        for b in range(batch_size):
            beam_idx = 0
            broken = False
            for tok, sc, idx in zip(next_tokens[b], next_scores[b], next_indices[b]):
                batch_beam_idx = b * num_beams + idx.item()
                hyp_add(b, input_ids[batch_beam_idx].tolist() + [tok.item()], sc.item())   # 662-668
                if broken:
                    pass
                elif tok.item() == eos_token_id:
                    pass
                else:
                    nb_scores[b, beam_idx] = sc
                    nb_tokens[b, beam_idx] = tok
                    nb_indices[b, beam_idx] = batch_beam_idx
                    beam_idx += 1
                if beam_idx == num_beams:
                    broken = True
            assert beam_idx == num_beams
        beam_scores = nb_scores.view(-1)
        bidx = nb_indices.view(-1)
        input_ids = torch.cat([input_ids[bidx, :], nb_tokens.view(-1).unsqueeze(-1)], dim=-1)   # 326
        if input_ids.shape[-1] >= max_length:                                   # 340 (MaxLengthCriteria)
            break
    for b in range(batch_size):                                                # finalize, 717-725
        for j in range(num_beams):
            r = b * num_beams + j
            hyp_add(b, input_ids[r].tolist(), beam_scores[r].item())
    # fm_index_generate's return comprehension, 555
    return [[(h[0] * len(h[1]) ** length_penalty, h[1]) for h in hh if h[0] > float("-inf")] for hh in hyps]
