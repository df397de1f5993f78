"""TEST INFRASTRUCTURE ONLY -- plain restatements of the reference's decode-side
Python semantics, run on the CPU against an ``OracleFMIndex``.

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
