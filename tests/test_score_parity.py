"""The float half of parity: every hypothesis score the beam loop records == the running fp32 sum of HF's own
cache-free forward along the hypothesis (oracle/hf_scores.py; reference beam_search.py:231-253,302-307), and the
prefix-tree rescoring == one HF row per key (reference keys.py:64-141).  On CPU with the tiny model (the checker
itself), on the GPU at BART-LARGE geometry (``BartConfig()``: d_model 1024, 16 heads, 12 layers, vocabulary 50 265)
-- the model bench.py runs -- where ``tests/test_gpu_decode.py`` only reaches 2 layers."""
import numpy as np
import pytest
import torch


def _history(model, index, enc_ids, enc_mask, proc=None, bias=None, **kw):
    from seal_amd.beam_search import fm_index_generate
    pg = fm_index_generate(model, index, enc_ids, enc_mask, min_length=1, keep_history=True, pending=True, logit_bias=bias,
                           **({"constrained_decoding_processor": proc} if proc is not None else {}), **kw)
    steps, final, B, K, _ = pg._args
    return steps, final, B, K, pg


@pytest.mark.parametrize("kw", [dict(max_length=6, num_beams=3, length_penalty=0.0),
                                dict(max_length=7, num_beams=4, length_penalty=0.0, force_decoding_from=[2], eos_token_id=7)])
def test_recorded_beam_scores_equal_hf_teacher_forced_sums_on_cpu(kw):
    from oracle.hf_scores import compare_beam_history
    from oracle.seal_oracle import OracleFMIndex
    from tests.helpers import OracleLogitsProcessor, make_docs, tiny_bart
    vocab = 120
    m = tiny_bart(vocab)
    docs = make_docs(3, 150, vocab, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    torch.manual_seed(2)
    enc_ids = torch.randint(4, vocab, (4, 8))
    enc_mask = torch.ones_like(enc_ids)
    enc_mask[1, 5:] = 0
    enc_ids[1, 5:] = 1
    proc = OracleLogitsProcessor(orc, kw["num_beams"], vocab, pad_token_id=1, eos_token_id=kw.get("eos_token_id", 2),
                                 force_decoding_from=kw.get("force_decoding_from"))
    bias = torch.randn(4, vocab)
    steps, final, B, K, _ = _history(m, None, enc_ids, enc_mask, proc, bias, **kw)
    rep = compare_beam_history(m, enc_ids, enc_mask, steps, final, B, K, logit_bias=bias)
    assert rep["violations"] == 0 and rep["values"] > 50 and rep["max_abs_err"] <= 1e-4, rep
    # the checker must notice a score that is off by more than the tolerance
    steps[2][2][0, 0] += 3e-4
    rep = compare_beam_history(m, enc_ids, enc_mask, steps, final, B, K, logit_bias=bias)
    assert rep["violations"] == 1


@pytest.mark.gpu
def test_beam_and_rescoring_scores_at_bart_large_geometry_match_hf_fp32_forward():
    """2 queries x beam 15 through the fused step decoder (hipGraph, sealnn_* kernels, hipBLASLt fp32 GEMMs) and the HIP
    constraint at BART-large size; body decode (10 tokens) and title decode (15, forced first token): every recorded
    score within 1e-4 of HF's own forward (north_star), then the prefix-tree rescoring of the body keys against one HF
    row per key."""
    import bench
    from oracle.hf_scores import compare_beam_history, compare_rescoring
    from seal_amd import FMIndex
    from transformers import BartConfig, BartForConditionalGeneration
    dev = torch.device("cuda:0")
    data, beg, title_len, ids_by_rank = bench.synth_corpus(3000, dev, seed=0, phrases=2000)
    index = FMIndex()
    index.initialize_from_device(data, beg.tolist())
    queries, bias = bench.synth_queries(2, data, beg, title_len, ids_by_rank, dev, seed=3)
    torch.manual_seed(0)
    cfg = BartConfig()
    cfg.forced_bos_token_id = None
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg)
    model.eval()
    with torch.no_grad():
        for tok in (cfg.pad_token_id, cfg.bos_token_id, bench.VOCAB - 1):
            model.final_logits_bias[0, tok] = float("-inf")
    from seal_amd.keys import _pad_batch
    from seal_amd.beam_search import fm_index_generate_joint
    # the searcher's default: both decodes as ONE loop of 2 x batch x beams rows
    toks2 = [q[:-1] + m + [45056, 2055] + q[-1:] for m in ([45056, 809], [45056, 1270]) for q in queries]
    ids2 = _pad_batch(toks2, cfg.pad_token_id, dev)
    mask2 = (ids2 != cfg.pad_token_id).long()
    pend = fm_index_generate_joint(model, index, ids2, mask2, [dict(batch=2, max_length=10), dict(batch=2, max_length=15, force_decoding_from=[2],
                                                                                                 eos_token_id=bench.TITLE_EOS)],
                                   num_beams=15, length_penalty=0.0, logit_bias=torch.cat([bias, bias]))
    assert model._seal_step_decoder._st.fused is True and model._seal_step_decoder._st.shape[0] == 2
    for i, pg in enumerate(pend):
        steps, final, B, K, _ = pg._args
        rep = compare_beam_history(model, ids2[2 * i:2 * i + 2], mask2[2 * i:2 * i + 2], steps, final, B, K, logit_bias=bias)
        assert rep["violations"] == 0 and rep["max_abs_err"] <= 1e-4 and rep["values"] >= (9, 14)[i] * B * K, rep
    for marker, kw in (([45056, 809], dict(max_length=10, num_beams=15, length_penalty=0.0)),
                       ([45056, 1270], dict(max_length=15, num_beams=15, length_penalty=0.0, force_decoding_from=[2],
                                            eos_token_id=bench.TITLE_EOS))):
        toks = [q[:-1] + marker + [45056, 2055] + q[-1:] for q in queries]
        enc_ids = _pad_batch(toks, cfg.pad_token_id, dev)
        enc_mask = (enc_ids != cfg.pad_token_id).long()
        steps, final, B, K, pg = _history(model, index, enc_ids, enc_mask, None, bias, **kw)
        assert model._seal_step_decoder._st.fused is True            # the fused kernels, not the torch fallback
        rep = compare_beam_history(model, enc_ids, enc_mask, steps, final, B, K, logit_bias=bias)
        assert rep["violations"] == 0 and rep["max_abs_err"] <= 1e-4, rep
        assert rep["values"] >= (kw["max_length"] - 1) * B * K, rep    # at least the live beams of every step were compared
        if "force_decoding_from" not in kw:
            hyps = pg.result()
            keys = [[(s, k[1:]) for s, k in h if len(k) > 1][:120] for h in hyps]
            r2 = compare_rescoring(model, toks, keys, length_penalty=0.0, logit_bias=bias, strip_from_bos=[2, 49314],
                                   strip_from_eos=[49314, 45056, 2])
            assert r2["violations"] == 0 and r2["values"] == sum(len(k) for k in keys), r2
