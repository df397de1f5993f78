"""bf16 storage of the fused decoder kernels (include/sealnn.h ``*_bf16``; BASELINE.json configs[4] asks for a bf16 BART decode):
the kernels are the fp32 kernels instantiated for 2-byte elements -- loads widen, all arithmetic is fp32, stores round to
nearest even -- so on bf16-representable inputs every one of them must return EXACTLY the fp32 kernel's result rounded to
bf16; and the whole bf16 step decoder must be as close to HF's fp32 forward as HF's own bf16 forward is."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def test_bf16_kernels_equal_the_fp32_kernels_rounded_to_bf16():
    from seal_amd._lib import check, lib
    L = lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    bf = torch.bfloat16

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(bf).to(dev)           # bf16-representable values
    st = _st(dev)
    # add + LayerNorm
    rows, d = 37, 1024
    x, y, gam, bet = rnd(rows, d), rnd(rows, d), rnd(d), rnd(d)
    o16 = torch.empty(rows, d, dtype=bf, device=dev)
    o32 = torch.empty(rows, d, device=dev)
    check(L.sealnn_add_layernorm_bf16(st, x.data_ptr(), y.data_ptr(), gam.data_ptr(), bet.data_ptr(), rows, d, 1e-5, o16.data_ptr()))
    x32, y32, gam32, bet32 = x.float(), y.float(), gam.float(), bet.float()          # (named: a temporary's memory is reused at once)
    check(L.sealnn_add_layernorm(st, x32.data_ptr(), y32.data_ptr(), gam32.data_ptr(), bet32.data_ptr(), rows, d, 1e-5, o32.data_ptr()))
    assert torch.equal(o16, o32.to(bf))
    # self-attention step with the ancestry-addressed cache, three positions deep
    R, H, T = 10, 3, 9
    kc16, vc16 = torch.zeros(R, H, T, 64, dtype=bf, device=dev), torch.zeros(R, H, T, 64, dtype=bf, device=dev)
    kc32, vc32 = torch.zeros(R, H, T, 64, device=dev), torch.zeros(R, H, T, 64, device=dev)
    anc16 = torch.arange(R, dtype=torch.int32, device=dev).repeat(T, 1).contiguous()
    anc32 = anc16.clone()
    t = torch.zeros(1, dtype=torch.long, device=dev)
    for step in range(4):
        qkv = rnd(R, 3 * H * 64)
        a16, a32 = torch.empty(R, H * 64, dtype=bf, device=dev), torch.empty(R, H * 64, device=dev)
        check(L.sealnn_self_attn_step_bf16(st, qkv.data_ptr(), kc16.data_ptr(), vc16.data_ptr(), t.data_ptr(), R, H, T, 0.125, a16.data_ptr(), anc16.data_ptr()))
        qkv32 = qkv.float()
        check(L.sealnn_self_attn_step(st, qkv32.data_ptr(), kc32.data_ptr(), vc32.data_ptr(), t.data_ptr(), R, H, T, 0.125, a32.data_ptr(), anc32.data_ptr()))
        assert torch.equal(a16, a32.to(bf)), step
        perm = torch.randint(0, R, (R,), generator=g).to(dev)
        anc16.copy_(anc16.index_select(1, perm))
        anc32.copy_(anc32.index_select(1, perm))
        t.add_(1)
    assert torch.equal(kc16, kc32.to(bf)) and torch.equal(anc16, anc32)
    # cross-attention: per step (beams of a query share K/V), arbitrary rows, runs
    B, K, S = 3, 5, 23
    q, ck, cv = rnd(B * K, H * 64), rnd(B, H, 64, S), rnd(B, H, S, 64)
    bias = torch.zeros(B, S)
    bias[1, 17:] = torch.finfo(bf).min
    bias = bias.to(bf).to(dev)
    c16, c32 = torch.empty(B * K, H * 64, dtype=bf, device=dev), torch.empty(B * K, H * 64, device=dev)
    check(L.sealnn_cross_attn_step_bf16(st, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), B, K, H, S, 0.125, c16.data_ptr()))
    q32, ck32, cv32, bias32 = q.float(), ck.float(), cv.float(), bias.float()
    check(L.sealnn_cross_attn_step(st, q32.data_ptr(), ck32.data_ptr(), cv32.data_ptr(), bias32.data_ptr(), B, K, H, S, 0.125, c32.data_ptr()))
    assert torch.equal(c16, c32.to(bf))
    rb = torch.arange(B, dtype=torch.int32).repeat_interleave(K).to(dev)
    r16, r32 = torch.empty_like(c16), torch.empty_like(c32)
    check(L.sealnn_cross_attn_rows_bf16(st, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), rb.data_ptr(), B * K, H, S, 0.125, r16.data_ptr()))
    check(L.sealnn_cross_attn_rows(st, q32.data_ptr(), ck32.data_ptr(), cv32.data_ptr(), bias32.data_ptr(), rb.data_ptr(), B * K, H, S, 0.125, r32.data_ptr()))
    assert torch.equal(r16, r32.to(bf))
    u16 = torch.empty_like(c16)
    check(L.sealnn_cross_attn_runs_bf16(st, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), rb.data_ptr(), B * K, K, H, S, 0.125, u16.data_ptr()))
    assert torch.equal(u16, r16)
    # causal rows and the prefix tree on chains
    n_seq, Tq = 6, 7
    qkv = rnd(n_seq * Tq, 3 * H * 64)
    s16, s32 = torch.empty(n_seq * Tq, H * 64, dtype=bf, device=dev), torch.empty(n_seq * Tq, H * 64, device=dev)
    check(L.sealnn_causal_self_attn_bf16(st, qkv.data_ptr(), n_seq, Tq, H, 0.125, s16.data_ptr()))
    qkv32 = qkv.float()
    check(L.sealnn_causal_self_attn(st, qkv32.data_ptr(), n_seq, Tq, H, 0.125, s32.data_ptr()))
    assert torch.equal(s16, s32.to(bf))
    anc = torch.full((n_seq * Tq, Tq), -1, dtype=torch.int32)
    for n in range(n_seq):
        for j in range(Tq):
            anc[n * Tq + j, :j + 1] = torch.arange(n * Tq, n * Tq + j + 1, dtype=torch.int32)
    anc = anc.to(dev)
    t16 = torch.empty_like(s16)
    check(L.sealnn_tree_self_attn_bf16(st, qkv.data_ptr(), anc.data_ptr(), n_seq * Tq, Tq, H, 0.125, t16.data_ptr()))
    assert torch.equal(t16, s16)


def test_bf16_step_decoder_is_as_close_to_the_fp32_forward_as_hf_bf16_is():
    """the graph-captured fused step decoder on a bf16 model (fused path asserted) over a whole decode with beam re-ranking:
    its logits against HF's fp32 forward of the same (bf16-valued) weights, next to what HF's own bf16 forward makes of them"""
    import copy
    from seal_amd.bart_decoder import BartStepDecoder
    from tests.helpers import tiny_bart
    dev = torch.device("cuda:0")
    vocab, B, K, T, S = 120, 3, 5, 12, 19
    m16 = tiny_bart(vocab, d_model=128, heads=2, max_positions=64).to(dev).to(torch.bfloat16)
    m32 = copy.deepcopy(m16).float()
    g = torch.Generator(device="cpu").manual_seed(4)
    enc_ids = torch.randint(4, vocab, (B, S), generator=g).to(dev)
    enc_mask = torch.ones_like(enc_ids)
    enc_mask[1, S // 2:] = 0
    enc_ids[1, S // 2:] = 1
    dec = BartStepDecoder(m16)
    enc = dec.encode(enc_ids, enc_mask)
    assert enc.dtype == torch.bfloat16
    dec.start(enc, enc_mask, K, T)
    rows = torch.full((B * K, 1), 2, dtype=torch.long, device=dev)
    ids_rep, am_rep = enc_ids.repeat_interleave(K, 0), enc_mask.repeat_interleave(K, 0)
    worst_ours = worst_hf = 0.0
    for t in range(T - 1):
        got = dec.step(rows[:, -1])
        assert dec._st.fused is True and got.dtype == torch.float32
        with torch.no_grad():
            ref = m32(input_ids=ids_rep, attention_mask=am_rep, decoder_input_ids=rows).logits[:, -1, :]
            hf16 = m16(input_ids=ids_rep, attention_mask=am_rep, decoder_input_ids=rows).logits[:, -1, :].float()
        fin = torch.isfinite(ref)
        assert torch.equal(fin, torch.isfinite(got))
        worst_ours = max(worst_ours, (got[fin] - ref[fin]).abs().max().item())
        worst_hf = max(worst_hf, (hf16[fin] - ref[fin]).abs().max().item())
        nxt = torch.randint(4, vocab, (B * K,), generator=g).to(dev)
        perm = (torch.arange(B * K).view(B, K).gather(1, torch.randint(0, K, (B, K), generator=g))).reshape(-1).to(dev)
        rows = torch.cat([rows[perm], nxt[:, None]], 1)
        dec.reorder(perm)
    assert worst_hf > 0 and worst_ours <= 1.5 * worst_hf + 1e-3, (worst_ours, worst_hf)


def test_bf16_searcher_runs_the_fused_paths_and_its_scores_track_hf_bf16(monkeypatch):
    """complete search on a bf16 model: joint decode, prefix-tree rescoring and unigram scores through the bf16 kernels (no
    fallback), hypothesis scores within 0.1 of HF's bf16 forward teacher-forced along the same histories"""
    import numpy as np
    from oracle.hf_scores import compare_beam_history
    from seal_amd import FMIndex, retrieval
    from seal_amd.beam_search import fm_index_generate_joint
    from seal_amd.retrieval import SEALSearcher
    from tests.helpers import make_docs, tiny_bart
    vocab, title_eos = 120, 7
    dev = torch.device("cuda:0")
    docs = make_docs(5, 300, vocab - 8, min_len=6, max_len=18, title_sep=title_eos)
    ix = FMIndex()
    ix.initialize(docs)
    monkeypatch.setattr(retrieval, "TITLE_MAX_LENGTH", 8)
    m = tiny_bart(vocab, d_model=128, heads=2).to(dev).to(torch.bfloat16)
    s = SEALSearcher(ix, None, m, backbone="bart-tiny", length=6, beam=4, batch_size=3, add_query_to_keys=True, detokenize=False,
                     title_eos_token_id=title_eos, code_eos_token_id=vocab - 6, code_bos_token_id=title_eos,
                     marker_token_ids={"body": [vocab - 2, vocab - 3], "title": [vocab - 2, vocab - 4], "+": [vocab - 2, vocab - 5]})
    rng = np.random.default_rng(0)
    queries = [[0] + rng.integers(4, vocab - 8, size=int(rng.integers(4, 9))).tolist() + [2] for _ in range(5)]
    from seal_amd.bart_decoder import BartStepDecoder
    tree_calls = []
    real_tree, real_graph = BartStepDecoder.tree_logits, BartStepDecoder.tree_hidden_graph
    monkeypatch.setattr(BartStepDecoder, "tree_logits", lambda self, *a: (tree_calls.append(a[6] is not None), real_tree(self, *a))[1])
    monkeypatch.setattr(BartStepDecoder, "tree_hidden_graph", lambda self, *a: (lambda out: (tree_calls.append(out is not None), out)[1])(real_graph(self, *a)))
    res = s.batch_search(queries, k=10)
    assert all(len(r) > 0 for r in res) and m._seal_step_decoder._st.fused is True and tree_calls and all(tree_calls)
    from seal_amd.keys import _pad_batch
    toks = [q[:-1] + mk + [vocab - 2, vocab - 5] + q[-1:] for mk in ([vocab - 2, vocab - 3], [vocab - 2, vocab - 4]) for q in queries[:3]]
    ids = _pad_batch(toks, 1, dev)
    mask = (ids != 1).long()
    pend = fm_index_generate_joint(m, ix, ids, mask, [dict(batch=3, max_length=6), dict(batch=3, max_length=8, force_decoding_from=[2], eos_token_id=title_eos)],
                                   num_beams=4, length_penalty=0.0)
    for i, pg in enumerate(pend):
        steps, final, B, K, _ = pg._args
        rep = compare_beam_history(m, ids[3 * i:3 * i + 3], mask[3 * i:3 * i + 3], steps, final, B, K, tol=0.1)
        assert rep["violations"] == 0 and rep["values"] > 30, rep
