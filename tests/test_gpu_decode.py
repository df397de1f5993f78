"""GPU parity of the decode path: fm_index_generate (HIP constraint kernels +
step decoder on cuda:0) against the CPU restatement of the reference loop driven
by HF's cache-free forward on the same seeded tiny BART."""
import pytest
import torch

from tests.helpers import MODEL_GEOMETRIES, MODEL_IDS

pytestmark = pytest.mark.gpu


def _assert_decoder_path(model, geom):
    """a silent fall back from the fused sealnn_* kernels to the torch ops must fail the test"""
    st = model._seal_step_decoder._st
    assert st.fused is (geom["d_model"] // geom["heads"] == 64), (geom, st.fused)


@pytest.mark.parametrize("geom", MODEL_GEOMETRIES, ids=MODEL_IDS)
@pytest.mark.parametrize("kw", [
    dict(max_length=6, num_beams=3, length_penalty=0.0),
    dict(max_length=8, num_beams=5, length_penalty=0.0, force_decoding_from=[2], eos_token_id=7),
    dict(max_length=5, num_beams=4, length_penalty=1.0, always_allow_eos=True),
    dict(max_length=5, num_beams=3, length_penalty=0.0, stop_at_count=2),
    dict(max_length=5, num_beams=15, length_penalty=0.0),        # BASELINE configs[1]: beam 15
    dict(max_length=4, num_beams=30, length_penalty=0.0),        # configs[4]: beam 30 (top-2K = 60 of the 64 k_row_pick ranks)
])
def test_fm_index_generate_matches_reference_restatement(kw, geom):
    from oracle.beam_oracle import oracle_fm_index_generate
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex, fm_index_generate
    from tests.helpers import hf_logits_fn, make_docs, tiny_bart, valid_set
    vocab = 120
    dev = torch.device("cuda:0")
    m_cpu = tiny_bart(vocab, **geom)
    m_gpu = tiny_bart(vocab, **geom).to(dev)
    docs = make_docs(3, 150, vocab, title_sep=7)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)
    torch.manual_seed(2)
    enc_ids = torch.randint(4, vocab, (3, 8))
    enc_mask = torch.ones_like(enc_ids)
    K = kw["num_beams"]
    eos = kw.get("eos_token_id", 2)
    got = fm_index_generate(m_gpu, ix, enc_ids.to(dev), enc_mask.to(dev), min_length=1, keep_history=True, **kw)
    _assert_decoder_path(m_gpu, geom)
    want = oracle_fm_index_generate(hf_logits_fn(m_cpu, enc_ids, enc_mask, K), orc, 3, K, kw["max_length"], vocab,
                                    decoder_start_token_id=2, pad_token_id=1, eos_token_id=eos,
                                    length_penalty=kw["length_penalty"], force_decoding_from=kw.get("force_decoding_from"),
                                    stop_at_count=kw.get("stop_at_count", 0), always_allow_eos=kw.get("always_allow_eos", False))
    for g, w in zip(got, want):
        te = eos if kw.get("force_decoding_from") else None
        gv, wv = valid_set(g, orc, te), valid_set(w, orc, te)
        assert set(gv) == set(wv) and (len(gv) > 0 or te is not None)
        for k in gv:
            assert len(gv[k]) == len(wv[k])
            for a, b in zip(sorted(gv[k]), sorted(wv[k])):
                assert abs(a - b) <= 1e-4, (k, a, b)     # north_star: beam scores within 1e-4


def test_graph_captured_step_matches_eager_step():
    from seal_amd.bart_decoder import BartStepDecoder
    from tests.helpers import tiny_bart
    dev = torch.device("cuda:0")
    m = tiny_bart(120).to(dev)
    torch.manual_seed(3)
    enc_ids = torch.randint(4, 120, (3, 11), device=dev)
    enc_mask = torch.ones_like(enc_ids)
    enc_mask[2, 7:] = 0
    enc_ids[2, 7:] = 1
    K, T = 4, 9
    eager, graph = BartStepDecoder(m), BartStepDecoder(m)
    eager.use_graph = False
    for rep in range(2):                      # second round re-uses the captured graph and the stale cache
        enc = eager.encode(enc_ids, enc_mask)
        eager.start(enc, enc_mask, K, T)
        graph.start(enc, enc_mask, K, T)
        toks = torch.full((3 * K,), 2, device=dev)
        for t in range(T - 1):
            a, b = eager.step(toks), graph.step(toks)
            assert torch.allclose(a, b, atol=2e-5, rtol=1e-5), (rep, t)
            toks = torch.randint(4, 120, (3 * K,), device=dev)
            perm = torch.arange(3 * K, device=dev).view(3, K)[:, torch.randperm(K, device=dev)].reshape(-1)
            eager.reorder(perm)
            graph.reorder(perm)
        enc_ids = torch.roll(enc_ids, 1, 0)


@pytest.mark.parametrize("S", [1, 17, 63, 64])
def test_fused_step_decoder_matches_hf_cache_free_forward(S):
    """``BartStepDecoder.step`` at head_dim 64 -- sealnn_self_attn_step (ancestry-addressed KV cache),
    sealnn_cross_attn_step, sealnn_add_layernorm inside the captured graph -- against HF's own cache-free fp32
    forward over a whole 17-position decode with the beams re-ranked at random after every step
    (reference beam_search.py:231-253 forward + 331-332 ``_reorder_cache``), masked encoder positions included."""
    from seal_amd.bart_decoder import BartStepDecoder
    from tests.helpers import tiny_bart
    dev = torch.device("cuda:0")
    vocab, B, K, T = 120, 3, 5, 17
    m = tiny_bart(vocab, d_model=128, heads=2, max_positions=64).to(dev)
    g = torch.Generator(device="cpu").manual_seed(S)
    enc_ids = torch.randint(4, vocab, (B, S), generator=g).to(dev)
    enc_mask = torch.ones_like(enc_ids)
    if S > 2:                                  # ragged batch: padded + masked encoder tails
        enc_mask[1, S // 2:] = 0
        enc_ids[1, S // 2:] = 1
        enc_mask[2, S - 1:] = 0
        enc_ids[2, S - 1:] = 1
    dec = BartStepDecoder(m)
    for rep in range(2):                       # the second decode replays the captured graph over a stale cache
        enc = dec.encode(enc_ids, enc_mask)
        dec.start(enc, enc_mask, K, T)
        rows = torch.full((B * K, 1), 2, dtype=torch.long, device=dev)
        ids_rep, am_rep = enc_ids.repeat_interleave(K, 0), enc_mask.repeat_interleave(K, 0)
        for t in range(T - 1):
            got = dec.step(rows[:, -1])
            assert dec._st.fused is True
            with torch.no_grad():
                want = m(input_ids=ids_rep, attention_mask=am_rep, decoder_input_ids=rows).logits[:, -1, :]
            finite = torch.isfinite(want)
            assert torch.equal(finite, torch.isfinite(got))
            err = (got[finite] - want[finite]).abs().max().item()
            assert err <= 2e-5, (rep, t, err)
            nxt = torch.randint(4, vocab, (B * K,), generator=g).to(dev)
            perm = (torch.arange(B * K).view(B, K).gather(1, torch.randint(0, K, (B, K), generator=g))).reshape(-1).to(dev)
            rows = torch.cat([rows[perm], nxt[:, None]], 1)     # beams may be duplicated and dropped, as in a real search
            dec.reorder(perm)
        enc_ids = torch.roll(enc_ids, 1, 0)
        enc_mask = torch.roll(enc_mask, 1, 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_first_step_shared_by_the_beams_of_a_query(dtype):
    """``step(tokens, beams_identical=True)`` at position 0 (one row of arithmetic per query, position 0 of the cache written for
    the first beam only, the ancestry table pointing the other beams at it): the logits of every beam == the logits of the
    full-width step within fp32 GEMM noise, and so are the logits of the following steps -- which read that cache slot through
    random re-rankings and through a ``narrow`` -- against HF's cache-free forward; with a logit bias; on a replayed graph."""
    from seal_amd.bart_decoder import BartStepDecoder
    from tests.helpers import tiny_bart
    dev = torch.device("cuda:0")
    vocab, B, K, T, S = 120, 4, 5, 9, 11
    m = tiny_bart(vocab, d_model=128, heads=2, max_positions=64).to(dev)
    tol = 2e-5 if dtype == torch.float32 else 0.15
    run = m if dtype == torch.float32 else __import__("copy").deepcopy(m).to(dtype)
    g = torch.Generator(device="cpu").manual_seed(5)
    enc_ids = torch.randint(4, vocab, (B, S), generator=g).to(dev)
    enc_mask = torch.ones_like(enc_ids)
    enc_mask[1, 6:] = 0
    enc_ids[1, 6:] = 1
    bias = torch.randn(B, vocab, generator=g).to(dev)
    dec = BartStepDecoder(run)
    assert dec.shared_first_step is True             # the default (a class attribute of BartStepDecoder)
    for rep in range(2):
        enc = dec.encode(enc_ids, enc_mask)
        dec.start(enc, enc_mask, K, T, narrow_plan=(2,))
        dec.logit_bias = bias
        assert dec._st.first_graph is not None
        rows = torch.full((B * K, 1), 2, dtype=torch.long, device=dev)
        ids_rep, am_rep, bias_rep = enc_ids.repeat_interleave(K, 0), enc_mask.repeat_interleave(K, 0), bias.repeat_interleave(K, 0)
        for t in range(T - 1):
            got = dec.step(rows[:, -1], beams_identical=(t == 0))
            assert got.shape == (rows.shape[0], vocab) and dec._st.fused is True
            if t == 0:
                assert torch.equal(got.view(B, K, vocab), got.view(B, K, vocab)[:, :1].expand(B, K, vocab))
            with torch.no_grad():
                want = m(input_ids=ids_rep, attention_mask=am_rep, decoder_input_ids=rows).logits[:, -1, :] + bias_rep
            finite = torch.isfinite(want)
            assert torch.equal(finite, torch.isfinite(got))
            err = (got[finite] - want[finite]).abs().max().item()
            assert err <= tol, (rep, t, err)
            n, b = rows.shape[0], rows.shape[0] // K
            nxt = torch.randint(4, vocab, (n,), generator=g).to(dev)
            perm = (torch.arange(n).view(b, K).gather(1, torch.randint(0, K, (b, K), generator=g))).reshape(-1).to(dev)
            rows = torch.cat([rows[perm], nxt[:, None]], 1)
            dec.reorder(perm)
            if t == 3:                          # the first two queries leave: the others keep reading position 0 through the re-based table
                dec.narrow(2)
                rows, ids_rep, am_rep, bias_rep = rows[2 * K:], ids_rep[2 * K:], am_rep[2 * K:], bias_rep[2 * K:]
        enc_ids, enc_mask, bias = torch.roll(enc_ids, 1, 0), torch.roll(enc_mask, 1, 0), torch.roll(bias, 1, 0)
    # the promise is only honoured at position 0; the switch turns the path off
    dec2 = BartStepDecoder(run)
    dec2.shared_first_step = False
    dec2.start(dec2.encode(enc_ids, enc_mask), enc_mask, K, T)
    assert dec2._st.first_graph is None
    full = dec2.step(torch.full((B * K,), 2, dtype=torch.long, device=dev), beams_identical=True)
    dec.start(dec.encode(enc_ids, enc_mask), enc_mask, K, T)
    dec.logit_bias = None
    one = dec.step(torch.full((B * K,), 2, dtype=torch.long, device=dev), beams_identical=True)
    finite = torch.isfinite(full)
    assert torch.equal(finite, torch.isfinite(one)) and (full[finite] - one[finite]).abs().max().item() <= tol


@pytest.mark.parametrize("narrow", ["1024", "0"], ids=["lds-select", "radix-select"])
@pytest.mark.parametrize("kw", [dict(), dict(force_decoding_from=[2], eos_token_id=7), dict(always_allow_eos=True),
                                dict(stop_at_count=2)])
def test_fused_constrained_topk_matches_unfused_step(kw, narrow, monkeypatch):
    """fmi_dev_constrained_topk against the reference's own sequence of ops on the same logits: same picks (as a set
    per query: ties/-inf fillers are unordered in torch.topk too), unconstrained scores within fp32 noise.
    Both ends of k_row_pick: rows of <= 1024 allowed tokens gathered from the bitmap, and the wide-row path."""
    from seal_amd import FMIndex
    from seal_amd.beam_search import IndexBasedLogitsProcessor, _inf_nan_remove
    from tests.helpers import kernel_options, make_docs
    vocab, B, K = 120, 5, 4
    dev = torch.device("cuda:0")
    docs = make_docs(3, 150, vocab - 8, title_sep=7)
    ix = FMIndex()
    ix.initialize(docs)
    from seal_amd._lib import check, lib
    check(lib().fmi_dev_set_option(ix.handle, b"topk_narrow", int(narrow)))      # (the index is this test's own: nothing to restore)
    eos = kw.get("eos_token_id", 2)
    proc = IndexBasedLogitsProcessor(ix, K, pad_token_id=1, eos_token_id=eos, force_decoding_from=kw.get("force_decoding_from"),
                                     stop_at_count=kw.get("stop_at_count", 0), always_allow_eos=kw.get("always_allow_eos", False))
    g = torch.Generator(device="cpu").manual_seed(0)
    import random
    rng = random.Random(0)
    for cur_len in (1, 2, 3, 5):
        rows = []
        for i in range(B * K):
            d = rng.choice(docs)
            a = 0 if kw.get("force_decoding_from") else rng.randrange(len(d))
            sent = ([2] + d[a:a + cur_len - 1] + [1] * cur_len)[:cur_len]
            if i % 6 == 5 and cur_len > 2:
                sent[-1] = rng.randrange(8, vocab - 8)
            rows.append(sent)
        ids = torch.tensor(rows, device=dev)
        logits = torch.randn(B * K, vocab, generator=g).to(dev) * 3
        logits[:, 0] = float("-inf")
        beam_scores = (torch.randn(B * K, generator=g) * 2).to(dev)
        flat, unc = proc.fused_topk(ids, logits, beam_scores, B, K)
        processed = _inf_nan_remove(torch.log_softmax(logits, -1))
        u = (processed + beam_scores[:, None]).view(B, K * vocab)
        c = (proc(ids, processed) + beam_scores[:, None]).view(B, K * vocab)
        want_c, want_i = torch.topk(c, 2 * K, dim=1)
        for q in range(B):
            finite = int(torch.isfinite(want_c[q]).sum())
            assert set(flat[q, :finite].tolist()) == set(want_i[q, :finite].tolist()), (cur_len, q)
            assert torch.allclose(unc[q, :finite], u[q].gather(0, flat[q, :finite]), atol=1e-5)
            assert torch.allclose(torch.sort(unc[q, :finite], descending=True).values, want_c[q, :finite], atol=1e-5)
            # fillers: not allowed (constrained -inf) but carry their real unconstrained score
            for j in range(finite, 2 * K):
                assert c[q, flat[q, j]] == float("-inf")
                a, b = unc[q, j].item(), u[q, flat[q, j]].item()
                assert a == b or abs(a - b) <= 1e-5
            assert len(set(flat[q].tolist())) == 2 * K


@pytest.mark.parametrize("tree", ["1", "0"], ids=["prefix-tree", "maximal-parent-rows"])
@pytest.mark.parametrize("geom", MODEL_GEOMETRIES, ids=MODEL_IDS)
def test_rescore_keys_tree_shared_teacher_forcing_matches_one_hf_row_per_key(geom, tree, monkeypatch):
    """prefix-sharing rescoring (``share_prefixes=True``: one decoder position per distinct prefix through
    sealnn_tree_self_attn, or one row per maximal parent through sealnn_causal_self_attn; at head_dim 64 both with
    sealnn_cross_attn_* / sealnn_add_layernorm and per-query cross K/V) == one row per key through HF's own
    forward (``share_prefixes=False``, the reference's batching, keys.py:64-141).  Encoder lengths 1, 63 and 64
    in one ragged batch, keys up to 17 tokens (T = 17 decoder positions)."""
    import numpy as np
    from seal_amd.bart_decoder import BartStepDecoder
    from seal_amd.keys import rescore_keys
    from tests.helpers import tiny_bart
    dev = torch.device("cuda:0")
    m = tiny_bart(120, **geom).to(dev)
    from seal_amd import keys as keys_mod
    monkeypatch.setattr(keys_mod, "RESCORE_TREE", tree == "1")
    fused_calls = []
    real, real_tree, real_graph = BartStepDecoder.teacher_logits, BartStepDecoder.tree_logits, BartStepDecoder.tree_hidden_graph
    monkeypatch.setattr(BartStepDecoder, "teacher_logits", lambda self, *a: (fused_calls.append(1), real(self, *a))[1])
    monkeypatch.setattr(BartStepDecoder, "tree_logits",
                        lambda self, *a: (fused_calls.append(1) if a[6] is not None else None, real_tree(self, *a))[1])

    def graphed(self, *a):            # the tree forward as one graph replay (the fused kernels inside): None = not applicable
        out = real_graph(self, *a)
        if out is not None:
            fused_calls.append(1)
        return out
    monkeypatch.setattr(BartStepDecoder, "tree_hidden_graph", graphed)
    rng = np.random.default_rng(0)
    inputs = [[0] + rng.integers(4, 118, size=n).tolist() + [2] for n in (0, 61, 62, 5)]
    inputs[0] = [2]                                      # a one-token encoder input
    keys = []
    for _ in range(4):
        base = rng.integers(4, 118, size=17).tolist()
        other = rng.integers(4, 118, size=6).tolist()
        kk = [base[:i] for i in range(1, 18)] + [other[:i] for i in range(2, 7)] + [[2] + base[:3], base[:4] + [2], [7] + other[:2] + [7]]
        keys.append([(-1.0, k) for k in kk])
    bias = torch.randn(4, 120, device=dev)
    for kw in (dict(), dict(strip_from_bos=[2, 7], strip_from_eos=[7, 2]), dict(logit_bias=bias)):
        n0 = len(fused_calls)
        a = rescore_keys(m, inputs, keys, batch_size=4, share_prefixes=True, **kw)
        assert (len(fused_calls) > n0) is (geom["d_model"] // geom["heads"] == 64)      # no silent fallback
        b = rescore_keys(m, inputs, keys, batch_size=4, share_prefixes=False, **kw)
        for qa, qb in zip(a, b):
            assert [k for _, k in qa] == [k for _, k in qb]
            for (sa, _), (sb, _) in zip(qa, qb):
                assert abs(sa - sb) <= 3e-5 * max(1.0, abs(sb)), kw


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(force_decoding_from=[2], eos_token_id=7), dict(stop_at_count=2)])
def test_incremental_constraint_state_equals_full_prefix_search(kw):
    """constrained_beam_search hands the previous step's beam_idx to fmi_dev_constrained_topk_step, which then
    advances every row's prefix range by one backward-search step; a processor that withholds the parents
    makes the index re-search the whole prefix every step (the reference's way).  Same logits -> the two
    loops must produce identical histories, bit for bit."""
    from seal_amd import FMIndex
    from seal_amd.beam_search import IndexBasedLogitsProcessor, constrained_beam_search
    from tests.helpers import kernel_options, make_docs
    vocab, B, K, T = 120, 6, 5, 9
    dev = torch.device("cuda:0")
    docs = make_docs(5, 200, vocab - 8, title_sep=7)
    eos = kw.get("eos_token_id", 2)

    class RandomDecoder:
        """deterministic logits per (step, row): biased towards low token ids so that eos/pad and real
        corpus continuations all show up"""
        def __init__(self):
            self.t = 0

        def step(self, tokens):
            g = torch.Generator(device="cpu").manual_seed(100 + self.t)
            self.t += 1
            lg = torch.randn(B * K, vocab, generator=g) * 3
            lg[:, 0] = float("-inf")
            return lg.to(dev)

        def reorder(self, beam_idx):
            pass

    class NoParents(IndexBasedLogitsProcessor):
        def fused_topk(self, input_ids, logits, beam_scores, batch, num_beams, parent_rows=None, tag=None):
            return super().fused_topk(input_ids, logits, beam_scores, batch, num_beams, parent_rows=None, tag=tag)

    out = []
    for cls in (IndexBasedLogitsProcessor, NoParents):
        ix = FMIndex()                  # one index handle per loop: the state lives in the handle
        ix.initialize(docs)
        proc = cls(ix, K, pad_token_id=1, eos_token_id=eos, force_decoding_from=kw.get("force_decoding_from"),
                   stop_at_count=kw.get("stop_at_count", 0))
        steps, final = constrained_beam_search(RandomDecoder(), B, K, T, 2, eos, proc, device=dev)
        out.append(([tuple(x.tolist() for x in s) for s in steps], final[0].tolist(), final[1].tolist()))
    assert out[0] == out[1]


@pytest.mark.gpu
@pytest.mark.parametrize("chain", [1, 0], ids=["chained", "unchained"])
@pytest.mark.parametrize("corpus", [(40, 12, 3, 7), (20, 300, 8, 16)], ids=["tiny-corpus", "few-symbols-long-lists"])
@pytest.mark.parametrize("kw", [dict(), dict(force_decoding_from=[2], eos_token_id=7), dict(stop_at_count=2), dict(always_allow_eos=True)])
def test_masks_the_beam_step_applies_equal_the_oracle_masks_row_by_row(kw, chain, corpus):
    """every allowed-token bitmap a decode through ``fmi_dev_beam_step`` actually applies (captured per step: table call, chained calls with
    rows in interval and in list mode, finished rows) equals the reference's mask for the same ``input_ids`` (IndexBasedLogitsProcessor.__call__
    restated over the oracle).  A tiny corpus under a wide beam: queries run out of finite candidates, the beam fills up with not-allowed
    tokens (quirk Q4) and rows CONTINUE finished rows -- the case in which the kept ranges of rounds 2-4 went stale."""
    import numpy as np
    from oracle.beam_oracle import oracle_logits_mask
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex
    from seal_amd.beam_search import IndexBasedLogitsProcessor, constrained_beam_search
    from tests.helpers import kernel_options, make_docs
    # (the second corpus: a dozen symbols over 300 documents -- prefixes of one and two tokens hold hundreds of suffix-array rows, so the
    #  rows enter list mode with lists that span several 64-entry chunks of their wave)
    vocab, n_docs, min_len, max_len = corpus
    B, K, T = 4, 7, 9
    dev = torch.device("cuda:0")
    docs = make_docs(13, n_docs, vocab - 8, min_len=min_len, max_len=max_len, title_sep=7)
    eos = kw.get("eos_token_id", 2)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)

    class RandomDecoder:
        def __init__(self):
            self.t = 0

        def step(self, tokens):
            g = torch.Generator(device="cpu").manual_seed(300 + self.t)
            self.t += 1
            lg = torch.randn(B * K, vocab, generator=g) * 3
            lg[:, 0] = float("-inf")
            return lg.to(dev)

        def reorder(self, beam_idx):
            pass

    trace = []
    ix.set_trace(trace)
    proc = IndexBasedLogitsProcessor(ix, K, pad_token_id=1, eos_token_id=eos, force_decoding_from=kw.get("force_decoding_from"),
                                     stop_at_count=kw.get("stop_at_count", 0), always_allow_eos=kw.get("always_allow_eos", False))
    with kernel_options(ix, chain_steps=chain):
        steps, final = constrained_beam_search(RandomDecoder(), B, K, T, 2, eos, proc, device=dev)
    ix.set_trace(None)
    masks = [op for op in trace if op[0] == "mask"]
    assert len(masks) == T - 2 and all(len(op) == 5 and op[4] is not None for op in masks)
    continued_finished = 0
    for op in masks:
        ids, bits = op[1].tolist(), op[4].cpu().numpy().view(np.uint32)
        got = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :vocab].astype(bool)
        want = oracle_logits_mask(orc, ids, vocab, K, pad_token_id=1, eos_token_id=eos, force_decoding_from=kw.get("force_decoding_from"),
                                  stop_at_count=kw.get("stop_at_count", 0), always_allow_eos=kw.get("always_allow_eos", False))
        assert np.array_equal(got, np.asarray(want, dtype=bool)), (len(ids[0]), np.nonzero((got != want).any(axis=1))[0][:5])
        continued_finished += sum(1 for r in ids if any(t in (eos, 1) for t in r[1:-1]) and r[-1] not in (eos, 1))
    if not kw.get("stop_at_count") and not kw.get("always_allow_eos") and n_docs < 100:
        assert continued_finished > 0, "the case under test must occur: a live row whose prefix runs through an eos / pad"


@pytest.mark.gpu
def test_chains_run_for_a_call_that_never_comes_do_not_leak_into_the_next_one():
    """``k_beam_advance`` writes list rows' tokens into the workspace bitmap the NEXT constraint call will fill.  If that call never comes --
    the loop is abandoned, or another kind of call uses the handle next -- the pre-filled bits must not show up in it: a protocol call
    (``IndexBasedLogitsProcessor.__call__`` -> ``fmi_dev_constrain_scores``, which takes its bitmap from the same workspace) right after a
    chained step still gives the oracle's mask, and so does a fresh decode loop on the same handle afterwards."""
    import numpy as np
    from oracle.beam_oracle import oracle_logits_mask
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex
    from seal_amd.beam_search import IndexBasedLogitsProcessor, _BeamStepper, constrained_beam_search
    from tests.helpers import make_docs
    vocab, B, K = 60, 3, 5
    dev = torch.device("cuda:0")
    docs = make_docs(21, 80, vocab - 8, min_len=4, max_len=9, title_sep=7)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)
    proc = IndexBasedLogitsProcessor(ix, K, pad_token_id=1, eos_token_id=2)
    spec = [dict(batch=B, max_length=8, eos_token_id=2, processor=proc)]
    stepper = _BeamStepper(spec, K, 2, dev)
    g = torch.Generator(device="cpu").manual_seed(9)
    for cur in (1, 2, 3):                                     # the last of them runs chains (and pre-fills bits) for a 4th call ...
        lg = (torch.randn(B * K, vocab, generator=g) * 3).to(dev)
        lg[:, 0] = float("-inf")
        stepper.step([0], lg, cur, tag=777, chain_next=cur >= 2, tokens_out=None, anc=None)
    # ... that never comes: a protocol call on arbitrary rows instead
    rng = np.random.default_rng(3)
    rows = []
    for _ in range(B * K):
        d = docs[int(rng.integers(len(docs)))]
        a = int(rng.integers(len(d) - 2))
        rows.append([2] + d[a:a + 2])
    scores = torch.randn(len(rows), vocab, device=dev)
    out = proc(torch.tensor(rows, device=dev), scores)
    allowed = torch.from_numpy(oracle_logits_mask(orc, rows, vocab, K, pad_token_id=1, eos_token_id=2)).to(dev)
    assert torch.equal(out, torch.where(allowed, scores, torch.full_like(scores, float("-inf"))))
    # and a whole new loop on the same handle: every applied mask is the oracle's

    class Logits:
        t = 0

        def step(self, tokens):
            gg = torch.Generator(device="cpu").manual_seed(500 + self.t)
            self.t += 1
            lg = torch.randn(B * K, vocab, generator=gg) * 3
            lg[:, 0] = float("-inf")
            return lg.to(dev)

        def reorder(self, beam_idx):
            pass
    trace = []
    ix.set_trace(trace)
    constrained_beam_search(Logits(), B, K, 7, 2, 2, proc, device=dev)
    ix.set_trace(None)
    masks = [op for op in trace if op[0] == "mask"]
    assert len(masks) == 5
    for op in masks:
        got = np.unpackbits(op[4].cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :vocab].astype(bool)
        want = oracle_logits_mask(orc, op[1].tolist(), vocab, K, pad_token_id=1, eos_token_id=2)
        assert np.array_equal(got, np.asarray(want, dtype=bool)), len(op[1][0])


@pytest.mark.gpu
def test_topk_selection_paths_agree_on_ties(monkeypatch):
    """k_row_pick has two selection paths (bitmap gather for rows of <= 1024 allowed tokens, lower bound + collect beyond);
    on heavily tied logits both must return the same picks in the same order (ties go to the lower token id)."""
    from seal_amd import FMIndex
    from seal_amd.beam_search import IndexBasedLogitsProcessor
    from tests.helpers import kernel_options, make_docs
    vocab, B, K = 300, 4, 6
    dev = torch.device("cuda:0")
    docs = make_docs(11, 400, vocab - 8, title_sep=7)
    ix = FMIndex()
    ix.initialize(docs)
    proc = IndexBasedLogitsProcessor(ix, K, pad_token_id=1, eos_token_id=2)
    g = torch.Generator(device="cpu").manual_seed(3)
    import random
    rng = random.Random(3)
    for cur_len in (1, 2, 3):
        rows = []
        for _ in range(B * K):
            d = rng.choice(docs)
            a = rng.randrange(len(d))
            rows.append(([2] + d[a:a + cur_len - 1] + [1] * cur_len)[:cur_len])
        ids = torch.tensor(rows, device=dev)
        logits = torch.randint(-2, 3, (B * K, vocab), generator=g).float().to(dev)     # five distinct values: ties everywhere
        beam_scores = torch.randint(-2, 2, (B * K,), generator=g).float().to(dev)
        got = []
        for narrow in (1024, 0):
            with kernel_options(ix, topk_narrow=narrow):
                flat, unc = proc.fused_topk(ids, logits, beam_scores, B, K)
            got.append((flat.tolist(), unc.tolist()))
        assert got[0] == got[1], cur_len


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["randn", "five-values", "constant", "nan-row"])
def test_row_pick_paths_agree_at_bart_vocabulary_and_match_torch(kind, monkeypatch):
    """k_row_pick at BART's 50 265 tokens: its three selection paths -- bitmap walk + LDS ranking (narrow rows), the
    thread-maxima lower bound (wide rows), the exact radix select (mass ties; forced with SEALFM_TOPK_LEGACY) -- return
    exactly the same picks, order and scores, ties included, and agree with the reference's op sequence in torch
    (beam_search.py:244-307).  Allowed sets from 5 to 45 000 tokens per row; constant logits put > 1024 equal keys in
    front of the lower bound, i.e. exercise the automatic fall-through to the radix select and its token-order ties."""
    import ctypes
    from seal_amd import FMIndex
    from seal_amd._lib import check, lib
    from seal_amd.beam_search import _inf_nan_remove
    from tests.helpers import kernel_options, make_docs
    V, B, K = 50265, 3, 5
    want, rows = 2 * K, B * K
    dev = torch.device("cuda:0")
    ix = FMIndex()
    ix.initialize(make_docs(3, 20, 100, title_sep=7))
    g = torch.Generator(device="cpu").manual_seed(11)
    st = torch.cuda.current_stream(dev).cuda_stream
    scratch = torch.empty(rows * (3 + 2 * want) + 64, dtype=torch.float32, device=dev)
    ids = torch.full((rows, 1), 2, dtype=torch.long, device=dev)
    ff = (ctypes.c_int64 * 1)(0)
    for n_allowed in (5, 40, 1000, 1025, 3000, 45000):
        allowed = torch.zeros(V, dtype=torch.bool)
        allowed[torch.randperm(V - 1, generator=g)[:n_allowed] + 1] = True
        words = torch.zeros((V + 31) // 32 * 32, dtype=torch.int64)
        words[:V] = allowed.long()
        bits = (words.view(-1, 32) << torch.arange(32)).sum(1)
        bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32).to(dev)
        if kind == "randn":
            logits = torch.randn(rows, V, generator=g) * 4
        elif kind == "five-values":
            logits = torch.randint(-2, 3, (rows, V), generator=g).float()
        elif kind == "constant":
            logits = torch.full((rows, V), 0.25)
        else:
            logits = torch.randn(rows, V, generator=g)
            logits[1, 77] = float("nan")
            logits[2, 5] = float("inf")
        logits[:, 0] = float("-inf")
        logits = logits.to(dev).contiguous()
        beam_scores = (torch.randn(rows, generator=g) * 2).to(dev)
        got = {}
        for mode, opts in (("select", {}), ("wide-path", dict(topk_narrow=0)), ("radix", dict(topk_legacy=1)),
                           ("radix-everywhere", dict(topk_legacy=1, topk_narrow=0))):
            top_idx = torch.empty(B, want, dtype=torch.int64, device=dev)
            top_con = torch.empty(B, want, dtype=torch.float32, device=dev)
            top_unc = torch.empty(B, want, dtype=torch.float32, device=dev)
            with kernel_options(ix, **opts):
                check(lib().fmi_dev_constrained_topk_step(ix.handle, st, B, K, 1, ids.data_ptr(), logits.data_ptr(), beam_scores.data_ptr(), V, 0,
                                                          1, 2, ff, 0, 0, 0, bits.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                                          top_idx.data_ptr(), top_con.data_ptr(), top_unc.data_ptr(), 0, None))
            torch.cuda.synchronize()
            got[mode] = (top_idx.cpu(), top_con.cpu(), top_unc.cpu())
        for mode in ("wide-path", "radix", "radix-everywhere"):
            for a, b in zip(got["select"], got[mode]):
                assert torch.equal(a, b) or torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0)), (n_allowed, mode)
        if kind == "nan-row":
            continue        # torch.topk orders NaN rows its own way; the paths above agree with each other
        processed = _inf_nan_remove(torch.log_softmax(logits, -1))
        u = (processed + beam_scores[:, None]).view(B, K * V)
        c = torch.where(allowed.to(dev)[None, :], processed, torch.full_like(processed, float("-inf")))
        c = (c + beam_scores[:, None]).view(B, K * V)
        want_c, _ = torch.topk(c, want, dim=1)
        flat, con, unc = (t.to(dev) for t in got["select"])
        assert torch.allclose(con, want_c, atol=1e-5), n_allowed                      # the same scores, in descending order
        assert torch.allclose(unc, u.gather(1, flat), atol=1e-5)                       # each pick carries its own unconstrained score
        assert torch.allclose(c.gather(1, flat), con, atol=1e-5)                       # ... and is an allowed token with that score
        for q in range(B):
            assert len(set(flat[q].tolist())) == want
            # ties go to the lower flat index: among candidates of equal score the picks are the lowest ones
            if kind != "randn":
                vals = c[q]
                mine = vals[flat[q]].tolist()
                for v in set(mine):
                    cand = torch.nonzero(vals == v).flatten().tolist()
                    picked = sorted(i for i, s in zip(flat[q].tolist(), mine) if s == v)
                    assert picked == cand[:len(picked)], (n_allowed, q, v)


@pytest.mark.gpu
@pytest.mark.parametrize("S,T", [(17, 11), (64, 16), (1, 1), (33, 17)])
def test_cross_attention_over_runs_is_bit_identical_to_the_per_row_kernel(S, T):
    """sealnn_cross_attn_runs (teacher forcing: K/V of a (sequence, head) staged once for its T positions) against
    sealnn_cross_attn_rows (one wave per row, K/V from memory) and a torch reference"""
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(S * 100 + T)
    nq, heads, n_seq = 5, 3, 7
    rows = n_seq * T
    q = torch.randn(rows, heads, 64, generator=g).to(dev)
    ck = torch.randn(nq, heads, 64, S, generator=g).to(dev)
    cv = torch.randn(nq, heads, S, 64, generator=g).to(dev)
    bias = torch.zeros(nq, S)
    bias[torch.rand(nq, S, generator=g) < 0.2] = torch.finfo(torch.float32).min
    bias[:, 0] = 0
    bias = bias.to(dev)
    seq_q = torch.randint(0, nq, (n_seq,), generator=g)
    row_batch = seq_q.repeat_interleave(T).to(torch.int32).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    a, b = torch.empty(rows, heads * 64, device=dev), torch.empty(rows, heads * 64, device=dev)
    check(lib().sealnn_cross_attn_rows(st, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), row_batch.data_ptr(), rows, heads, S,
                                       0.125, a.data_ptr()))
    check(lib().sealnn_cross_attn_runs(st, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), row_batch.data_ptr(), rows, T, heads, S,
                                       0.125, b.data_ptr()))
    assert torch.equal(a, b)
    # runs that do NOT line up with the queries (the rescoring tree: a query's nodes are consecutive, 16 rows per workgroup; a row of another
    # query than its run's first reads its own K / V from memory), the last run short
    for group in (16, 5, 64):
        c = torch.empty(rows, heads * 64, device=dev)
        check(lib().sealnn_cross_attn_runs(st, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), row_batch.data_ptr(), rows, group, heads, S,
                                           0.125, c.data_ptr()))
        assert torch.equal(a, c), group
    rb = row_batch.long()
    att = torch.softmax(torch.einsum("rhd,rhds->rhs", q * 0.125, ck[rb]) + bias[rb][:, None, :], -1)
    ref = torch.einsum("rhs,rhsd->rhd", att, cv[rb]).reshape(rows, heads * 64)
    assert torch.allclose(b, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 5, 17])
def test_tree_self_attention_matches_the_row_kernel_and_torch(T):
    """sealnn_tree_self_attn: (a) on chains -- node (n, j) = position j of sequence n, ancestors = its own row -- bit-identical
    to sealnn_causal_self_attn; (b) on a random forest against a torch gather-softmax reference"""
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(T)
    heads, n_seq = 3, 9
    st = torch.cuda.current_stream(dev).cuda_stream
    qkv = torch.randn(n_seq * T, 3 * heads * 64, generator=g).to(dev)
    a, b = torch.empty(n_seq * T, heads * 64, device=dev), torch.empty(n_seq * T, heads * 64, device=dev)
    check(lib().sealnn_causal_self_attn(st, qkv.data_ptr(), n_seq, T, heads, 0.125, a.data_ptr()))
    anc = torch.full((n_seq * T, T), -1, dtype=torch.int32)
    for n in range(n_seq):
        for j in range(T):
            anc[n * T + j, :j + 1] = torch.arange(n * T, n * T + j + 1, dtype=torch.int32)
    anc = anc.to(dev)
    check(lib().sealnn_tree_self_attn(st, qkv.data_ptr(), anc.data_ptr(), n_seq * T, T, heads, 0.125, b.data_ptr()))
    assert torch.equal(a, b)
    # random forest: node i > 0 hangs below a random earlier node (or is a root), depth < T
    N = 200
    parent = [-1] * N
    rows = [[0]]
    for i in range(1, N):
        p = int(torch.randint(-1, i, (1,), generator=g))
        if p >= 0 and len(rows[p]) >= T:
            p = -1
        parent[i] = p
        rows.append((rows[p] if p >= 0 else []) + [i])
    A = max(len(r) for r in rows)
    anc = torch.full((N, A), -1, dtype=torch.int32)
    for i, r in enumerate(rows):
        anc[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
    qkv = torch.randn(N, 3 * heads * 64, generator=g).to(dev)
    out = torch.empty(N, heads * 64, device=dev)
    anc_d = anc.to(dev)
    check(lib().sealnn_tree_self_attn(st, qkv.data_ptr(), anc_d.data_ptr(), N, A, heads, 0.125, out.data_ptr()))
    v = qkv.view(N, 3, heads, 64)
    ix = anc_d.long().clamp(min=0)
    bias = torch.zeros(N, A, device=dev).masked_fill_(anc_d < 0, torch.finfo(torch.float32).min)[:, None, :]
    w = torch.softmax(torch.einsum("nhd,nahd->nha", v[:, 0] * 0.125, v[:, 1][ix]) + bias, -1)
    ref = torch.einsum("nha,nahd->nhd", w, v[:, 2][ix]).reshape(N, heads * 64)
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["torch-ops", "beam-step", "beam-step-unchained"])
@pytest.mark.parametrize("n_groups,stops", [(2, (0, 0, 0)), (3, (0, 0, 0)), (2, (3, 0, 0)), (3, (2, 0, 4))],
                         ids=["2", "3", "2-body-stop_at_count", "3-mixed-stop_at_count"])
def test_lockstep_groups_on_the_gpu_equal_separate_loops_bit_for_bit(n_groups, stops, mode, monkeypatch):
    """``constrained_beam_search_groups`` over the HIP constraint (ONE ``fmi_dev_constrained_topk_groups`` call per step for
    the stacked rows of 2 / 3 decodes with their own eos / forced prefix / length) on deterministic per-row logits: every
    group's history must equal its own separate loop's exactly -- per-row masks, per-query picks, the incremental prefix
    ranges across the step where the shorter decodes leave the loop (parents then name rows of the wider previous call).
    ``stops``: every decode keeps its OWN stop_at_count in the joint call (the reference gives it to the body decode only,
    retrieval.py:70-83 vs 162-176; round-3 advisor finding: the joint path applied the body's to the title rows as well).
    ``mode``: the separate loops always run round 4's form (the constraint + top-2K call, then the loop's own torch ops: the
    specification); the joint loop runs that form too, or round 5's ONE call per step (``fmi_dev_beam_step``: k_beam_advance does the
    scorer's bookkeeping, rewrites the ids in place and runs the chains of the next constraint call, whose rows are numbered from the
    rows that left), or the same with the chains left to the next call (``chain_steps=0``)."""
    import contextlib
    from seal_amd import FMIndex
    from seal_amd import beam_search as bs_mod
    from seal_amd.beam_search import IndexBasedLogitsProcessor, constrained_beam_search, constrained_beam_search_groups
    from tests.helpers import kernel_options, make_docs
    vocab, K = 120, 5
    dev = torch.device("cuda:0")
    docs = make_docs(5, 200, vocab - 8, title_sep=7)
    cfgs = [dict(batch=4, T=6, eos=2, ff=None, stop=stops[0]), dict(batch=3, T=9, eos=7, ff=[2], stop=stops[1]),
            dict(batch=2, T=11, eos=9, ff=[7], stop=stops[2])][:n_groups]

    class RandomDecoder:
        """deterministic logits per (group, step, row)"""
        def __init__(self, groups):
            self.groups, self.t = list(groups), 0

        def step(self, tokens):
            parts = []
            for gi, c in self.groups:
                g = torch.Generator(device="cpu").manual_seed(1000 * gi + self.t)
                lg = torch.randn(c["batch"] * K, vocab, generator=g) * 3
                lg[:, 0] = float("-inf")
                parts.append(lg)
            self.t += 1
            return torch.cat(parts).to(dev)

        def reorder(self, beam_idx):
            pass

        def narrow(self, nq):
            while nq > 0:
                nq -= self.groups.pop(0)[1]["batch"]
            assert nq == 0

    def proc(ix, c):
        return IndexBasedLogitsProcessor(ix, K, pad_token_id=1, eos_token_id=c["eos"], force_decoding_from=c["ff"], stop_at_count=c["stop"])
    want = []
    monkeypatch.setattr(bs_mod, "FUSED_BEAM_STEP", False)
    for gi, c in enumerate(cfgs):
        ix = FMIndex()
        ix.initialize(docs)
        steps, final = constrained_beam_search(RandomDecoder([(gi, c)]), c["batch"], K, c["T"], 2, c["eos"], proc(ix, c), device=dev)
        want.append(([tuple(x.tolist() for x in s) for s in steps], final[0].tolist(), final[1].tolist()))
    ix = FMIndex()
    ix.initialize(docs)
    specs = [dict(batch=c["batch"], max_length=c["T"], eos_token_id=c["eos"], processor=proc(ix, c)) for c in cfgs]
    monkeypatch.setattr(bs_mod, "FUSED_BEAM_STEP", mode != "torch-ops")
    with (kernel_options(ix, chain_steps=0) if mode == "beam-step-unchained" else contextlib.nullcontext()):
        got = constrained_beam_search_groups(RandomDecoder(list(enumerate(cfgs))), specs, K, 2, device=dev)
    for (steps, final), w in zip(got, want):
        assert ([tuple(x.tolist() for x in s) for s in steps], final[0].tolist(), final[1].tolist()) == w
    if mode != "torch-ops":
        assert got[0][0].packed is not None          # the history was written in place by fmi_dev_beam_step


@pytest.mark.gpu
def test_joint_generate_through_the_fused_decoder_matches_separate_generates():
    """``fm_index_generate_joint`` (body-like and title-like decode of the same queries as one loop through the graph-captured
    fused step decoder, the title rows carrying on alone in the views ``decoder.narrow`` switches to) against two
    ``fm_index_generate`` calls: same keys after the searcher's filters, scores within 1e-4 (GEMMs of another height)"""
    from oracle.seal_oracle import OracleFMIndex
    from seal_amd import FMIndex, fm_index_generate
    from seal_amd.beam_search import fm_index_generate_joint
    from tests.helpers import make_docs, tiny_bart, valid_set
    vocab, K = 120, 4
    dev = torch.device("cuda:0")
    m = tiny_bart(vocab, d_model=128, heads=2).to(dev)
    docs = make_docs(3, 150, vocab, title_sep=7)
    ix, orc = FMIndex(), OracleFMIndex()
    ix.initialize(docs)
    orc.initialize(docs)
    torch.manual_seed(2)
    body_ids = torch.randint(4, vocab, (3, 9), device=dev)
    title_ids = torch.randint(4, vocab, (3, 9), device=dev)
    body_ids[2, 6:] = 1
    title_ids[2, 6:] = 1
    bias = torch.randn(3, vocab, device=dev)
    jobs = [dict(batch=3, max_length=6, eos_token_id=None, force_decoding_from=None),
            dict(batch=3, max_length=9, eos_token_id=7, force_decoding_from=[2])]
    for rep in range(2):                          # the second round replays the captured graphs (wide and narrowed) over stale caches
        ids = torch.cat([body_ids, title_ids])
        pend = fm_index_generate_joint(m, ix, ids, (ids != 1).long(), jobs, num_beams=K, length_penalty=0.0, logit_bias=torch.cat([bias, bias]))
        assert m._seal_step_decoder._st.fused is True and m._seal_step_decoder._st.shape[0] == 3      # ended on the narrowed state
        got = [p.result() for p in pend]
        assert pend[0].enc.shape[0] == 3 and pend[1].first_logits.shape == (3, vocab)
        want = [fm_index_generate(m, ix, body_ids, (body_ids != 1).long(), min_length=1, max_length=6, num_beams=K, length_penalty=0.0,
                                  keep_history=True, logit_bias=bias),
                fm_index_generate(m, ix, title_ids, (title_ids != 1).long(), min_length=1, max_length=9, num_beams=K, length_penalty=0.0,
                                  keep_history=True, force_decoding_from=[2], eos_token_id=7, logit_bias=bias)]
        for gj, wj, te in zip(got, want, (None, 7)):
            for g, w in zip(gj, wj):
                gv, wv = valid_set(g, orc, te), valid_set(w, orc, te)
                assert set(gv) == set(wv)
                for k in gv:
                    for a, b in zip(sorted(gv[k]), sorted(wv[k])):
                        assert abs(a - b) <= 1e-4, (k, a, b)
        body_ids, title_ids = torch.roll(body_ids, 1, 0), torch.roll(title_ids, 1, 0)


@pytest.mark.gpu
def test_graph_replayed_tree_forward_equals_the_launch_by_launch_forward(monkeypatch):
    """``rescore_keys`` with the prefix-tree forward as ONE hipGraph replay (node count padded to a bucket, stale rows behind the
    real nodes, encoder length padded) against the same forward issued launch by launch (``keys.RESCORE_GRAPH = False``): the same
    scores up to the rounding of GEMMs of another height (<= 1e-5), call after call with different key sets in the same bucket"""
    import numpy as np
    from seal_amd import keys as keys_mod
    from seal_amd.keys import rescore_keys
    from tests.helpers import tiny_bart
    dev = torch.device("cuda:0")
    m = tiny_bart(120, d_model=128, heads=2).to(dev)
    rng = np.random.default_rng(3)
    for rep in range(3):
        inputs = [[0] + rng.integers(4, 118, size=int(rng.integers(1, 30))).tolist() + [2] for _ in range(4)]
        keys = []
        for _ in range(4):
            base = rng.integers(4, 118, size=int(rng.integers(3, 17))).tolist()
            kk = [base[:i] for i in range(1, len(base) + 1)] + [rng.integers(4, 118, size=int(rng.integers(1, 6))).tolist() for _ in range(20 - 5 * rep)]
            keys.append([(-1.0, k) for k in kk])
        bias = torch.randn(4, 120, device=dev)
        monkeypatch.setattr(keys_mod, "RESCORE_GRAPH", True)
        a = rescore_keys(m, inputs, keys, logit_bias=bias)
        monkeypatch.setattr(keys_mod, "RESCORE_GRAPH", False)
        b = rescore_keys(m, inputs, keys, logit_bias=bias)
        for qa, qb in zip(a, b):
            assert [k for _, k in qa] == [k for _, k in qb]
            assert max(abs(sa - sb) for (sa, _), (sb, _) in zip(qa, qb)) <= 1e-5
