"""Host logic of the decode side on CPU: the step decoder against HF's own
forward, and the tensorised beam loop + history scorer against the scalar
restatement of the reference loop (oracle/beam_oracle.py)."""
import pytest
import torch

from oracle.beam_oracle import oracle_fm_index_generate
from oracle.seal_oracle import OracleFMIndex
from seal_amd.bart_decoder import BartStepDecoder
from seal_amd.beam_search import fm_index_generate
from tests.helpers import OracleLogitsProcessor, hf_logits_fn, make_docs, tiny_bart, valid_set


def test_step_decoder_matches_hf_forward():
    m = tiny_bart()
    torch.manual_seed(1)
    enc_ids = torch.randint(4, 120, (3, 9))
    enc_mask = torch.ones_like(enc_ids)
    enc_mask[1, 6:] = 0
    enc_ids[1, 6:] = 1
    K = 2
    dec = BartStepDecoder(m)
    enc = dec.encode(enc_ids, enc_mask)
    dec.start(enc, enc_mask, K, 8)
    seq = torch.randint(4, 120, (3 * K, 6))
    seq[:, 0] = 2
    fn = hf_logits_fn(m, enc_ids, enc_mask, K)
    for t in range(6):
        got = dec.step(seq[:, t])
        want = fn(seq[:, :t + 1])
        assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), t
    # reorder: permute rows within each query and continue
    perm = torch.tensor([1, 0, 2, 3, 5, 4])
    dec.reorder(perm)
    seq2 = torch.cat([seq[perm], torch.randint(4, 120, (6, 1))], 1)
    got = dec.step(seq2[:, -1])
    assert torch.allclose(got, fn(seq2), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("kw", [
    dict(max_length=6, num_beams=3, length_penalty=0.0),
    dict(max_length=5, num_beams=4, length_penalty=1.0),
    dict(max_length=7, num_beams=3, length_penalty=0.0, force_decoding_from=[2], eos_token_id=7),
    dict(max_length=5, num_beams=2, length_penalty=0.0, always_allow_eos=True),
    dict(max_length=5, num_beams=3, length_penalty=0.0, stop_at_count=2),
    dict(max_length=4, num_beams=3, length_penalty=0.0, disable_fm_index=True),
])
def test_beam_loop_matches_reference_restatement(kw):
    vocab = 120
    m = tiny_bart(vocab)
    docs = make_docs(3, 150, vocab, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    torch.manual_seed(2)
    enc_ids = torch.randint(4, vocab, (4, 8))
    enc_mask = torch.ones_like(enc_ids)
    K = kw["num_beams"]
    eos = kw.get("eos_token_id", 2)
    pkw = dict(pad_token_id=1, eos_token_id=eos, force_decoding_from=kw.get("force_decoding_from"),
               stop_at_count=kw.get("stop_at_count", 0), always_allow_eos=kw.get("always_allow_eos", False))
    proc = OracleLogitsProcessor(orc, K, vocab, **pkw)
    got = fm_index_generate(m, None, enc_ids, enc_mask, min_length=1, keep_history=True,
                            constrained_decoding_processor=proc, **kw)
    want = oracle_fm_index_generate(hf_logits_fn(m, enc_ids, enc_mask, K), orc, 4, K, kw["max_length"], vocab,
                                    decoder_start_token_id=2, pad_token_id=1, eos_token_id=eos,
                                    length_penalty=kw["length_penalty"], force_decoding_from=kw.get("force_decoding_from"),
                                    stop_at_count=kw.get("stop_at_count", 0), always_allow_eos=kw.get("always_allow_eos", False),
                                    disable_fm_index=kw.get("disable_fm_index", False))
    assert len(got) == len(want) == 4
    for g, w in zip(got, want):
        te = eos if kw.get("force_decoding_from") else None
        gv, wv = valid_set(g, orc, te), valid_set(w, orc, te)
        assert set(gv) == set(wv)
        for k in gv:
            assert len(gv[k]) == len(wv[k])
            for a, b in zip(sorted(gv[k]), sorted(wv[k])):
                assert abs(a - b) <= 1e-4, (k, a, b)     # north_star tolerance on beam scores
        if kw.get("disable_fm_index"):
            assert [t for _, t in g] == [t for _, t in w]   # no -inf ties without the constraint


def test_unsupported_modes_say_so():
    with pytest.raises(NotImplementedError):
        fm_index_generate(tiny_bart(), None, torch.zeros(1, 2, dtype=torch.long), torch.ones(1, 2, dtype=torch.long),
                          keep_history=False)


@pytest.mark.parametrize("lengths", [(6, 9), (7, 7), (5, 6, 8)])
def test_lockstep_groups_record_what_separate_loops_record(lengths):
    """``constrained_beam_search_groups``: several decodes with their own encoder inputs / end token / forced prefix /
    length as ONE loop over stacked rows (what the searcher does with its body and title decodes), the ones that end first
    leaving the loop (``decoder.narrow``) -- every group's history equals its own separate loop's"""
    from seal_amd.beam_search import constrained_beam_search, constrained_beam_search_groups
    vocab, K = 120, 3
    m = tiny_bart(vocab)
    docs = make_docs(3, 150, vocab, title_sep=7)
    orc = OracleFMIndex()
    orc.initialize(docs)
    torch.manual_seed(5)
    cfgs = [dict(eos_token_id=2, force_decoding_from=None), dict(eos_token_id=7, force_decoding_from=[2]),
            dict(eos_token_id=9, force_decoding_from=[7])][:len(lengths)]
    batches = [3, 2, 2][:len(lengths)]
    enc_ids = [torch.randint(4, vocab, (b, 8)) for b in batches]
    for e in enc_ids:
        e[-1, 6:] = 1
    bias = [torch.randn(b, vocab) for b in batches]

    def proc(c):
        return OracleLogitsProcessor(orc, K, vocab, pad_token_id=1, eos_token_id=c["eos_token_id"], force_decoding_from=c["force_decoding_from"])
    want = []
    for ids, c, T, lb in zip(enc_ids, cfgs, lengths, bias):
        dec = BartStepDecoder(m)
        mask = (ids != 1).long()
        dec.start(dec.encode(ids, mask), mask, K, T)
        dec.logit_bias = lb
        want.append(constrained_beam_search(dec, ids.shape[0], K, T, 2, c["eos_token_id"], proc(c)))
    dec = BartStepDecoder(m)
    ids = torch.cat(enc_ids)
    mask = (ids != 1).long()
    cuts = [sum(batches[:i + 1]) for i in range(len(batches) - 1)]
    dec.start(dec.encode(ids, mask), mask, K, lengths[-1], narrow_plan=cuts)
    dec.logit_bias = torch.cat(bias)
    specs = [dict(batch=b, max_length=T, eos_token_id=c["eos_token_id"], processor=proc(c)) for b, T, c in zip(batches, lengths, cfgs)]
    got = constrained_beam_search_groups(dec, specs, K, 2)
    assert len(got) == len(want)
    for (gs, gf), (ws, wf), T in zip(got, want, lengths):
        assert len(gs) == len(ws) == T - 1
        for (gp, gt, gsc), (wp, wt, wsc) in zip(gs, ws):
            fin = torch.isfinite(wsc) & (wsc > -1e8)
            assert torch.equal(torch.isfinite(gsc), torch.isfinite(wsc))
            assert torch.equal(gp[fin], wp[fin]) and torch.equal(gt[fin], wt[fin])
            assert torch.allclose(gsc[fin], wsc[fin], atol=1e-5)
        assert torch.equal(gf[0], wf[0]) and torch.allclose(gf[1], wf[1], atol=1e-5)


def test_joint_generate_gives_every_job_its_own_stop_at_count(monkeypatch):
    """``fm_index_generate_joint``: a job's ``stop_at_count`` wins over the call's (the searcher passes the body decode's
    value to the body job and 0 to the title / code jobs, as the reference's separate ``fm_index_generate`` calls do,
    retrieval.py:70-83 vs 162-176)"""
    from seal_amd import beam_search as bs
    vocab, K = 120, 2
    m = tiny_bart(vocab)
    seen = []

    def fake_groups(decoder, specs, num_beams, start, device=None, fused=True):
        seen.extend((sp["processor"].stop_at_count, sp["eos_token_id"]) for sp in specs)
        raise StopIteration
    monkeypatch.setattr(bs, "constrained_beam_search_groups", fake_groups)
    orc = OracleFMIndex()
    orc.initialize(make_docs(3, 40, vocab, title_sep=7))
    ids = torch.randint(4, vocab, (4, 6))
    jobs = [dict(batch=2, max_length=5, eos_token_id=None, force_decoding_from=None, stop_at_count=3),
            dict(batch=2, max_length=8, eos_token_id=7, force_decoding_from=[2], stop_at_count=0)]
    with pytest.raises(StopIteration):
        bs.fm_index_generate_joint(m, orc, ids, torch.ones_like(ids), jobs, num_beams=K, stop_at_count=9)
    assert seen == [(3, m.config.eos_token_id), (0, 7)]


def test_beam_loop_promises_identical_beams_only_for_the_first_step_of_a_decoder_that_asks():
    """``constrained_beam_search_groups`` passes ``beams_identical=True`` to ``decoder.step`` for the first position only, and only to a
    decoder that says ``shared_first_step`` (BartStepDecoder's opt-in one-row-per-query first step); the eager decoder ignores the
    promise and the histories are the same either way"""
    from seal_amd.beam_search import constrained_beam_search
    vocab, K, T = 120, 3, 5
    m = tiny_bart(vocab)
    torch.manual_seed(1)
    ids = torch.randint(4, vocab, (2, 7))
    mask = torch.ones_like(ids)
    seen = []

    class Spy(BartStepDecoder):
        def step(self, tokens, beams_identical=False):
            seen.append((self.t, bool(beams_identical), tokens.view(-1, K).eq(tokens.view(-1, K)[:, :1]).all().item()))
            return super().step(tokens, beams_identical=beams_identical)
    out = []
    for asks in (True, False):
        del seen[:]
        dec = Spy(m)
        dec.shared_first_step = asks
        dec.start(dec.encode(ids, mask), mask, K, T)
        out.append(constrained_beam_search(dec, 2, K, T, 2, 2, None))
        assert [s[:2] for s in seen] == [(t, asks and t == 0) for t in range(T - 1)]
        assert seen[0][2] is True                      # the promise holds: every beam starts from the same token
    for (pa, ta, sa), (pb, tb, sb) in zip(out[0][0], out[1][0]):
        assert torch.equal(pa, pb) and torch.equal(ta, tb) and torch.equal(sa, sb)
