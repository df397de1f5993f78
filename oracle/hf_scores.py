"""TEST INFRASTRUCTURE (checker, never the product path): the float half of the parity contract.

What the reference computes for a beam hypothesis is the running sum of HF's fp32 ``log_softmax`` of the model's
next-token logits along the hypothesis' tokens (reference seal/beam_search.py:231-253: cache-assisted forward,
``log_softmax`` at 251, ``+ beam_scores`` at 253/306; recorded by ``BeamSearchScorerWithMemory.process``, 662-668).
This module recomputes those sums with HF's OWN cache-free forward (``model(input_ids, attention_mask,
decoder_input_ids)``, the stock modules, no seal_amd kernel anywhere) by teacher forcing every recorded hypothesis, in
fp32, accumulated in the reference's order (beam score + next log-prob, one position after the other), and compares
them with what the product recorded.  Used by ``bench.py``'s ``parity_check.by_kind.beam_scores`` at BART-large
geometry and by ``tests/test_gpu_score_parity.py``.

The same for rescoring: ``seal_amd.keys.rescore_keys(share_prefixes=False)`` is the reference's one-row-per-key
batching through HF's forward (reference seal/keys.py:64-141); ``compare_rescoring`` holds the prefix-tree path to it.
"""
from typing import Optional

import torch


@torch.no_grad()
def hf_path_logprob_sums(model, enc_ids, enc_mask, seqs, query_of_row, logit_bias: Optional[torch.Tensor] = None,
                         init: Optional[torch.Tensor] = None, chunk: int = 150):
    """``seqs`` [N, L] = decoder start token followed by L-1 hypothesis tokens; row n belongs to query
    ``query_of_row[n]``.  Returns fp32 [N]: ``init[n] + sum_j log_softmax(logits(seqs[n, :j+1]) (+ bias))[seqs[n, j+1]]``
    accumulated left to right in fp32 (the reference adds the beam score and the step's log-prob in fp32 every step)."""
    N, L = seqs.shape
    out = torch.zeros(N, dtype=torch.float32, device=seqs.device) if init is None else init.clone().float()
    if L < 2:
        return out
    for a in range(0, N, chunk):
        b = min(N, a + chunk)
        q = query_of_row[a:b]
        logits = model(input_ids=enc_ids[q], attention_mask=enc_mask[q], decoder_input_ids=seqs[a:b, :-1]).logits.float()
        if logit_bias is not None:
            logits = logits + logit_bias[q][:, None, :]
        logp = torch.log_softmax(logits, dim=-1)
        lp = torch.gather(logp, -1, seqs[a:b, 1:].unsqueeze(-1)).squeeze(-1)          # [n, L-1]
        acc = out[a:b]
        for j in range(L - 1):
            acc = acc + lp[:, j]
        out[a:b] = acc
    return out


@torch.no_grad()
def compare_beam_history(model, enc_ids, enc_mask, steps, final, batch: int, num_beams: int,
                         logit_bias: Optional[torch.Tensor] = None, tol: float = 1e-4):
    """``steps`` / ``final`` = the raw history of ``seal_amd.beam_search.constrained_beam_search``: per decode step
    (prefix ids [B, 2K, t], tokens [B, 2K], sum of log-probs [B, 2K]) and the live beams (ids [B*K, T], scores [B*K]).
    Every recorded score is recomputed through HF's forward.  Hypotheses that descend from the K-1 beams the search
    starts at -1e9 (beam_search.py:214-216) are only checked to be that low on both sides (fp32 swallows the log-probs
    there).  Returns {values, max_abs_err, violations, tol, nonfinite_mismatches, dead_beam_hypotheses}."""
    dev = enc_ids.device
    B, K = batch, num_beams
    vals = viol = nonfinite_bad = dead = 0
    worst = 0.0

    def tally(got, want):
        nonlocal vals, viol, nonfinite_bad, dead, worst
        got, want = got.reshape(-1).float(), want.reshape(-1)
        is_dead = got < -1e8
        dead += int((is_dead & torch.isfinite(got)).sum())
        fin_g, fin_w = torch.isfinite(got), torch.isfinite(want)
        nonfinite_bad += int((fin_g != fin_w).sum())
        m = fin_g & fin_w & ~is_dead
        if bool(m.any()):
            err = (got[m] - want[m]).abs()
            vals += int(m.sum())
            viol += int((err > tol).sum())
            worst = max(worst, float(err.max()))

    for prefix, tokens, scores in steps:
        n = prefix.shape[1]
        seqs = torch.cat([prefix, tokens.unsqueeze(-1)], dim=-1).view(B * n, -1)
        q = torch.arange(B, device=dev).repeat_interleave(n)
        tally(scores, hf_path_logprob_sums(model, enc_ids, enc_mask, seqs, q, logit_bias))
    ids, fscores = final
    q = torch.arange(B, device=dev).repeat_interleave(K)
    tally(fscores, hf_path_logprob_sums(model, enc_ids, enc_mask, ids, q, logit_bias))
    return {"values": vals, "max_abs_err": worst, "violations": viol + nonfinite_bad, "tol": tol,
            "nonfinite_mismatches": nonfinite_bad, "dead_beam_hypotheses": dead,
            "against": "HF BartForConditionalGeneration cache-free fp32 forward, every recorded hypothesis teacher-forced "
                       "(reference beam_search.py:231-253,302-307)"}


def compare_rescoring(model, inputs, keys, tol: float = 1e-4, **kw):
    """the product's prefix-tree rescoring against one HF row per key (reference keys.py:64-141) on the same keys"""
    from seal_amd.keys import rescore_keys
    a = rescore_keys(model, inputs, keys, share_prefixes=True, **kw)
    b = rescore_keys(model, inputs, keys, batch_size=100, share_prefixes=False, **kw)
    vals = viol = 0
    worst = 0.0
    for qa, qb in zip(a, b):
        assert [k for _, k in qa] == [k for _, k in qb]
        for (sa, _), (sb, _) in zip(qa, qb):
            err = abs(sa - sb)
            vals += 1
            worst = max(worst, err)
            viol += int(not err <= tol)
    return {"values": vals, "max_abs_err": worst, "violations": viol, "tol": tol,
            "against": "one decoder row per key through HF's forward (reference keys.py:64-141)"}


@torch.no_grad()
def recompute_beam_history(model, enc_ids, enc_mask, steps, final, batch: int, num_beams: int, logit_bias: Optional[torch.Tensor] = None):
    """the recorded history's scores recomputed through ``model``'s cache-free forward: [(scores [B, 2K]) per step], final [B * K]"""
    dev = enc_ids.device
    B, K = batch, num_beams
    out = []
    for prefix, tokens, scores in steps:
        n = prefix.shape[1]
        seqs = torch.cat([prefix, tokens.unsqueeze(-1)], dim=-1).view(B * n, -1)
        q = torch.arange(B, device=dev).repeat_interleave(n)
        out.append(hf_path_logprob_sums(model, enc_ids, enc_mask, seqs, q, logit_bias).view(B, n))
    ids, _ = final
    q = torch.arange(B, device=dev).repeat_interleave(K)
    return out, hf_path_logprob_sums(model, enc_ids, enc_mask, ids, q, logit_bias)


@torch.no_grad()
def compare_with_fp32_forward(model_lowp, model_fp32, enc_ids, enc_mask, steps, final, batch: int, num_beams: int,
                              logit_bias: Optional[torch.Tensor] = None):
    """A reduced-precision decode (configs[4]: bf16 storage) against the FP32 answer: the recorded scores vs HF's fp32 forward of the same
    weights widened to fp32, HF's own reduced-precision forward vs the same (what the storage type costs by itself), and how many
    recorded candidates would sit at another rank within their step's 2K if the fp32 scores ordered them (a reordered beam is what an
    error of this size can do; a bound on the summed log-probs alone does not say)."""
    B, K = batch, num_beams
    ref_steps, ref_final = recompute_beam_history(model_fp32, enc_ids, enc_mask, steps, final, B, K, logit_bias.float() if logit_bias is not None else None)
    low_steps, low_final = recompute_beam_history(model_lowp, enc_ids, enc_mask, steps, final, B, K, logit_bias)
    worst_prod = worst_hf = 0.0
    ranked = moved = 0
    for (prefix, tokens, scores), ref, low in zip(steps, ref_steps, low_steps):
        got = scores.float()
        m = torch.isfinite(got) & torch.isfinite(ref) & (got > -1e8)
        if bool(m.any()):
            worst_prod = max(worst_prod, float((got[m] - ref[m]).abs().max()))
            worst_hf = max(worst_hf, float((low.float()[m] - ref[m]).abs().max()))
        # rank within the step's 2K candidates of a query, by the recorded scores and by the fp32 ones (stable: ties keep the recorded order)
        g = torch.where(m, got, torch.full_like(got, -1e30))
        r = torch.where(m, ref, torch.full_like(ref, -1e30))
        rg = torch.argsort(torch.argsort(-g, dim=1, stable=True), dim=1)
        rr = torch.argsort(torch.argsort(-r, dim=1, stable=True), dim=1)
        ranked += int(m.sum())
        moved += int(((rg != rr) & m).sum())
    gf = final[1].float()
    mf = torch.isfinite(gf) & torch.isfinite(ref_final) & (gf > -1e8)
    if bool(mf.any()):
        worst_prod = max(worst_prod, float((gf[mf] - ref_final[mf]).abs().max()))
        worst_hf = max(worst_hf, float((low_final.float()[mf] - ref_final[mf]).abs().max()))
    return {"max_abs_err_vs_hf_fp32": worst_prod, "hf_lowp_vs_hf_fp32_max_abs_err": worst_hf, "candidates_ranked": ranked,
            "candidates_at_another_rank_under_fp32_scores": moved, "rank_changed_fraction": (moved / ranked) if ranked else 0.0,
            "against": "HF BartForConditionalGeneration, the same weights widened to fp32, cache-free forward"}
