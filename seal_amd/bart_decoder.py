"""Incremental BART decoder step for beam search on one GPU.

The reference drives ``BartForConditionalGeneration`` through HF 4.13's
``generate`` internals (``prepare_inputs_for_generation``, ``_reorder_cache``,
encoder outputs repeated ``num_beams`` times; reference seal/beam_search.py:
231-238,331-332,517-521).  This module reads the same weights out of the HF
module (so a real SEAL checkpoint drops in) and runs the decoder step with:

* one fused QKV projection per self-attention, one fused KV projection per
  cross-attention (computed once per query, NOT per beam: all beams of a query
  attend to the same encoder states);
* a statically allocated self-attention KV cache ``[layers, 2, rows, heads,
  max_len, head_dim]`` re-ordered in place of HF's tuple-of-tuples cache;
* the ``lm_head`` logits GEMM (the only large dense contraction of the path,
  ``[rows, d] x [d, vocab]``) in fp32 through hipBLASLt/MFMA.

Architecture facts used (BART-large, HF ``BartDecoderLayer``): post-LN,
learned positions with offset 2, ``layernorm_embedding``, GELU FFN, query
scaled by ``head_dim ** -0.5``, logits = ``x @ shared^T + final_logits_bias``.
"""
from typing import Optional

import os
import sys
import torch
import torch.nn.functional as F


import logging


logger = logging.getLogger(__name__)
# the encoder's linear layers through the split GEMM as well (False: HF's own encoder layers, fp32 library GEMMs; the tests flip it)
ENCODER_SPLIT = True

class BartStepDecoder:
    def __init__(self, model):
        self.model = model
        dec = model.model.decoder
        cfg = model.config
        self.d = cfg.d_model
        self.h = cfg.decoder_attention_heads
        self.dh = self.d // self.h
        self.scale = self.dh ** -0.5
        self.embed = dec.embed_tokens
        self.embed_scale = float(getattr(dec.embed_tokens, "embed_scale", 1.0))
        self.pos = dec.embed_positions
        self.pos_offset = int(getattr(dec.embed_positions, "offset", 2))
        self.ln_emb = dec.layernorm_embedding
        self.layers = []
        for l in dec.layers:
            sa, ca = l.self_attn, l.encoder_attn
            self.layers.append(dict(
                qkv_w=torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight], 0).detach(),
                qkv_b=torch.cat([sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias], 0).detach(),
                so=sa.out_proj, ln1=l.self_attn_layer_norm,
                cq=ca.q_proj,
                ckv_w=torch.cat([ca.k_proj.weight, ca.v_proj.weight], 0).detach(),
                ckv_b=torch.cat([ca.k_proj.bias, ca.v_proj.bias], 0).detach(),
                co=ca.out_proj, ln2=l.encoder_attn_layer_norm,
                fc1=l.fc1, fc2=l.fc2, ln3=l.final_layer_norm, act=l.activation_fn))
        self.lm_w = model.lm_head.weight
        self.lm_b = model.final_logits_bias
        self.batch = self.beams = self.rows = 0
        # optional additive bias on the next-token logits, [batch, vocab], shared by the beams of a query
        # (bench.py shapes a random-init model's preferences towards corpus n-grams with it)
        self.logit_bias: Optional[torch.Tensor] = None

    @staticmethod
    def _nn(dtype):
        """the fused kernels of include/sealnn.h for a storage dtype: ``nn.self_attn_step`` = ``sealnn_self_attn_step[_bf16]``"""
        from ._lib import lib
        L_ = lib()
        suffix = "" if dtype == torch.float32 else "_bf16"

        class _NN:
            def __getattr__(self, name):
                return getattr(L_, "sealnn_" + name + suffix)
        return _NN()

    FUSED_DTYPES = (torch.float32, torch.bfloat16)

    # fp32 linear layers of the fused paths on the fp16 matrix cores (seal_amd/split_gemm.py; SEAL_SPLIT_GEMM=0: plain fp32 GEMMs)
    split_gemm = None

    def _lin(self, x: torch.Tensor, w: torch.Tensor, b, defer: bool = False, slabs_ok: bool = False, pairs: bool = False):
        """``F.linear(x, w, b)`` of an fp32 [rows, K] activation on the GPU -- through the split GEMM when that is switched on.
        ``defer``: the caller hands the result to a sealnn_*_acc kernel, which applies the epilogue of a split product itself
        (``split_gemm.Deferred``; a product that does not go through the split comes back finished)"""
        if self.split_gemm is None:
            from . import split_gemm
            BartStepDecoder.split_gemm = split_gemm.SplitLinears() if split_gemm.ENABLED else False
        if self.split_gemm and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
            return self.split_gemm(x, w, b, defer, slabs_ok, pairs)
        if x.is_cuda:
            from . import split_gemm
            split_gemm.LIBRARY_GEMMS[0] += 1
        return F.linear(x, w, b)

    def _mod(self, x: torch.Tensor, m, defer: bool = False, slabs_ok: bool = False, pairs: bool = False):
        return self._lin(x, m.weight, m.bias, defer, slabs_ok, pairs)

    def _pairs_ok(self, rows: int, ffn: int, w_qkv, w_fc1, w_fc2) -> bool:
        """may the planes of a layer stack at this height be hi / lo PAIRS?  Only the hand-written kernel reads them, so every product must have a
        PAIRS configuration and be a split product at all (split_gemm.PAIRS, HAND_CONFIGS_PAIRS)"""
        from . import split_gemm
        return bool(self.split_gemm and split_gemm.PAIRS and split_gemm.HAND_GEMM and split_gemm.DEFER_EPILOGUE and self.d % 32 == 0 and ffn % 32 == 0 and
                    all(split_gemm.hand_config(rows, n, 3 * k, True) is not None for n, k in ((3 * self.d, self.d), (self.d, self.d), (ffn, self.d), (self.d, ffn))) and
                    all(self.split_gemm.wants(w, rows) for w in (w_qkv, w_fc1, w_fc2)))

    # -- the consumers of a product: the finished tensor through the plain kernel, a Deferred one through its _acc twin --
    def _add_ln(self, L_, stream, res, y, ln, rows, planes, pairs=False):
        """LayerNorm(res + y) -> (fp32, its split planes or None); ``y``: a tensor or a ``split_gemm.Deferred``; ``pairs``: the planes as
        hi / lo pairs per 32 columns ([rows, 2d]: the operand of the hand-written product's PAIRS form) -- ``y`` is then a ``Deferred``"""
        from . import split_gemm
        from ._lib import check, lib
        out = torch.empty_like(res)
        p = torch.empty(rows, (2 if pairs else 3) * self.d, dtype=torch.float16, device=res.device) if planes else None
        flag = split_gemm._flag(res.device).data_ptr() if planes else None
        if pairs:
            if not planes:
                raise RuntimeError("BartStepDecoder: pair planes asked for where no planes are written")
            if not isinstance(y, split_gemm.Deferred):
                # (a finished addend -- a product too small for the split -- as the trivial deferred one: 1.0 * y + 0)
                zero = self.__dict__.get("_zero_bias")
                if zero is None or zero.device != y.device:
                    zero = self._zero_bias = torch.zeros(self.d, dtype=torch.float32, device=y.device)
                y = split_gemm.Deferred(y.contiguous(), zero, 1.0)
            check(lib().sealnn_add_layernorm_acc_slabs_pairs(stream, res.data_ptr(), y.acc.data_ptr(), y.slabs, y.acc.stride(0) if y.slabs > 1 else 0,
                                                             y.bias.data_ptr(), float(y.alpha), ln.weight.data_ptr(), ln.bias.data_ptr(), rows, self.d,
                                                             float(ln.eps), out.data_ptr(), p.data_ptr(), flag))
        elif isinstance(y, split_gemm.Deferred) and y.slabs > 1:
            check(lib().sealnn_add_layernorm_acc_slabs(stream, res.data_ptr(), y.acc.data_ptr(), y.slabs, y.acc.stride(0), y.bias.data_ptr(), float(y.alpha),
                                                       ln.weight.data_ptr(), ln.bias.data_ptr(), rows, self.d, float(ln.eps), out.data_ptr(),
                                                       p.data_ptr() if planes else None, flag))
        elif isinstance(y, split_gemm.Deferred):
            check(lib().sealnn_add_layernorm_acc(stream, res.data_ptr(), y.acc.data_ptr(), y.bias.data_ptr(), float(y.alpha), ln.weight.data_ptr(),
                                                 ln.bias.data_ptr(), rows, self.d, float(ln.eps), out.data_ptr(),
                                                 p.data_ptr() if planes else None, flag))
        elif planes:
            check(lib().sealnn_add_layernorm_planes(stream, res.data_ptr(), y.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                                    rows, self.d, float(ln.eps), out.data_ptr(), p.data_ptr(), flag))
        else:
            check(L_.add_layernorm(stream, res.data_ptr(), y.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                   rows, self.d, float(ln.eps), out.data_ptr()))
        return out, p

    # -- split GEMM with the operand planes written by the kernels that produce the activations (split_gemm.FUSED) --
    def _planes_on(self, x: torch.Tensor) -> bool:
        from . import split_gemm
        if self.split_gemm is None:
            BartStepDecoder.split_gemm = split_gemm.SplitLinears() if split_gemm.ENABLED else False
        # (no product of fewer than MIN_ROWS rows takes the split: nobody would read the planes of such an activation)
        return bool(self.split_gemm) and split_gemm.FUSED and x.is_cuda and x.dtype == torch.float32 and x.shape[0] >= split_gemm.MIN_ROWS

    def _lin_p(self, x: torch.Tensor, xp, w: torch.Tensor, b, defer: bool = False, slabs_ok: bool = False):
        """``F.linear(x, w, b)`` where ``xp`` (or None) holds x's split planes already (``defer``: see ``_lin``; ``slabs_ok``: the consumer
        adds split-K slabs, so the hand-written kernel may serve the product)"""
        if xp is not None and xp.shape[1] == 2 * x.shape[1]:
            # pair planes (a hand step's): only the hand-written kernel reads them; a product it has no configuration for splits x itself
            from . import split_gemm
            if split_gemm.hand_config(x.shape[0], w.shape[0], 3 * w.shape[1], True) is None:
                xp = None
        if xp is not None and self.split_gemm and self.split_gemm.wants(w, x.shape[0]):
            return self.split_gemm.from_planes(xp, w, b, defer, slabs_ok=slabs_ok)
        return self._lin(x, w, b, defer)

    def _planes_of(self, x: torch.Tensor, pairs: bool = False) -> torch.Tensor:
        from . import split_gemm
        from ._lib import check, lib
        p = torch.empty(x.shape[0], (2 if pairs else 3) * x.shape[1], dtype=torch.float16, device=x.device)
        fn = lib().sealnn_split_planes_pairs if pairs else lib().sealnn_split_planes
        check(fn(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), x.shape[0], x.shape[1], p.data_ptr(), split_gemm._flag(x.device).data_ptr()))
        return p

    def _ffn(self, x: torch.Tensor, xp, L, defer: bool = False, hand: bool = False, pairs: bool = False):
        """fc2(gelu(fc1(x))): with planes, gelu's output exists as fc2's operand only (``defer``: see ``_lin``; fc1's own epilogue is
        always gelu's to apply; ``hand``: fc1 may run in the hand-written kernel too, split-K slabs and all -- GELU adds them)"""
        w2 = L["fc2"].weight
        # (the fused kernel computes the erf form: nn.GELU(approximate="tanh") shares the class name and must not take it)
        erf_gelu = (getattr(L["act"], "__class__", type(None)).__name__ in ("GELUActivation", "GELU")
                    and getattr(L["act"], "approximate", "none") == "none")
        if xp is not None and self.split_gemm.wants(w2, x.shape[0]) and erf_gelu:
            from . import split_gemm
            from ._lib import check, lib
            h = self._lin_p(x, xp, L["fc1"].weight, L["fc1"].bias, defer=True, slabs_ok=hand)
            acc = h.acc if isinstance(h, split_gemm.Deferred) else h
            rows, d1 = acc.shape[-2], acc.shape[-1]
            hp = torch.empty(rows, (2 if pairs else 3) * d1, dtype=torch.float16, device=acc.device)
            stream = torch.cuda.current_stream(acc.device).cuda_stream
            flag = split_gemm._flag(acc.device).data_ptr()
            if pairs:                                                      # (fc2's operand as hi / lo pairs: the hand-written product's PAIRS form)
                if not isinstance(h, split_gemm.Deferred):
                    raise RuntimeError("BartStepDecoder: pair planes are written behind a deferred product only")
                check(lib().sealnn_gelu_planes_acc_slabs_pairs(stream, acc.data_ptr(), h.slabs, acc.stride(0) if h.slabs > 1 else 0, h.bias.data_ptr(),
                                                               float(h.alpha), rows, d1, hp.data_ptr(), flag))
            elif isinstance(h, split_gemm.Deferred) and h.slabs > 1:        # (fc1 as a split-K product: GELU adds the slabs as it reads them)
                check(lib().sealnn_gelu_planes_acc_slabs(stream, acc.data_ptr(), h.slabs, acc.stride(0), h.bias.data_ptr(), float(h.alpha), rows, d1,
                                                         hp.data_ptr(), flag))
            elif isinstance(h, split_gemm.Deferred):
                check(lib().sealnn_gelu_planes_acc(stream, acc.data_ptr(), h.bias.data_ptr(), float(h.alpha), rows, d1, hp.data_ptr(), flag))
            else:
                check(lib().sealnn_gelu_planes(stream, acc.data_ptr(), rows, d1, hp.data_ptr(), flag))
            return self.split_gemm.from_planes(hp, w2, L["fc2"].bias, defer, slabs_ok=defer)      # (its consumer, add + LayerNorm, adds split-K slabs)
        h = self._lin_p(x, xp, L["fc1"].weight, L["fc1"].bias)
        return self._mod(L["act"](h), L["fc2"], defer)

    @torch.no_grad()
    def encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """``model.model.encoder(...).last_hidden_state`` through the encoder's own modules, with the padding mask built
        here: HF's mask helper asks the device whether the mask is all ones (``padding_mask.all()``, a device -> host
        read-back: measured 12 ms of blocked host per call once decodes are queued ahead), which stops the host from
        enqueueing anything behind the encoder.  Same layers, same attention function, the mask always passed."""
        enc = self.model.model.encoder
        impl = getattr(self.model.config, "_attn_implementation", "sdpa")
        if impl not in ("sdpa", "eager") or self.model.training:
            return enc(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        x = enc.embed_tokens(input_ids)
        x = x + enc.embed_positions(x[:, :, -1]).to(x.device)
        x = enc.layernorm_embedding(x)
        B, S = input_ids.shape
        keep = attention_mask.to(torch.bool)[:, None, None, :].expand(B, 1, S, S)
        if impl == "sdpa":
            mask = keep
        else:
            mask = torch.zeros(B, 1, S, S, dtype=x.dtype, device=x.device).masked_fill_(~keep, torch.finfo(x.dtype).min)
        if (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == self.d and self._planes_on(x.view(B * S, -1)) and ENCODER_SPLIT
                and all(getattr(l.activation_fn, "__class__", type(None)).__name__ in ("GELUActivation", "GELU")
                        and getattr(l.activation_fn, "approximate", "none") == "none" for l in enc.layers)):
            # The encoder's linear layers through the split GEMM too (round 5: they were 72 fp32 library launches of ~41 us per batch of 60
            # inputs, 5 ms of a 63 ms step): the layer written out with this module's `_lin` (q, k, v as one product), torch's fused attention
            # and LayerNorm in between -- the same arithmetic as BartEncoderLayer.forward (post-LN), products at fp32 accuracy (split_gemm.py).
            # (residual + LayerNorm and GELU through the decoder's fused kernels: they apply the products' epilogues as they read them and
            #  write the next product's operand planes -- no bias copy, no split pass, no separate GELU between the products)
            x2 = x.reshape(B * S, -1).contiguous()
            rows = B * S
            H, dh = enc.layers[0].self_attn.num_heads, enc.layers[0].self_attn.head_dim
            L_ = self._nn(x2.dtype)
            stream = torch.cuda.current_stream(x2.device).cuda_stream
            # (round 6: the planes as hi / lo PAIRS where every product of the stack has a hand-written configuration at this height -- the library's 37 / 44 /
            #  68 us for qkv / fc1 / fc2 at 1 280 rows become 26 / 33 / 44: split_gemm.HAND_CONFIGS_PAIRS)
            l0 = enc.layers[0]
            pairs = self._pairs_ok(rows, int(l0.fc1.weight.shape[0]), self._encoder_qkv(l0)[0], l0.fc1.weight, l0.fc2.weight)
            xp = self._planes_of(x2, pairs)
            for layer in enc.layers:
                sa = layer.self_attn
                w, b = self._encoder_qkv(layer)
                qkv = self._lin_p(x2, xp, w, b).view(B, S, 3, H, dh)
                q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
                a = F.scaled_dot_product_attention(q, k, v, attn_mask=keep, scale=float(sa.scaling))
                a = a.transpose(1, 2).reshape(rows, H * dh)
                if pairs:       # (the d x d product is below the split's own threshold at these heights; as pairs it is the hand-written kernel's)
                    y = self.split_gemm._of(sa.out_proj.weight, sa.out_proj.bias)(a, True, True, True)
                else:
                    y = self._lin(a, sa.out_proj.weight, sa.out_proj.bias, defer=True, slabs_ok=True)
                x2, xp = self._add_ln(L_, stream, x2, y, layer.self_attn_layer_norm, rows, True, pairs)
                ffn = self._ffn(x2, xp, {"fc1": layer.fc1, "fc2": layer.fc2, "act": layer.activation_fn}, defer=True, hand=True, pairs=pairs)
                x2, xp = self._add_ln(L_, stream, x2, ffn, layer.final_layer_norm, rows, True, pairs)
            return x2.view(B, S, -1)
        for layer in enc.layers:
            x = layer(x, mask)
        return x

    def _ckv(self, enc_hidden: torch.Tensor, L) -> torch.Tensor:
        """the cross-attention K / V projection of the encoder states, one product per layer ([B, S, d] -> [B, S, 2d]); through the split GEMM
        where that pays (fp32 on the GPU, enough rows), F.linear otherwise"""
        if enc_hidden.is_cuda and enc_hidden.dtype == torch.float32 and enc_hidden.dim() == 3 and ENCODER_SPLIT:
            B, S, d = enc_hidden.shape
            return self._lin(enc_hidden.reshape(B * S, d), L["ckv_w"], L["ckv_b"], pairs=True).view(B, S, -1)      # (pairs where this height has a configuration)
        return F.linear(enc_hidden, L["ckv_w"], L["ckv_b"])

    def _encoder_qkv(self, layer):
        """(weight [3d, d], bias [3d]) of an encoder layer's q / k / v projections as one product, made once"""
        # validated by the six source parameters' modification counters: a checkpoint loaded in place keeps every address
        from .split_gemm import tensor_version
        cache = self.__dict__.setdefault("_enc_qkv", {})
        sa = layer.self_attn
        src = (sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight, sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias)
        stamp = tuple((t.data_ptr(), tensor_version(t)) for t in src)
        hit = cache.get(id(layer))
        if hit is None or hit[0] != stamp:
            hit = cache[id(layer)] = (stamp, torch.cat(src[:3], 0).detach().contiguous(), torch.cat(src[3:], 0).detach().contiguous())
        return hit[1], hit[2]

    # ------------------------------------------------------------------
    # static-shape state: one set of buffers (and one hipGraph) per
    # (batch, beams, padded encoder length, max decode length).  The decode position lives in a
    # device tensor, the self-attention runs over the whole (masked) cache, so every step of a
    # decode -- and every later decode of the same shape -- replays the same graph instead of
    # launching ~300 small kernels from Python.
    # ------------------------------------------------------------------
    use_graph = True

    class _Static:
        pass

    def _static_for(self, B, K, S_pad, T, dtype, dev):
        key = (B, K, S_pad, T, dtype, str(dev))
        cache = self.__dict__.setdefault("_static_cache", {})
        st = cache.get(key)
        if st is None:
            st = self._Static()
            R, H, dh, nl = B * K, self.h, self.dh, len(self.layers)
            st.tokens = torch.zeros(R, dtype=torch.long, device=dev)
            st.t = torch.zeros(1, dtype=torch.long, device=dev)
            st.kv = torch.zeros(nl, 2, R, H, T, dh, dtype=dtype, device=dev)
            # fused path: history of row r at position p = cache slot (anc[p, r], p); beams are re-ranked
            # by permuting the columns of this table, the cache itself never moves
            st.anc = torch.arange(R, dtype=torch.int32, device=dev).repeat(T, 1).contiguous()
            st.fused = False
            st.library_free = False
            st.ck = torch.zeros(nl, B, H, dh, S_pad, dtype=dtype, device=dev)
            st.cv = torch.zeros(nl, B, H, S_pad, dh, dtype=dtype, device=dev)
            st.cbias = torch.zeros(B, 1, 1, S_pad, dtype=dtype, device=dev)
            st.pos_idx = torch.arange(T, device=dev).view(1, 1, 1, T)
            st.logits = None
            st.graph = None
            # the first step of a beam search, where the K beams of a query are the same row (_step_static_first)
            st.first_graph = None
            st.first_logits = None
            st.beam0 = (torch.arange(R, dtype=torch.int32, device=dev) // K * K).contiguous()
            st.shape = (B, K, S_pad, T)
            st.dropped = 0
            st.tails = {}       # dropped leading queries -> the static state of the rest (views of these buffers; narrow())
            cache[key] = st
        return st

    # ------------------------------------------------------------------
    # teacher-forced forward for rescoring (reference keys.py:64-141 runs HF's full model per chunk of
    # keys, re-projecting the encoder states for every row): encoder K/V are projected once per QUERY
    # and every decoder row points at its query.
    # ------------------------------------------------------------------
    def can_teacher_force(self, enc_hidden: torch.Tensor, T: int) -> bool:
        return (self.use_fused_kernels and enc_hidden.is_cuda and enc_hidden.dtype in self.FUSED_DTYPES and self.dh == 64
                and T <= 17 and enc_hidden.shape[1] <= 64)

    @torch.no_grad()
    def teacher_prepare(self, enc_hidden: torch.Tensor, attention_mask: torch.Tensor):
        """per-query cross-attention K/V ([layers][B, H, 64, S] / [B, H, S, 64]) and additive bias [B, S]"""
        B, S, _ = enc_hidden.shape
        cross = []
        for L in self.layers:
            kv = self._ckv(enc_hidden, L).view(B, S, 2, self.h, self.dh)
            cross.append((kv[:, :, 0].permute(0, 2, 3, 1).contiguous(), kv[:, :, 1].permute(0, 2, 1, 3).contiguous()))
        bias = torch.zeros(B, S, dtype=enc_hidden.dtype, device=enc_hidden.device)
        bias.masked_fill_(attention_mask == 0, torch.finfo(enc_hidden.dtype).min)
        return cross, bias, S

    @torch.no_grad()
    def teacher_logits(self, dec_ids: torch.Tensor, qidx: torch.Tensor, prepared) -> torch.Tensor:
        """decoder input ids [N, T] (row n belongs to query qidx[n]) -> logits [N, T, vocab]"""
        from ._lib import check
        cross, bias, S = prepared
        L_ = self._nn(bias.dtype)
        N, T = dec_ids.shape
        dev = dec_ids.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        x = self.embed(dec_ids) + self.pos.weight[self.pos_offset:self.pos_offset + T]
        x = self.ln_emb(x).view(N * T, self.d)
        row_batch = qidx.to(torch.int32).repeat_interleave(T).contiguous()

        def add_ln(res, y, ln):
            out = torch.empty_like(res)
            check(L_.add_layernorm(stream, res.data_ptr(), y.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                   N * T, self.d, float(ln.eps), out.data_ptr()))
            return out
        for li, L in enumerate(self.layers):
            qkv = F.linear(x, L["qkv_w"], L["qkv_b"])
            a = torch.empty(N * T, self.d, dtype=x.dtype, device=dev)
            check(L_.causal_self_attn(stream, qkv.data_ptr(), N, T, self.h, float(self.scale), a.data_ptr()))
            x = add_ln(x, L["so"](a), L["ln1"])
            q = L["cq"](x)
            c = torch.empty(N * T, self.d, dtype=x.dtype, device=dev)
            ck, cv = cross[li]
            # the T positions of a sequence attend the same query: one staging of its K/V per (sequence, head)
            check(L_.cross_attn_runs(stream, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), row_batch.data_ptr(),
                                     N * T, T, self.h, S, float(self.scale), c.data_ptr()))
            x = add_ln(x, L["co"](c), L["ln2"])
            x = add_ln(x, L["fc2"](L["act"](L["fc1"](x))), L["ln3"])
        return F.linear(x, self.lm_w, self.lm_b.view(-1)).view(N, T, -1)

    @torch.no_grad()
    def lm_head(self, x: torch.Tensor) -> torch.Tensor:
        """decoder states [N, d] -> next-token logits [N, vocab] (``x @ shared^T + final_logits_bias``)"""
        return self._lin(x, self.lm_w, self.lm_b.view(-1))

    def tree_logits(self, tok: torch.Tensor, depth: torch.Tensor, anc: torch.Tensor, qidx: torch.Tensor, enc_hidden: torch.Tensor,
                    attention_mask: torch.Tensor, prepared=None, hidden_only: bool = False) -> torch.Tensor:
        """Teacher forcing over a prefix tree: node i is one decoder position -- input token ``tok[i]`` at position
        ``depth[i]`` of the distinct prefix whose nodes are ``anc[i, :depth[i] + 1]`` (root first, itself last, -1 beyond) --
        of query ``qidx[i]``.  Returns the logits after every node, [N, vocab] (``hidden_only``: the decoder states [N, d]
        before the output projection, for a caller that projects them a slice at a time): exactly what row-per-key teacher forcing
        (reference keys.py:64-141) computes at that position of any row that starts with that prefix, each computed once.
        ``prepared`` = ``teacher_prepare(...)`` runs the fused sealnn_* kernels; otherwise plain torch ops (any device / dtype)."""
        N = tok.numel()
        dev = tok.device
        x = self.embed(tok) + self.pos.weight[self.pos_offset + depth]
        x = self.ln_emb(x)
        if prepared is not None:
            from ._lib import check
            cross, bias, S = prepared
            L_ = self._nn(bias.dtype)
            stream = torch.cuda.current_stream(dev).cuda_stream
            anc32 = anc.to(torch.int32).contiguous()
            row_batch = qidx.to(torch.int32).contiguous()
            A = anc32.shape[1]

            from . import split_gemm
            from ._lib import lib
            planes = self._planes_on(x)

            L0 = self.layers[0]
            # (round 6: hi / lo PAIRS planes and the hand-written kernel for every product of the forest where its height has configurations)
            pairs = bool(planes) and self._pairs_ok(N, int(self.model.config.decoder_ffn_dim), L0["qkv_w"], L0["fc1"].weight, L0["fc2"].weight)

            def add_ln(res, y, ln):
                return self._add_ln(L_, stream, res, y, ln, N, planes, pairs)
            xp = self._planes_of(x, pairs) if planes else None
            for li, L in enumerate(self.layers):
                qkv = self._lin_p(x, xp, L["qkv_w"], L["qkv_b"], defer=True)
                if isinstance(qkv, split_gemm.Deferred) and qkv.slabs > 1:       # (the tree kernel reads ONE slab)
                    qkv = split_gemm.Deferred(qkv.acc.sum(0), qkv.bias, qkv.alpha)
                a = torch.empty(N, self.d, dtype=x.dtype, device=dev)
                if isinstance(qkv, split_gemm.Deferred):
                    check(lib().sealnn_tree_self_attn_acc(stream, qkv.acc.data_ptr(), qkv.bias.data_ptr(), float(qkv.alpha), anc32.data_ptr(), N, A,
                                                          self.h, float(self.scale), a.data_ptr()))
                else:
                    check(L_.tree_self_attn(stream, qkv.data_ptr(), anc32.data_ptr(), N, A, self.h, float(self.scale), a.data_ptr()))
                x, xp = add_ln(x, self._mod(a, L["so"], defer=True, slabs_ok=True, pairs=pairs), L["ln1"])
                q = self._lin_p(x, xp, L["cq"].weight, L["cq"].bias)
                c = torch.empty(N, self.d, dtype=x.dtype, device=dev)
                ck, cv = cross[li]
                # (a query's nodes are consecutive: sixteen rows per workgroup share its K / V through LDS; a row of another query than its
                #  run's first reads its own from memory -- the same arithmetic as sealnn_cross_attn_rows on either path)
                check(L_.cross_attn_runs(stream, q.data_ptr(), ck.data_ptr(), cv.data_ptr(), bias.data_ptr(), row_batch.data_ptr(),
                                         N, 16, self.h, S, float(self.scale), c.data_ptr()))
                x, xp = add_ln(x, self._mod(c, L["co"], defer=True, slabs_ok=True, pairs=pairs), L["ln2"])
                x, xp = add_ln(x, self._ffn(x, xp, L, defer=True, hand=pairs, pairs=pairs), L["ln3"])
            if hidden_only:
                return x            # (the caller projects slices of x: lm_head -> _lin -> the split kernel per slice)
            return self._lin_p(x, xp, self.lm_w, self.lm_b.view(-1))
        # plain torch ops: the ancestors' keys / values gathered per node, the encoder's per query
        B, S, _ = enc_hidden.shape
        H, dh = self.h, self.dh
        anc_ok = anc >= 0                                   # [N, A]
        anc_ix = anc.clamp(min=0)
        neg = torch.finfo(x.dtype).min
        self_bias = torch.zeros(anc.shape, dtype=x.dtype, device=dev).masked_fill_(~anc_ok, neg)[:, None, :]      # [N, 1, A]
        cross_bias = torch.zeros(B, S, dtype=x.dtype, device=dev).masked_fill_(attention_mask == 0, neg)[qidx][:, None, :]   # [N, 1, S]
        for L in self.layers:
            qkv = F.linear(x, L["qkv_w"], L["qkv_b"]).view(N, 3, H, dh)
            q, k, v = qkv[:, 0] * self.scale, qkv[:, 1], qkv[:, 2]
            ka, va = k[anc_ix], v[anc_ix]                   # [N, A, H, dh]
            w = torch.softmax(torch.einsum("nhd,nahd->nha", q, ka) + self_bias, dim=-1)
            a = torch.einsum("nha,nahd->nhd", w, va).reshape(N, self.d)
            x = L["ln1"](x + L["so"](a))
            kv = self._ckv(enc_hidden, L).view(B, S, 2, H, dh)
            cq = (L["cq"](x) * self.scale).view(N, H, dh)
            w = torch.softmax(torch.einsum("nhd,nshd->nhs", cq, kv[:, :, 0][qidx]) + cross_bias, dim=-1)
            c = torch.einsum("nhs,nshd->nhd", w, kv[:, :, 1][qidx]).reshape(N, self.d)
            x = L["ln2"](x + L["co"](c))
            x = L["ln3"](x + L["fc2"](L["act"](L["fc1"](x))))
        return x if hidden_only else self.lm_head(x)

    TREE_NODE_BUCKET = 256       # tree_hidden_graph: node count rounded up to a multiple (round 6: 1 024 before; the neighbours of the first bucket are captured with it)
    tree_capture_neighbours = True

    @torch.no_grad()
    def tree_hidden_graph(self, tok, depth, anc, qidx, enc_hidden, attention_mask):
        """``tree_logits(..., hidden_only=True)`` through the fused kernels as ONE hipGraph replay (the forward over a prefix tree
        is ~160 launches that the host would otherwise issue one by one, 10 ms per batch of the searcher).  On by default since
        round 4 (``keys.RESCORE_GRAPH = False``: launch by launch): with the split GEMM the search is bound by its one python thread,
        and the launches saved are worth more than the padded nodes cost -- 217 -> 236 queries/s (profiles/r4_soak_ab.txt; round 3,
        GPU-bound on fp32 GEMMs, measured the opposite: 256 -> 247).
        Shapes are made static: the node count is rounded up to ``TREE_NODE_BUCKET`` (the rows behind the real nodes keep whatever
        valid nodes an earlier call left there; their results are ignored), the encoder length to 16, the ancestor table to 17
        columns; one graph per (nodes, encoder length, queries) bucket, captured on first use, cross-attention K/V of the queries
        computed inside.  Returns [N, d] (a view of the graph's output buffer, valid until the next call), or None when the fused
        path does not apply."""
        N = tok.numel()
        Bq, S, d = enc_hidden.shape
        A = anc.shape[1]
        if not (self.use_graph and enc_hidden.is_cuda and self.can_teacher_force(enc_hidden, A) and N > 0):
            return None
        # (the bucket grows with the forest: 1/16 of the power of two at or above N -- 256 rows for the ~3 100 nodes of a beam-15 batch, 512 for
        #  the ~6 200 of beam 30 -- so that the forests of one workload fall into three or four buckets whatever their size)
        bucket = max(64, int(self.TREE_NODE_BUCKET))
        while bucket * 16 < N:
            bucket *= 2
        Np = (N + bucket - 1) // bucket * bucket
        Sp = max(16, (S + 15) // 16 * 16)
        if Sp > 64:
            return None
        dev, dt = enc_hidden.device, enc_hidden.dtype
        cache = self.__dict__.setdefault("_static_cache", {})
        key = ("tree", Np, Sp, Bq, dt, str(dev))
        st = cache.get(key)
        # The caller (rescoring) runs under torch.inference_mode(); buffers and graph are made OUTSIDE it: a capture updates the
        # CUDA generator's graph-state tensors in place, and if those were first created as inference tensors every later capture
        # outside inference mode (the decoder's, under no_grad) raises "Inplace update to inference tensor outside InferenceMode".
        with torch.inference_mode(False), torch.no_grad():
            def make(n_rows):
                z = self._Static()
                z.tok = torch.full((n_rows,), int(self.model.config.pad_token_id), dtype=torch.long, device=dev)
                z.depth = torch.zeros(n_rows, dtype=torch.long, device=dev)
                z.anc = torch.full((n_rows, 17), -1, dtype=torch.long, device=dev)
                z.anc[:, 0] = torch.arange(n_rows, device=dev)                  # every row a root of its own until a real node lands on it
                z.qidx = torch.zeros(n_rows, dtype=torch.long, device=dev)
                z.enc = torch.zeros(Bq, Sp, d, dtype=dt, device=dev)
                z.mask = torch.zeros(Bq, Sp, dtype=torch.uint8, device=dev)
                z.hidden = None
                z.graph = None
                return z

            def capture(z):
                def forward():
                    prepared = self.teacher_prepare(z.enc, z.mask)
                    return self.tree_logits(z.tok, z.depth, z.anc, z.qidx, z.enc, z.mask, prepared, True)
                cur = torch.cuda.current_stream(dev)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(2):
                        forward()
                cur.wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # its own capture stream: torch keeps ONE BLAS workspace per (handle, stream) and captures on a process-wide
                # default stream otherwise -- this graph would then share the decode graphs' workspace (stream-K partial tiles
                # and flags of the library's GEMMs) while it replays on another stream beside them
                cap = self.__dict__.get("_tree_capture_stream")
                if cap is None:
                    cap = self._tree_capture_stream = torch.cuda.Stream(device=dev)
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    z.hidden = forward()
                z.graph = g
            first_of_family = st is None and not any(k[0] == "tree" and k[2:] == key[2:] for k in cache if isinstance(k, tuple))
            if st is None:
                st = cache[key] = make(Np)
            st.tok[:N] = tok
            st.depth[:N] = depth
            st.anc[:N, :A] = anc
            if A < 17:
                st.anc[:N, A:] = -1
            st.qidx[:N] = qidx
            st.enc[:, :S] = enc_hidden
            st.mask.zero_()
            st.mask[:, :S] = attention_mask.to(torch.uint8)
            if st.graph is None:
                capture(st)
                if first_of_family and self.tree_capture_neighbours:
                    # The first forest of a searcher decides where its node counts live (a batch of the bench: 2 900 .. 3 400 nodes); the buckets
                    # next to it are captured right away, behind this one -- a capture costs ~0.1 s of a stalled search whenever it happens,
                    # and it is better spent while the searcher warms up than in the middle of a run.  (What this buys: buckets of 256 rows
                    # instead of 1 024 -- a forest of 3 130 nodes no longer runs as 4 096 rows, a quarter of the rescoring's GEMM work.)
                    for n_rows in (Np + bucket, Np - bucket, Np + 2 * bucket, Np - 2 * bucket):
                        k2 = ("tree", n_rows) + key[2:]
                        if n_rows >= bucket and k2 not in cache:
                            z = cache[k2] = make(n_rows)
                            z.enc.copy_(st.enc)
                            z.mask.copy_(st.mask)
                            capture(z)
        st.graph.replay()
        return st.hidden[:N]

    use_fused_kernels = True      # include/sealnn.h: self-attn / cross-attn / add+LayerNorm as single HIP kernels

    def _step_static(self, st):
        B, K, S_pad, T = st.shape
        R, H, dh = B * K, self.h, self.dh
        x = self.embed(st.tokens)
        x = x + self.pos.weight.index_select(0, st.t + self.pos_offset)
        x = self.ln_emb(x)
        fused = (self.use_fused_kernels and x.dtype in self.FUSED_DTYPES and dh == 64 and T <= 17 and S_pad <= 64 and x.is_cuda)
        st.fused = fused
        if not fused and x.is_cuda and self.use_fused_kernels and not self.__dict__.get("_fallback_logged"):
            # the limits of the fused sealnn_* step kernels, said once instead of silently running ~6x more launches
            self._fallback_logged = True
            logger.warning("BartStepDecoder: decode of shape (batch %d, beams %d, encoder length %d, %d positions, %s, head_dim %d) runs on the "
                           "torch-op path: the fused step kernels need fp32 or bf16, head_dim 64, <= 17 decoder positions and <= 64 encoder tokens",
                           B, K, S_pad, T, str(x.dtype).replace("torch.", ""), dh)
        if fused:
            from ._lib import check, lib
            L_ = self._nn(x.dtype)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            cbias = st.cbias.view(B, S_pad)

            from . import split_gemm
            planes = self._planes_on(x)

            def add_ln(res, y, ln):
                """LayerNorm(res + y) -> (fp32, its split planes or None)"""
                return self._add_ln(L_, stream, res, y, ln, R, planes, pairs)
            # The hand-written product (sealnn_hgemm_nt) between kernels of this repository: at the decode step's heights the d x d projections
            # (self-attention output, cross-attention query and output) run as 4 split-K slabs that the consumer adds as it reads them, and
            # the attention kernels hand their result over as the next projection's split planes -- the library's fp32 GEMM of 17 us (600 rows)
            # / 12 us (300) becomes 11 / 7 us (profiles/r5_hgemm_probe.txt), with no pass over the activations in between.
            hand = bool(planes and x.dtype == torch.float32 and split_gemm.DEFER_EPILOGUE and split_gemm.hand_config(R, self.d, 3 * self.d) is not None)
            # every plane of a hand step as hi / lo PAIRS ([rows, 2d]): the products then move four tiles per K step instead of six (split_gemm.PAIRS)
            L0 = self.layers[0]
            pairs = hand and self._pairs_ok(R, int(self.model.config.decoder_ffn_dim), L0["qkv_w"], L0["fc1"].weight, L0["fc2"].weight)
            st.pairs = pairs
            xp = self._planes_of(x, pairs) if planes else None
            pw = 2 if pairs else 3
            attn_self = lib().sealnn_self_attn_step_x_pairs if pairs else lib().sealnn_self_attn_step_x
            attn_cross = lib().sealnn_cross_attn_step_x_pairs if pairs else lib().sealnn_cross_attn_step_x
            flag = split_gemm._flag(x.device).data_ptr() if hand else None
            library_before = split_gemm.LIBRARY_GEMMS[0]
            for li, L in enumerate(self.layers):
                qkv = self._lin_p(x, xp, L["qkv_w"], L["qkv_b"], defer=True, slabs_ok=hand)
                if hand and isinstance(qkv, split_gemm.Deferred):
                    ap = torch.empty(R, pw * self.d, dtype=torch.float16, device=x.device)
                    check(attn_self(stream, qkv.acc.data_ptr(), qkv.slabs, qkv.acc.stride(0) if qkv.slabs > 1 else 0, qkv.bias.data_ptr(),
                                                        float(qkv.alpha), st.kv[li, 0].data_ptr(), st.kv[li, 1].data_ptr(), st.t.data_ptr(), R, H, T,
                                                        float(self.scale), None, ap.data_ptr(), flag, st.anc.data_ptr()))
                    y = self.split_gemm.from_planes(ap, L["so"].weight, L["so"].bias, defer=True, slabs_ok=True)
                else:
                    a = torch.empty(R, self.d, dtype=x.dtype, device=x.device)
                    if isinstance(qkv, split_gemm.Deferred):
                        check(lib().sealnn_self_attn_step_acc(stream, qkv.acc.data_ptr(), qkv.bias.data_ptr(), float(qkv.alpha), st.kv[li, 0].data_ptr(),
                                                              st.kv[li, 1].data_ptr(), st.t.data_ptr(), R, H, T, float(self.scale), a.data_ptr(),
                                                              st.anc.data_ptr()))
                    else:
                        check(L_.self_attn_step(stream, qkv.data_ptr(), st.kv[li, 0].data_ptr(), st.kv[li, 1].data_ptr(),
                                                st.t.data_ptr(), R, H, T, float(self.scale), a.data_ptr(), st.anc.data_ptr()))
                    y = self._mod(a, L["so"], defer=True)
                x, xp = add_ln(x, y, L["ln1"])
                if hand:
                    qd = self.split_gemm.from_planes(xp, L["cq"].weight, L["cq"].bias, defer=True, slabs_ok=True)
                    cp = torch.empty(R, pw * self.d, dtype=torch.float16, device=x.device)
                    check(attn_cross(stream, qd.acc.data_ptr(), qd.slabs, qd.acc.stride(0) if qd.slabs > 1 else 0, qd.bias.data_ptr(),
                                                         float(qd.alpha), st.ck[li].data_ptr(), st.cv[li].data_ptr(), cbias.data_ptr(), B, K, H, S_pad,
                                                         float(self.scale), None, cp.data_ptr(), flag))
                    y = self.split_gemm.from_planes(cp, L["co"].weight, L["co"].bias, defer=True, slabs_ok=True)
                else:
                    q = self._lin_p(x, xp, L["cq"].weight, L["cq"].bias)
                    c = torch.empty(R, self.d, dtype=x.dtype, device=x.device)
                    check(L_.cross_attn_step(stream, q.data_ptr(), st.ck[li].data_ptr(), st.cv[li].data_ptr(), cbias.data_ptr(),
                                             B, K, H, S_pad, float(self.scale), c.data_ptr()))
                    y = self._mod(c, L["co"], defer=True)
                x, xp = add_ln(x, y, L["ln2"])
                x, xp = add_ln(x, self._ffn(x, xp, L, defer=True, hand=hand, pairs=pairs), L["ln3"])
            st.t.add_(1)
            # The output projection leaves the graph as RAW accumulators when it goes through the split GEMM: its epilogue (alpha * acc +
            # final_logits_bias) is applied by `step` in the same pass that adds a per-query logit bias, if there is one -- torch.addmm would
            # first copy the broadcast bias into the [rows, vocab] output (120 MB at 600 rows) and have the GEMM read it back.
            cfg_lm = split_gemm.hand_config(R, self.lm_w.shape[0], 3 * self.d, True) if pairs else None
            if self.lm_head_finished_in_store and cfg_lm is not None and (cfg_lm & 0x7f) in (5, 6, 7) and ((cfg_lm >> 16) & 0x1fff) <= 1:
                # the output projection FINISHED in the kernel's store: alpha * acc + (final_logits_bias + the query's logit bias), the bias rows in a
                # buffer of this static state that `step` refreshes when the searcher's bias changes -- no pass over the [rows, vocab] logits behind it
                lin = self.split_gemm._of(self.lm_w, self.lm_b.view(-1))
                if getattr(st, "lm_bias_q", None) is None:
                    st.lm_bias_q = lin.bias[None, :].repeat(B, 1).contiguous()
                    st.lm_bias_src, st.lm_bias_key = lin.bias, None
                out = torch.empty(R, self.lm_w.shape[0], dtype=torch.float32, device=x.device)
                check(lib().sealnn_hgemm_nt_ep(stream, xp.data_ptr(), lin.pair_planes().data_ptr(), out.data_ptr(), R, self.lm_w.shape[0], xp.shape[1],
                                               self.lm_w.shape[0], cfg_lm | split_gemm.PAIRS_BIT, st.lm_bias_q.data_ptr(), K, float(lin.alpha)))
                st.library_free = hand and split_gemm.LIBRARY_GEMMS[0] == library_before
                st.lm_epilogue = "in the store"
                return out
            y = self._lin_p(x, xp, self.lm_w, self.lm_b.view(-1), defer=True, slabs_ok=hand)
            # no library GEMM in this step: nothing in it waits for partner workgroups (hipBLASLt's kernels are stream-K), so another stream's
            # library GEMMs may run beside it (retrieval.py: the rescoring forward overlaps the decode steps)
            st.library_free = hand and split_gemm.LIBRARY_GEMMS[0] == library_before
            if isinstance(y, split_gemm.Deferred) and y.slabs == 1:
                st.lm_epilogue = (float(y.alpha), y.bias)
                return y.acc
            st.lm_epilogue = None
            return (y.value() if isinstance(y, split_gemm.Deferred) else y).float()
        st.lm_epilogue = None
        future = st.pos_idx > st.t                                   # cache slots not written yet
        for li, L in enumerate(self.layers):
            qkv = F.linear(x, L["qkv_w"], L["qkv_b"]).view(R, 3, H, dh)
            st.kv[li, 0].index_copy_(2, st.t, qkv[:, 1].unsqueeze(2))
            st.kv[li, 1].index_copy_(2, st.t, qkv[:, 2].unsqueeze(2))
            q = qkv[:, 0].unsqueeze(2) * self.scale
            sc = (q @ st.kv[li, 0].transpose(-1, -2)).masked_fill(future, float("-inf"))
            a = (torch.softmax(sc, dim=-1) @ st.kv[li, 1]).reshape(R, self.d)
            x = L["ln1"](x + L["so"](a))
            cq = (L["cq"](x) * self.scale).view(B, K, H, dh).transpose(1, 2)
            catt = torch.softmax(cq @ st.ck[li] + st.cbias, dim=-1)
            c = (catt @ st.cv[li]).transpose(1, 2).reshape(R, self.d)
            x = L["ln2"](x + L["co"](c))
            x = L["ln3"](x + L["fc2"](L["act"](L["fc1"](x))))
        st.t.add_(1)
        return F.linear(x, self.lm_w, self.lm_b.view(-1)).float()

    # A beam search starts every beam of a query from the same token (reference beam_search.py:214-216 tells them apart by
    # the initial scores [0, -1e9, ...] only), so the first step of the K beams is ONE row of arithmetic: the model runs on
    # `batch` rows instead of `batch x beams` (every GEMM 15 x smaller: the step is then bound by reading the weights once),
    # position 0 of the cache is written for the first beam's row only and the ancestry table points the other beams at it,
    # and the logits row is handed to all K beams.  Self-attention over the single position 0 is softmax([s]) = [1]:
    # the output is V itself, as sealnn_self_attn_step computes it (1.0 * v / 1.0).
    # ON by default since round 4 (``shared_first_step = False``: the full-width first step).  Round 3 kept it off because bench.py stalled
    # with it on; the stall was never this path's -- it was two library GEMM streams in flight at once (a TunableOp pick of the lm_head
    # GEMM then), which the searcher no longer allows.  Clean in every soak run of round 4
    # (profiles/r4_soak_*.txt), +2..6 % queries/s.
    shared_first_step = True

    def _step_static_first(self, st):
        from ._lib import check
        B, K, S_pad, T = st.shape
        H, dh = self.h, self.dh
        x = self.embed(st.tokens.view(B, K)[:, 0])
        x = x + self.pos.weight.index_select(0, st.t + self.pos_offset)
        x = self.ln_emb(x)
        L_ = self._nn(x.dtype)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        cbias = st.cbias.view(B, S_pad)

        def add_ln(res, y, ln):
            out = torch.empty_like(res)
            check(L_.add_layernorm(stream, res.data_ptr(), y.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                   B, self.d, float(ln.eps), out.data_ptr()))
            return out
        if self._first_step_by_hand(x, B):
            return self._step_static_first_by_hand(st, x)
        for li, L in enumerate(self.layers):
            qkv = F.linear(x, L["qkv_w"], L["qkv_b"]).view(B, 3, H, dh)
            st.kv[li, 0].view(B, K, H, T, dh)[:, 0, :, 0].copy_(qkv[:, 1])
            st.kv[li, 1].view(B, K, H, T, dh)[:, 0, :, 0].copy_(qkv[:, 2])
            x = add_ln(x, L["so"](qkv[:, 2].reshape(B, self.d)), L["ln1"])
            q = L["cq"](x)
            c = torch.empty(B, self.d, dtype=x.dtype, device=x.device)
            check(L_.cross_attn_step(stream, q.data_ptr(), st.ck[li].data_ptr(), st.cv[li].data_ptr(), cbias.data_ptr(),
                                     B, 1, H, S_pad, float(self.scale), c.data_ptr()))
            x = add_ln(x, L["co"](c), L["ln2"])
            x = add_ln(x, L["fc2"](L["act"](L["fc1"](x))), L["ln3"])
        st.anc[0].copy_(st.beam0)
        st.t.add_(1)
        return F.linear(x, self.lm_w, self.lm_b.view(-1)).float()

    lm_head_finished_in_store = True    # a hand step's output projection applies alpha, final_logits_bias and the per-query logit bias in its store
    first_step_by_hand = True      # the shared first step through the hand-written kernel (pair planes) where its few rows have configurations

    def _first_step_by_hand(self, x: torch.Tensor, B: int) -> bool:
        """The shared first step is ``batch`` rows (40 of the bench's 600): far below the heights where the split pays in the library, and 73 fp32 library
        GEMMs of ~12 us -- each bound by reading its weight once.  As pair planes through ``sealnn_hgemm_nt`` the same products stream the same bytes in
        ~5 us, and a decode holds no library GEMM at all"""
        from . import split_gemm
        if self.split_gemm is None:
            BartStepDecoder.split_gemm = split_gemm.SplitLinears() if split_gemm.ENABLED else False
        ffn = int(self.model.config.decoder_ffn_dim)
        return bool(self.first_step_by_hand and self.use_fused_kernels and self.split_gemm and split_gemm.PAIRS and split_gemm.HAND_GEMM and split_gemm.DEFER_EPILOGUE
                    and x.is_cuda and x.dtype == torch.float32 and self.dh == 64 and self.d % 32 == 0 and ffn % 32 == 0 and split_gemm.FUSED
                    and all(split_gemm.hand_config(B, n, 3 * k, True) is not None
                            for n, k in ((3 * self.d, self.d), (self.d, self.d), (ffn, self.d), (self.d, ffn))))

    def _step_static_first_by_hand(self, st, x):
        from . import split_gemm
        from ._lib import check, lib
        B, K, S_pad, T = st.shape
        H, dh = self.h, self.dh
        L_ = self._nn(x.dtype)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        cbias = st.cbias.view(B, S_pad)
        flag = split_gemm._flag(x.device).data_ptr()
        of = self.split_gemm._of
        xp = self._planes_of(x, True)
        for li, L in enumerate(self.layers):
            qkv = of(L["qkv_w"], L["qkv_b"]).from_planes(xp).view(B, 3, H, dh)                 # (finished: the cache takes k and v)
            st.kv[li, 0].view(B, K, H, T, dh)[:, 0, :, 0].copy_(qkv[:, 1])
            st.kv[li, 1].view(B, K, H, T, dh)[:, 0, :, 0].copy_(qkv[:, 2])
            # self-attention over the single position 0 is softmax([s]) = [1]: its output is v
            y = of(L["so"].weight, L["so"].bias)(qkv[:, 2].reshape(B, self.d).contiguous(), True, True, True)
            x, xp = self._add_ln(L_, stream, x, y, L["ln1"], B, True, True)
            qd = of(L["cq"].weight, L["cq"].bias).from_planes(xp, True, True)
            cp = torch.empty(B, 2 * self.d, dtype=torch.float16, device=x.device)
            check(lib().sealnn_cross_attn_step_x_pairs(stream, qd.acc.data_ptr(), qd.slabs, qd.acc.stride(0) if qd.slabs > 1 else 0, qd.bias.data_ptr(),
                                                       float(qd.alpha), st.ck[li].data_ptr(), st.cv[li].data_ptr(), cbias.data_ptr(), B, 1, H, S_pad,
                                                       float(self.scale), None, cp.data_ptr(), flag))
            x, xp = self._add_ln(L_, stream, x, of(L["co"].weight, L["co"].bias).from_planes(cp, True, True), L["ln2"], B, True, True)
            h = of(L["fc1"].weight, L["fc1"].bias).from_planes(xp, True, True)
            d1 = h.acc.shape[-1]
            hp = torch.empty(B, 2 * d1, dtype=torch.float16, device=x.device)
            check(lib().sealnn_gelu_planes_acc_slabs_pairs(stream, h.acc.data_ptr(), h.slabs, h.acc.stride(0) if h.slabs > 1 else 0, h.bias.data_ptr(),
                                                           float(h.alpha), B, d1, hp.data_ptr(), flag))
            x, xp = self._add_ln(L_, stream, x, of(L["fc2"].weight, L["fc2"].bias).from_planes(hp, True, True), L["ln3"], B, True, True)
        st.anc[0].copy_(st.beam0)
        st.t.add_(1)
        if split_gemm.hand_config(B, self.lm_w.shape[0], 3 * self.d, True) is None:       # (a vocabulary without a configuration: the library's product)
            split_gemm.LIBRARY_GEMMS[0] += 1
            return F.linear(x, self.lm_w, self.lm_b.view(-1)).float()
        return of(self.lm_w, self.lm_b.view(-1)).from_planes(xp, True, True).value().float()

    @torch.no_grad()
    def start(self, enc_hidden: torch.Tensor, attention_mask: torch.Tensor, num_beams: int, max_len: int, narrow_plan=()) -> None:
        """``narrow_plan``: ascending numbers of leading queries that will have left the decode at its ``narrow`` calls
        (a loop over several decodes in lockstep drops the ones that end first): their buffers and graphs are set up here,
        before the decode writes anything."""
        self._dropped = 0
        if self.use_graph and enc_hidden.is_cuda:
            return self._start_static(enc_hidden, attention_mask, num_beams, max_len, tuple(narrow_plan))
        self._st = None
        return self._start_eager(enc_hidden, attention_mask, num_beams, max_len)

    @torch.no_grad()
    def narrow(self, nq: int) -> None:
        """the first ``nq`` queries (``nq * beams`` rows) leave the decode; the others carry on at the same position with their
        caches where they are"""
        if nq <= 0:
            return
        K = self.beams
        self._dropped = getattr(self, "_dropped", 0) + nq
        if getattr(self, "_st", None) is not None:
            cur = self._st
            nxt = self._st_root.tails[self._dropped]
            off = (self._dropped - cur.dropped) * K
            # ancestry of the remaining rows, re-based to the narrower view of the same cache
            nxt.anc.copy_(cur.anc[:, off:] - off)
            self._st = nxt
        else:
            self.kv = self.kv[:, :, nq * K:]
            self.cross = [(k[nq:], v[nq:]) for k, v in self.cross]
            self.cross_bias = self.cross_bias[nq:]
        self.batch -= nq
        self.rows = self.batch * K
        if self.logit_bias is not None:
            self.logit_bias = self.logit_bias[nq:]

    def _static_tail(self, root, dropped):
        """the static state of the last ``B - dropped`` queries of ``root``: the same caches / cross-attention tensors /
        position counter seen from row ``dropped * K`` on (views, nothing is copied), its own token and logits buffers,
        ancestry table and graph"""
        st = root.tails.get(dropped)
        if st is None:
            B, K, S_pad, T = root.shape
            st = self._Static()
            R2 = (B - dropped) * K
            dev = root.tokens.device
            st.tokens = torch.zeros(R2, dtype=torch.long, device=dev)
            st.t = root.t
            st.kv = root.kv[:, :, dropped * K:]
            st.anc = torch.arange(R2, dtype=torch.int32, device=dev).repeat(T, 1).contiguous()
            st.fused = False
            st.library_free = False
            st.ck, st.cv, st.cbias = root.ck[:, dropped:], root.cv[:, dropped:], root.cbias[dropped:]
            st.pos_idx = root.pos_idx
            st.logits = None
            st.graph = None
            st.shape = (B - dropped, K, S_pad, T)
            st.dropped = dropped
            st.tails = None
            root.tails[dropped] = st
        return st

    def _capture(self, st, dev, first=False):
        forward = self._step_static_first if first else self._step_static
        # (never under torch.inference_mode(): see tree_hidden_graph)
        with torch.inference_mode(False), torch.no_grad():
            # warm up on a side stream, then capture (standard torch recipe); the cache contents
            # written by the warm-up steps are overwritten/masked once t is reset
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    st.t.zero_()
                    forward(st)
            torch.cuda.current_stream(dev).wait_stream(side)
            st.t.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # the aggregation thread may touch the GPU meanwhile
                out = forward(st)
            if first:
                st.first_logits, st.first_graph = out, g
            else:
                st.logits, st.graph = out, g
            st.t.zero_()

    @torch.no_grad()
    def _start_static(self, enc_hidden, attention_mask, num_beams, max_len, narrow_plan=()):
        B, S, d = enc_hidden.shape
        self.batch, self.beams, self.rows, self.max_len = B, num_beams, B * num_beams, max_len
        S_pad = max(16, (S + 15) // 16 * 16)
        st = self._static_for(B, num_beams, S_pad, max_len, enc_hidden.dtype, enc_hidden.device)
        self._st_root = st
        for dropped in narrow_plan:          # graphs of the narrower continuations: captured BEFORE the decode writes its caches
            tail = self._static_tail(st, dropped)
            if tail.graph is None:
                self._capture(tail, enc_hidden.device)
        for li, L in enumerate(self.layers):
            kv = self._ckv(enc_hidden, L).view(B, S, 2, self.h, self.dh)
            st.ck[li, :, :, :, :S] = kv[:, :, 0].permute(0, 2, 3, 1)
            st.cv[li, :, :, :S] = kv[:, :, 1].permute(0, 2, 1, 3)
        st.cbias.fill_(torch.finfo(enc_hidden.dtype).min)
        st.cbias[:, :, :, :S].masked_fill_(attention_mask[:, None, None, :] != 0, 0.0)
        st.t.zero_()
        self._st = st
        self.t = 0
        if st.graph is None:
            self._capture(st, enc_hidden.device)
        if st.first_graph is None and st.fused and self.shared_first_step and num_beams > 1:
            self._capture(st, enc_hidden.device, first=True)

    @torch.no_grad()
    def _start_eager(self, enc_hidden: torch.Tensor, attention_mask: torch.Tensor, num_beams: int, max_len: int) -> None:
        B, S, d = enc_hidden.shape
        self.batch, self.beams, self.rows, self.max_len = B, num_beams, B * num_beams, max_len
        dt, dev = enc_hidden.dtype, enc_hidden.device
        self.cross = []
        for L in self.layers:
            kv = self._ckv(enc_hidden, L).view(B, S, 2, self.h, self.dh)
            k = kv[:, :, 0].permute(0, 2, 3, 1).contiguous()      # [B, H, dh, S]
            v = kv[:, :, 1].permute(0, 2, 1, 3).contiguous()      # [B, H, S, dh]
            self.cross.append((k, v))
        self.cross_bias = torch.zeros(B, 1, 1, S, dtype=dt, device=dev)
        self.cross_bias.masked_fill_(attention_mask[:, None, None, :] == 0, torch.finfo(dt).min)
        self.kv = torch.zeros(len(self.layers), 2, self.rows, self.h, max_len, self.dh, dtype=dt, device=dev)
        self.t = 0

    @torch.no_grad()
    def reorder(self, beam_idx: torch.Tensor) -> None:
        """new row r continues old row beam_idx[r] (HF ``_reorder_cache``)."""
        if getattr(self, "_st", None) is not None:
            if self._st.fused:
                self._st.anc.copy_(self._st.anc.index_select(1, beam_idx))
            else:
                self._st.kv.copy_(self._st.kv.index_select(2, beam_idx))
            return
        self.kv[:, :, :, :, :self.t] = self.kv[:, :, beam_idx, :, :self.t]

    def _bias_per_query(self, bias: torch.Tensor) -> torch.Tensor:
        """final_logits_bias + the per-query logit bias, [batch, vocab]; made once per (bias tensor) and reused by the steps of a decode"""
        import weakref
        lb = self.logit_bias
        hit = self.__dict__.get("_bias_q")           # (validated by the tensor OBJECTS and the version: an address alone is reused)
        if hit is None or hit[0]() is not lb or hit[1] != lb._version or hit[2]() is not bias:
            hit = self.__dict__["_bias_q"] = (weakref.ref(lb), lb._version, weakref.ref(bias), bias[None, :] + lb)
        return hit[3]

    def steps_are_library_free(self) -> bool:
        """do the remaining model steps of the running decode (its static state and the narrower continuations set up for it) hold no library
        GEMM?  (The encoder, the cross-attention K / V and the shared first step do: the decode's PREFIX.)"""
        root = getattr(self, "_st_root", None)
        if root is None or getattr(self, "_st", None) is None:
            return False
        states = [root] + [t for t in (root.tails or {}).values() if t.graph is not None]
        return all(getattr(s, "library_free", False) for s in states)

    def _library_prefix_done(self):
        """called once the first model step of a decode is enqueued: tells whoever asked (``model._seal_after_library_prefix``, set by the
        overlapped searcher around its decode enqueues) that everything this decode still has to run is free of library GEMMs"""
        cb = getattr(self.model, "_seal_after_library_prefix", None)
        if cb is not None and self.steps_are_library_free():
            cb()

    def step_buffers(self):
        """(token buffer, ancestry table) of the static fused decode state, for a caller that advances the beams on the device
        (``fmi_dev_beam_step``): the next step's input tokens are written straight into the buffer ``step`` reads, and re-ranking the
        beams permutes the columns of the table in the same launch.  (None, None) off the fused static path: ``step(tokens)`` +
        ``reorder(beam_idx)`` as before."""
        st = getattr(self, "_st", None)
        if st is None or not st.fused:
            return None, None
        return st.tokens, st.anc

    @torch.no_grad()
    def step(self, tokens: torch.Tensor, beams_identical: bool = False) -> torch.Tensor:
        """tokens [rows] at decoder position ``self.t`` -> next-token logits [rows, vocab] (fp32).  ``beams_identical``: the
        caller's promise, at position 0, that the ``beams`` rows of every query hold the same token (the start of a beam
        search): the model then runs once per query (``_step_static_first``)."""
        R, B, K, H, dh, t = self.rows, self.batch, self.beams, self.h, self.dh, self.t
        if getattr(self, "_st", None) is not None:
            st = self._st
            if tokens.data_ptr() != st.tokens.data_ptr():          # (fmi_dev_beam_step writes the next input where this step reads it)
                st.tokens.copy_(tokens)
            if beams_identical and t == 0 and st.first_graph is not None:
                st.first_graph.replay()
                self.t += 1
                self._library_prefix_done()
                logits = st.first_logits                                  # [B, V]
                if self.logit_bias is not None:
                    logits = logits + self.logit_bias
                return logits[:, None, :].expand(B, K, logits.shape[-1]).reshape(R, -1)
            if getattr(st, "lm_epilogue", None) == "in the store":
                # the bias rows the output projection's store adds: final_logits_bias (+ the searcher's per-query logit bias), refreshed when they change
                # (validated by the tensor OBJECT and its version, as _bias_per_query's own cache is: an address alone is reused by the allocator)
                src = self._bias_per_query(st.lm_bias_src) if self.logit_bias is not None else st.lm_bias_src
                key = st.lm_bias_key
                if key is None or key[0]() is not src or key[1] != src._version:
                    import weakref
                    st.lm_bias_q.copy_(src if src.dim() == 2 else src[None, :].expand_as(st.lm_bias_q))
                    st.lm_bias_key = (weakref.ref(src), src._version)
            st.graph.replay()
            self.t += 1
            if t == 0:
                self._library_prefix_done()
            logits = st.logits
            ep = getattr(st, "lm_epilogue", None)
            if ep == "in the store":
                return logits
            if ep is not None:
                alpha, bias = ep                                       # (alpha a power of two: alpha * acc is exact, one rounding in the add)
                if self.logit_bias is not None:
                    return torch.add(self._bias_per_query(bias)[:, None, :], logits.view(B, K, -1), alpha=alpha).view(R, -1)
                return torch.add(bias, logits, alpha=alpha)
            if self.logit_bias is not None:
                logits = (logits.view(B, K, -1) + self.logit_bias[:, None, :]).view(R, -1)
            return logits
        x = self.embed(tokens)       # HF's BartScaledWordEmbedding applies embed_scale itself
        x = x + self.pos.weight[t + self.pos_offset]
        x = self.ln_emb(x)
        for li, L in enumerate(self.layers):
            qkv = F.linear(x, L["qkv_w"], L["qkv_b"]).view(R, 3, H, dh)
            self.kv[li, 0, :, :, t] = qkv[:, 1]
            self.kv[li, 1, :, :, t] = qkv[:, 2]
            q = qkv[:, 0].unsqueeze(2) * self.scale                       # [R, H, 1, dh]
            keys = self.kv[li, 0, :, :, :t + 1]                            # [R, H, t+1, dh]
            vals = self.kv[li, 1, :, :, :t + 1]
            att = torch.softmax(q @ keys.transpose(-1, -2), dim=-1)        # [R, H, 1, t+1]
            a = (att @ vals).reshape(R, self.d)
            x = L["ln1"](x + L["so"](a))
            ck, cv = self.cross[li]
            cq = (L["cq"](x) * self.scale).view(B, K, H, dh).transpose(1, 2)   # [B, H, K, dh]
            catt = torch.softmax(cq @ ck + self.cross_bias, dim=-1)        # [B, H, K, S]
            c = (catt @ cv).transpose(1, 2).reshape(R, self.d)
            x = L["ln2"](x + L["co"](c))
            x = L["ln3"](x + L["fc2"](L["act"](L["fc1"](x))))
        self.t += 1
        logits = F.linear(x, self.lm_w, self.lm_b.view(-1)).float()
        if self.logit_bias is not None:
            logits = (logits.view(B, K, -1) + self.logit_bias[:, None, :]).view(R, -1)
        return logits
