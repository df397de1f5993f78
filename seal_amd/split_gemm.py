"""fp32-accurate linear layers on the fp16 matrix cores (on by default; ``SEAL_SPLIT_GEMM=0`` = plain fp32 GEMMs).

The reference runs BART in fp32 (seal/retrieval.py:560-588 loads the checkpoint as it is; beam_search.py:251 takes the
log-softmax of fp32 logits) and north_star wants the hypothesis scores within 1e-4 of that, which rules the bf16 / fp16 model
out (measured: 0.1).  But fp32 GEMMs are what a search step spends its GPU time on (~50 of 75 ms, at ~100
of the 157 TF/s the fp32 MFMA path has), while the fp16 matrix cores of gfx950 do 2.5 PFLOP/s dense.  An fp32 number is two
fp16 numbers to 22 bits,

    x = hi + lo,    hi = fp16(x),    lo = fp16(x - hi)            (|lo| <= 2^-11 |hi|)

and a product of two such numbers is three fp16 products up to a 2^-22 term:

    x . w  =  hi_x . hi_w  +  hi_x . lo_w  +  lo_x . hi_w  (+ lo_x . lo_w, dropped)

so   y = x W^T   becomes ONE fp16 GEMM with fp32 accumulation over a three times longer inner dimension,

    y = alpha * [ hi_x | hi_x | lo_x * 2^11 ]  .  [ hi_w | lo_w | hi_w * 2^-11 ]^T,

three times the FLOPs on units sixteen times as fast.  Emulated with numpy / torch on the CPU (tests/test_split_gemm.py) the
representation error of the split is 4-5 times SMALLER than the rounding error of an fp32 GEMM's own accumulation (K = 1024 .. 4096,
activations with outlier channels), i.e. the result is fp32-grade.

Scaling keeps every plane in fp16's normal range whatever the matrix cores do with subnormals:
* weights (offline): W is multiplied by a power of two ``s_w`` that puts max|W| just below 2^13; then ``lo_w`` and ``hi_w * 2^-11`` of
  every weight that matters are normal numbers; ``alpha = 1 / s_w`` undoes it (exact);
* activations (per call, ``sealnn_split_planes``): ``lo_x`` is stored multiplied by 2^11 -- the magnitude of x itself -- and its
  partner plane of W carries the 2^-11.  |x| must stay below fp16's 65504 (BART activations are O(10^2) at most): the kernel counts
  violations in a flag that ``overflowed()`` reads.

``SplitLinear(weight, bias)(x)`` is ``F.linear(x, weight, bias)`` for an fp32 ``x`` [rows, K].  On the GPU the planes come from the
HIP kernel and the product from ``torch.addmm(..., out_dtype=torch.float32)`` (hipBLASLt, fp16 in / fp32 out); on the CPU (the tests'
checker of the arithmetic) both are emulated with torch ops.  ``defer=True`` hands back the raw accumulators (``torch.mm``) with
``(alpha, bias)`` for a consumer that applies the epilogue as it reads (``Deferred``, ``DEFER_EPILOGUE``; the sealnn_*_acc kernels).
"""
import math
import os

import torch

# ON by default since round 4 (measured on an MI355X, profiles/r4_split_gemm_probe.txt: the products below run 1.3 - 2.2 x faster than the
# library's fp32 GEMM, the hypothesis scores stay within 1e-4 of HF's fp32 forward); SEAL_SPLIT_GEMM=0 runs every product as F.linear.
ENABLED = os.environ.get("SEAL_SPLIT_GEMM", "1") == "1"
# Which products go through the split: three times the FLOPs on the fp16 matrix cores only pays once the product fills the chip
# (tools/split_gemm_probe.py, us per call fp32 -> split GEMM alone / with the sealnn_split_planes pass in front of it):
#   600 x 3072 x 1024 (qkv)  39.6 -> 28.4 / 32.9     600 x 4096 x 1024 (fc1) 60.8 -> 31.1 / 35.6    600 x 1024 x 4096 (fc2) 56.9 -> 48.6 / 55.8
#   600 x 1024 x 1024        17.2 -> 18.1 / 24.1     600 x 50265 x 1024 (lm_head) 518 -> 275 / 282   300 x 50265 x 1024      274 -> 146 / 149
#   300 x 3072 x 1024        21.2 -> 19.9 / 23.8     300 x 4096 x 1024       27.5 -> 23.1 / 26.6    300 x 1024 x 4096       36.8 -> 36.1 / 41.5
#    40 x 50265 x 1024       56.8 -> 66.1 / 69.3     3200 rows (the rescoring forward): every product 1.9 - 2.4 x faster
# i.e. from ~0.9 GFLOP (rows x N x K multiply-adds) on when the operand planes exist already (written by the kernel that produced
# the activation), from ~1.8 GFLOP when a split pass has to run first, and never below 128 rows.
MIN_ROWS = 128
MIN_MACS_WITH_PLANES = 0.9e9
MIN_MACS_WITH_SPLIT_PASS = 1.8e9
# the planes written by the kernels that produce the activations (sealnn_add_layernorm_planes, sealnn_gelu_planes) instead of a pass of
# sealnn_split_planes in front of every product
FUSED = True
# the GEMM's epilogue (alpha * acc + bias) applied by the kernel that READS the product (sealnn_*_acc) instead of by the library call:
# torch.addmm(out_dtype=float32) does not use hipBLASLt's bias epilogue but copies the broadcast bias into the output first -- one
# [rows, N] strided copy in front of every product, 5 ms of a 72 ms search step (rocprofv3 trace of round 4: 817 copies per batch)
DEFER_EPILOGUE = True
LO_SHIFT = 11                      # bits between the planes: fp16 has an 11-bit significand
_flags = {}                        # device -> int32 counter of unsplittable activations


def _flag(device) -> torch.Tensor:
    f = _flags.get(device)
    if f is None:
        f = _flags[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


_seen = {}                         # device -> violations already reported


def flag_snapshot(device):
    """the violation counter of ``device`` on its way to pinned host memory behind what the current stream holds (nothing waits);
    ``check_snapshot`` reads it once that work is known to be complete.  None when no split product ever ran on the device."""
    f = _flags.get(torch.device(device))
    if f is None:
        return None
    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(f, non_blocking=True)
    return host


def check_snapshot(host, device=None) -> None:
    """raises when activations left fp16's range since the last check: the products that saw them are wrong"""
    if host is None:
        return
    n = int(host[0])
    key = str(torch.device(device)) if device is not None else "None"
    if n < _seen.get(key, 0):
        _seen[key] = 0                                # the device counter was reset behind our back (overflowed())
    if n > _seen.get(key, 0):
        _seen[key] = n
        raise RuntimeError(f"split GEMM: {n} groups of activations beyond fp16's range (|x| > 65504) were seen; the scores of this call are "
                           f"unreliable -- run with SEAL_SPLIT_GEMM=0 (plain fp32 GEMMs) for this model")


def overflowed(device) -> int:
    """activations beyond fp16's range seen by the split kernel on ``device`` since the last call (synchronises)"""
    f = _flags.get(torch.device(device))
    if f is None:
        return 0
    n = int(f.item())
    f.zero_()
    _seen.pop(str(torch.device(device)), None)      # the counter restarts: so does what check_snapshot has reported already
    return n


def split_planes_reference(x: torch.Tensor) -> torch.Tensor:
    """[rows, K] fp32 -> [rows, 3K] fp16 = [hi | hi | lo * 2^11]: the arithmetic of ``sealnn_split_planes`` in torch ops"""
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * float(2 ** LO_SHIFT)).to(torch.float16)
    return torch.cat([hi, hi, lo], dim=1)


def split_weight(weight: torch.Tensor):
    """[N, K] fp32 -> ([N, 3K] fp16 = [hi | lo | hi * 2^-11] of ``weight * s_w``, alpha = 1 / s_w)"""
    w = weight.detach().float()
    top = float(w.abs().max())
    s_w = 2.0 ** math.floor(math.log2(2.0 ** 13 / top)) if top > 0 and math.isfinite(top) else 1.0
    ws = w * s_w
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    hi_s = (hi.float() * float(2.0 ** -LO_SHIFT)).to(torch.float16)
    return torch.cat([hi, lo, hi_s], dim=1).contiguous(), 1.0 / s_w


class Deferred:
    """a product whose epilogue is left to the kernel that reads it: the value is ``alpha * acc + bias`` (bias over the last dimension).
    ``slabs`` > 1: ``acc`` is [slabs, rows, N], the slabs of a split-K product (``sealnn_hgemm_nt``), to be added in slab order."""
    __slots__ = ("acc", "bias", "alpha", "slabs")

    def __init__(self, acc, bias, alpha, slabs=1):
        self.acc, self.bias, self.alpha, self.slabs = acc, bias, alpha, slabs

    def value(self) -> torch.Tensor:
        """materialised (what the library's own epilogue computes: alpha a power of two, one rounding in the add)"""
        acc = self.acc
        if self.slabs > 1:
            tot = acc[0]
            for s in range(1, self.slabs):
                tot = tot + acc[s]
            acc = tot
        return acc * self.alpha + self.bias


# The hand-written fp16 product (sealnn_hgemm_nt, seal_amd/csrc/hgemm_kernels.hip) for the decode step's skinny products whose consumer can add
# split-K slabs as it reads them (sealnn_add_layernorm_acc_slabs): measured on an MI355X (profiles/r5_hgemm_probe.txt, us per call, library ->
# hand-written with 4 slabs): fc2 [rows, 3 x 4096] x [1024]: 600 rows 43.2 -> 32.2, 300 rows 31.5 -> 21.3.  (N, K') -> {max rows: config};
# config = tile | stages << 8 | K groups << 12 | slices << 16 (sealnn.h).  ``HAND_GEMM = False``: the library for everything.
HAND_GEMM = True
# Round 6: EVERY product of a decode step (fused path, fp32, 129 .. 640 rows) runs in the hand-written kernel -- also the three that round 5 left to
# the library (profiles/r6_hgemm_probe_decode.txt, us, library -> hand): qkv at 600 rows 23.8 -> 19.5; fc1 25.2 -> 27.9 (600), 18.3 -> 18.5 (300);
# lm_head 237.9 -> 229.6 (600), 126.4 -> 157.7 (300) with the row tile as the fastest grid index and the workgroups grouped by XCD (tile + 128:
# each 128-column tile of the 309 MB matrix is read from memory once, by the XCD whose five row tiles share it; column-major order: 305 us).
# Not for the microseconds -- they roughly cancel -- but because a step without a library GEMM has no stream-K kernel in it, so the rescoring
# forward (the library's GEMMs, on another stream) may run BESIDE the decode steps (retrieval.py; DESIGN.md section 9: at most one stream of
# stream-K kernels at a time).
# The tall tile (tile codes 5..7: 320 rows x 128 / 64 / 96 columns, eight waves, one workgroup per CU) where it wins with W streamed from memory as a
# decode step streams it (profiles/r6_hgemm_probe_tall.txt, us, 4-wave tile -> tall): fc2 38.9 -> 24.2 (600 rows, 8 slabs), 25.7 -> 16.8 (300, 16
# slabs); fc1 28.9 -> 25.0 (600, 2 slabs that GELU adds), 24.8 -> 17.0 (300, 4 slabs); qkv 24.8 -> 18.9 (600 rows, 320 x 96, 4 slabs).  The d x d
# projections stay (11.6 / 7.4 against 10.7 / 8.9 with twice the slabs), so does lm_head (234 against 248: 786 tall tiles are 3.07 rounds of 256 CUs).
# At the heights of the rescoring forest (~3 300 rows) the library stays: it runs qkv, fc1 and the K / V of the encoder states at 0.9 PFLOP/s, and the two
# products of 1 024 columns that the tall tile wins alone (112 -> 90, 33 -> 28 us: profiles/r6_hgemm_probe_rescoring.txt) do not win beside a decode
# whose own tall tiles want the same whole-CU LDS (430.9 against 433 .. 436 queries/s).
_TALL = lambda tile, stages, slices: tile | (stages << 8) | (1 << 12) | (slices << 16)
HAND_CONFIGS = {
    (1024, 12288): ((320, _TALL(6, 3, 16)), (640, _TALL(6, 3, 8))),                                                  # fc2
    (1024, 3072): ((320, 2 | (2 << 8) | (2 << 12) | (4 << 16)), (640, 2 | (2 << 8) | (1 << 12) | (4 << 16))),       # the d x d projections: 14.3 -> 10.7, 11.1 -> 6.6
    (3072, 3072): ((320, 2 | (2 << 8) | (2 << 12) | (2 << 16)), (640, _TALL(7, 3, 4))),                             # qkv
    (2048, 3072): ((320, 2 | (2 << 8) | (2 << 12) | (2 << 16)), (640, _TALL(6, 3, 4))),                             # the cross-attention K / V of a batch's encoder states
    (4096, 3072): ((320, _TALL(6, 3, 4)), (640, _TALL(6, 3, 2))),                                                    # fc1 (GELU adds the slabs)
    (50265, 3072): ((640, (1 + 128) | (2 << 8) | (1 << 12) | (1 << 16)),),                                           # lm_head (one slab)
}
# PAIRS (round 6): inside a decode step whose every product is the hand-written kernel's, the operand planes are hi / lo PAIRS per 32 columns -- a 128-byte
# line of a row = [hi of 32 columns | lo * 2^11 (activations) resp. lo (weights) of the same 32] -- and a K step is the three products of those columns
# (hi.hi + hi.lo into one accumulator, (lo 2^11).hi into a second that joins it times 2^-11).  The three-block operand [hi | hi | lo 2^11] x [hi | lo | hi 2^-11]
# feeds the matrix cores the same three products from six tiles of which two are copies; here four tiles travel: two thirds of the bytes through the CU's
# load path, which is what bounds these products (hgemm_kernels.hip).  The kernels that write the planes have _pairs twins (include/sealnn.h), the weights'
# pairs are made from their planes on first use.  Same terms, another order of the fp32 additions.  ``PAIRS = False``: three blocks everywhere.
PAIRS = True
PAIRS_BIT = 1 << 29
# (N, 3K) -> {max rows: config} for the PAIRS form (2K / 64 K steps); profiles/r6_hgemm_probe_pairs.txt, us, three blocks -> pairs, W from memory:
# 600 rows: d x d 11.5 -> 8.7, qkv 21.2 -> 18.2, fc1 28.1 -> 21.3, fc2 26.0 -> 21.3, lm_head 249 -> 209; 300 rows: 7.8 -> 6.6, 16.9 -> 10.5, 17.2 -> 14.7,
# 17.3 -> 16.1, 163 -> 114 (a third fewer bytes buys 14 .. 38 %: with the load path relieved the reads-and-MFMA phase of a step shows)
_T4 = lambda tile, stages, kgroups, slices: tile | (stages << 8) | (kgroups << 12) | (slices << 16)
HAND_CONFIGS_PAIRS = {
    (1024, 12288): ((64, _T4(2, 3, 1, 16)), (320, _TALL(6, 3, 16)), (640, _TALL(6, 3, 8)), (1536, _T4(4, 3, 1, 2)), (4096, _TALL(7, 3, 2))),
    (1024, 3072): ((64, _T4(2, 3, 1, 8)), (320, _T4(2, 2, 2, 4)), (640, _T4(2, 2, 1, 4)), (1536, _T4(4, 3, 1, 2)), (4096, _TALL(7, 3, 2))),
    (3072, 3072): ((64, _T4(2, 3, 1, 8)), (320, _T4(2, 2, 2, 2)), (640, _TALL(6, 3, 2)), (1536, _TALL(7, 3, 2)), (4096, _T4(3, 2, 1, 1))),
    (4096, 3072): ((64, _T4(2, 3, 1, 4)), (320, _T4(4, 2, 1, 4)), (640, _TALL(6, 3, 2)), (1536, _TALL(5, 2, 2)), (4096, _TALL(7, 3, 1))),
    (2048, 3072): ((1536, _TALL(6, 3, 2)), (4096, _TALL(7, 3, 1))),
    (50265, 3072): ((64, _T4(132, 3, 1, 1)), (320, _TALL(5, 2, 1)), (640, _TALL(7, 3, 1))),
}
# (<= 64 rows: the shared first step of a decode, 40 rows in the bench -- fp32 library GEMMs of ~12 us become 4 .. 7: BartStepDecoder._step_static_first_by_hand)
# Beyond a decode step's 640 rows only the PAIRS form has configurations: with three-block planes the library's 0.9 PFLOP/s stays ahead of the hand-written
# kernel at 1 280 rows (a batch's encoder: 40 inputs x 32 tokens) and at ~3 300 (the rescoring forest); with a third fewer bytes the hand-written kernel is
# (profiles/r6_hgemm_probe_rescoring.txt, us, library -> pairs): 1 280 rows: qkv 37.3 -> 25.9, d x d 19.6 -> 15.1, fc1 44.3 -> 32.6, fc2 67.7 -> 43.5, K / V of
# the encoder states 30.5 -> 19.0; 3 328 rows: 68.7 -> 61.8, 32.4 -> 24.2, 91.7 -> 81.6, 118.2 -> 79.1, 53.8 -> 41.2.
# library GEMMs issued through this module and BartStepDecoder._lin since the process started: a step decoder that captures its graph reads it
# before and after to learn whether the step is free of them (BartStepDecoder._step_static)
LIBRARY_GEMMS = [0]


def hand_config(rows: int, n: int, k3: int, pairs: bool = False):
    """the sealnn_hgemm_nt configuration for a [rows, k3] x [n, k3]^T product (``pairs``: as [rows, 2 k3 / 3] pair planes), or None: the
    library's GEMM serves it"""
    if not HAND_GEMM:
        return None
    for max_rows, cfg in (HAND_CONFIGS_PAIRS if pairs else HAND_CONFIGS).get((n, k3), ()):
        if rows <= max_rows:
            return cfg
    return None


class SplitLinear:
    """``F.linear(x, weight, bias)`` for fp32 ``x`` [rows, K] through one fp16 GEMM of inner dimension 3K (see the module text)."""

    def __init__(self, weight: torch.Tensor, bias=None):
        self.N, self.K = weight.shape
        if self.K % 4:
            raise ValueError(f"SplitLinear: K={self.K} must be a multiple of 4")
        self.planes, self.alpha = split_weight(weight)                 # [N, 3K] fp16, row-major like the weight F.linear takes
        self.wt = self.planes.t()                                      # the [3K, N] operand (a view)
        b = bias.detach().float().reshape(-1) if bias is not None else torch.zeros(self.N, device=weight.device)
        self.bias = b.contiguous()

    def __call__(self, x: torch.Tensor, defer: bool = False, slabs_ok: bool = False, pairs: bool = False):
        """``pairs``: split x into hi / lo PAIRS ([rows, 2K]: the hand-written kernel's three-products-per-K-step operand) where that product has a
        configuration"""
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.K:
            raise ValueError(f"SplitLinear: expected fp32 [rows, {self.K}], got {x.dtype} {tuple(x.shape)}")
        if not x.is_cuda:
            # the checker of the arithmetic (tests): same planes, products summed in fp32
            return self.from_planes(split_planes_reference(x), defer)
        from ._lib import check, lib
        x = x.contiguous()
        pairs = bool(pairs and PAIRS and self.K % 32 == 0 and (self.N % 4 == 0 or defer) and hand_config(x.shape[0], self.N, 3 * self.K, True) is not None)
        a = torch.empty(x.shape[0], (2 if pairs else 3) * self.K, dtype=torch.float16, device=x.device)
        fn = lib().sealnn_split_planes_pairs if pairs else lib().sealnn_split_planes
        check(fn(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), x.shape[0], self.K, a.data_ptr(), _flag(x.device).data_ptr()))
        return self.from_planes(a, defer, slabs_ok)

    def pair_planes(self) -> torch.Tensor:
        """the weight as hi / lo pairs per 32 columns, [N, 2K] fp16 (made from the planes on first use)"""
        p = self.__dict__.get("_pairs")
        if p is None:
            if self.K % 32:
                raise ValueError(f"SplitLinear: pair planes need K={self.K} to be a multiple of 32")
            hi, lo = self.planes[:, :self.K].reshape(self.N, -1, 32), self.planes[:, self.K:2 * self.K].reshape(self.N, -1, 32)
            p = self._pairs = torch.stack((hi, lo), 2).reshape(self.N, 2 * self.K).contiguous()
        return p

    def _hand(self, planes: torch.Tensor):
        """the product's raw accumulators [slices, rows, N] from sealnn_hgemm_nt, or None where the library serves it (pair planes -- [rows, 2K]
        -- have no other server: an error then)"""
        pairs = planes.shape[1] == 2 * self.K
        cfg = hand_config(planes.shape[0], self.N, 3 * self.K, pairs) if planes.is_cuda and planes.is_contiguous() else None
        if cfg is None:
            if pairs:
                raise RuntimeError(f"SplitLinear: no hand-written configuration for pair planes of a [{planes.shape[0]}, {self.K}] x [{self.N}, {self.K}]^T product")
            return None
        from ._lib import check, lib
        slices = max(1, (cfg >> 16) & 0x1fff)
        acc = torch.empty(slices, planes.shape[0], self.N, dtype=torch.float32, device=planes.device)
        w = self.pair_planes() if pairs else self.planes
        check(lib().sealnn_hgemm_nt(torch.cuda.current_stream(planes.device).cuda_stream, planes.data_ptr(), w.data_ptr(), acc.data_ptr(),
                                    planes.shape[0], self.N, planes.shape[1], self.N, cfg | (PAIRS_BIT if pairs else 0)))
        return acc

    def from_planes(self, planes: torch.Tensor, defer: bool = False, slabs_ok: bool = False):
        """the product for an activation whose planes [rows, 3K] fp16 exist already (``defer``: as ``Deferred`` raw accumulators;
        ``slabs_ok``: the consumer adds split-K slabs itself, so the hand-written kernel may serve the product).  A FINISHED product
        (``defer`` off) of a shape the hand-written kernel has a configuration for runs there too, with ``sealnn_finish_product`` behind it."""
        if defer:
            acc = self._hand(planes) if slabs_ok or planes.shape[1] == 2 * self.K else None
            if acc is not None:
                return Deferred(acc if acc.shape[0] > 1 else acc[0], self.bias, self.alpha, slabs=acc.shape[0])
            LIBRARY_GEMMS[0] += 1 if planes.is_cuda else 0
            acc = torch.mm(planes, self.wt, out_dtype=torch.float32) if planes.is_cuda else torch.mm(planes.float(), self.wt.float())
            return Deferred(acc, self.bias, self.alpha)
        if not planes.is_cuda:
            return torch.addmm(self.bias, planes.float(), self.wt.float(), alpha=self.alpha)
        acc = self._hand(planes) if self.N % 4 == 0 else None
        if acc is not None:
            from ._lib import check, lib
            out = torch.empty(planes.shape[0], self.N, dtype=torch.float32, device=planes.device)
            check(lib().sealnn_finish_product(torch.cuda.current_stream(planes.device).cuda_stream, acc.data_ptr(), acc.shape[0], acc.stride(0),
                                              self.bias.data_ptr(), float(self.alpha), planes.shape[0], self.N, out.data_ptr()))
            return out
        LIBRARY_GEMMS[0] += 1
        return torch.addmm(self.bias, planes, self.wt, alpha=self.alpha, out_dtype=torch.float32)


def tensor_version(t: torch.Tensor) -> int:
    """the tensor's in-place modification counter; an inference tensor (made under ``torch.inference_mode``: the rescoring entry points
    build their fused q/k/v weights there) has none and cannot be modified in place outside inference mode: -1 stands for ``as made``"""
    return -1 if t.is_inference() else t._version


class SplitLinears:
    """the split operands of a model's weights, made on first use and kept (keyed by the weight tensor's storage)"""

    def __init__(self):
        self._by_weight = {}

    @staticmethod
    def wants(weight: torch.Tensor, rows: int, have_planes: bool = True) -> bool:
        """does ``F.linear(x[rows, K], weight[N, K])`` gain from the split?  (``have_planes``: x's operand planes exist already)"""
        n, k = weight.shape
        macs = float(rows) * n * k
        return rows >= MIN_ROWS and k % 4 == 0 and macs >= (MIN_MACS_WITH_PLANES if have_planes else MIN_MACS_WITH_SPLIT_PASS)

    def _of(self, weight, bias) -> SplitLinear:
        # keyed by the weight's storage address, VALIDATED by the tensor object itself and its version counter: an address is reused once a
        # model is freed (a second model loaded into the same process got the first one's planes: found by a test that builds two models)
        # and a checkpoint loaded in place keeps its address
        import weakref
        key = (weight.data_ptr(), tuple(weight.shape))
        version = tensor_version(weight)
        hit = self._by_weight.get(key)
        if hit is not None:
            ref, seen, lin = hit
            if ref() is weight and seen == version:
                return lin
        lin = SplitLinear(weight, bias)
        self._by_weight[key] = (weakref.ref(weight), version, lin)
        return lin

    def __call__(self, x: torch.Tensor, weight: torch.Tensor, bias=None, defer: bool = False, slabs_ok: bool = False, pairs: bool = False):
        """``defer``: a product that goes through the split comes back as ``Deferred`` (one that does not, as the finished tensor)"""
        if not self.wants(weight, x.shape[0], have_planes=False):
            LIBRARY_GEMMS[0] += 1 if x.is_cuda else 0
            return torch.nn.functional.linear(x, weight, bias)
        return self._of(weight, bias)(x, defer and DEFER_EPILOGUE, slabs_ok, pairs and defer == (defer and DEFER_EPILOGUE))

    def from_planes(self, planes: torch.Tensor, weight: torch.Tensor, bias=None, defer: bool = False, slabs_ok: bool = False):
        return self._of(weight, bias).from_planes(planes, defer and DEFER_EPILOGUE, slabs_ok)
