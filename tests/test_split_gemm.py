"""seal_amd/split_gemm.py: an fp32 linear layer as ONE fp16 GEMM over three planes.  On the CPU the arithmetic itself (planes emulated
with torch ops, products summed in fp32): the split represents x and W to 22 bits and the product lands as close to the float64 result
as an fp32 GEMM does.  The GPU half: the HIP split kernels (stand-alone and fused into add+LayerNorm / GELU), the hipBLASLt fp16 -> fp32
product, capture in a hipGraph, and the size policy of ``SplitLinears.wants`` (measured on an MI355X, profiles/r4_split_gemm_probe.txt)."""
import os

import pytest
import torch

from seal_amd.split_gemm import LO_SHIFT, SplitLinear, SplitLinears, split_planes_reference, split_weight


def _activations(rows, K, seed, outliers=50.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, K, generator=g)
    x[:, :4] *= outliers                      # residual streams have a few channels far above the rest
    x[:, 7] *= 1e-4                           # and some far below
    return x


def test_planes_hold_22_bits_of_every_element():
    x = _activations(64, 256, 0)
    p = split_planes_reference(x)
    K = x.shape[1]
    assert p.dtype == torch.float16 and p.shape == (64, 3 * K) and torch.equal(p[:, :K], p[:, K:2 * K])
    back = p[:, :K].double() + p[:, 2 * K:].double() * 2.0 ** -LO_SHIFT
    err = (back - x.double()).abs()
    normal = x.abs() >= 2.0 ** -13            # hi and the stored lo are normal fp16 numbers
    assert normal.float().mean() > 0.98 and (~normal).any()
    assert (err[normal] / x.double().abs()[normal]).max().item() <= 2.0 ** -21
    assert err[~normal].max().item() <= 2.0 ** -35        # below fp16's normal range: absolute, far under anything a dot product notices
    # the stored lo plane has the magnitude of x, not 2^-11 of it
    lo = p[:, 2 * K:].float().abs()
    assert (lo <= x.abs() * 1.0001 + 2.0 ** -14).all()          # (2^-14: where hi itself is subnormal)


@pytest.mark.parametrize("top", [0.5, 1e-3, 300.0])
def test_weight_planes_and_scale(top):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(48, 64, generator=g) * 0.05
    w[0, 0] = top
    planes, alpha = split_weight(w)
    K = w.shape[1]
    s_w = 1.0 / alpha
    assert 2.0 ** 12 < float(w.abs().max()) * s_w <= 2.0 ** 13 and s_w == 2.0 ** round(__import__("math").log2(s_w))
    hi, lo, his = planes[:, :K].double(), planes[:, K:2 * K].double(), planes[:, 2 * K:].double()
    err = ((hi + lo) * alpha - w.double()).abs().max().item()
    assert err <= float(w.abs().max()) * 2.0 ** -21
    big = hi.abs() >= 1.0                       # where it matters: hi * 2^-11 is exact (a normal fp16 number)
    assert torch.equal(his[big], hi[big] * 2.0 ** -LO_SHIFT)
    zeros, a0 = split_weight(torch.zeros(4, 8))
    assert a0 == 1.0 and not zeros.any()


@pytest.mark.parametrize("K,N", [(1024, 768), (4096, 256)])
def test_split_linear_is_as_close_to_float64_as_an_fp32_gemm(K, N, monkeypatch):
    from seal_amd import split_gemm
    monkeypatch.setattr(split_gemm, "MIN_ROWS", 0)
    monkeypatch.setattr(split_gemm, "MIN_MACS_WITH_SPLIT_PASS", 0.0)          # the size policy aside: every product through the split
    g = torch.Generator().manual_seed(K)
    x = _activations(96, K, 2)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    fp32 = torch.nn.functional.linear(x, w, b)
    got = SplitLinear(w, b)(x)
    assert got.dtype == torch.float32 and got.shape == (96, N)
    e_split = (got.double() - ref).pow(2).mean().sqrt().item()
    e_fp32 = (fp32.double() - ref).pow(2).mean().sqrt().item()
    assert e_split <= 1.5 * e_fp32 + 1e-9, (e_split, e_fp32)
    # the split's own share of that error (products summed exactly): several times below an fp32 GEMM's rounding
    lin = SplitLinear(w, b)
    exact = split_planes_reference(x).double() @ lin.planes.double().t() * lin.alpha + b.double()
    assert (exact - ref).pow(2).mean().sqrt().item() <= 0.5 * e_fp32
    # without a bias, through the per-weight cache
    cache = SplitLinears()
    y1, y2 = cache(x, w), cache(x, w)
    assert torch.equal(y1, y2) and len(cache._by_weight) == 1
    assert (y1.double() - (ref - b.double())).pow(2).mean().sqrt().item() <= 1.5 * e_fp32 + 1e-9
    with pytest.raises(ValueError):
        lin(x.half())


@pytest.mark.gpu
def test_split_linear_on_the_gpu():
    """the HIP split kernel == the torch emulation bit for bit; the hipBLASLt product is fp32-grade; out-of-range activations are
    counted; the whole call is capturable"""
    from seal_amd import split_gemm
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    x = _activations(600, 1024, 3).to(dev)
    out = torch.empty(600, 3072, dtype=torch.float16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().sealnn_split_planes(torch.cuda.current_stream(dev).cuda_stream, x.data_ptr(), 600, 1024, out.data_ptr(), flag.data_ptr()))
    assert torch.equal(out.cpu(), split_planes_reference(x.cpu())) and int(flag.item()) == 0
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(4096, 1024, generator=g) * 0.05).to(dev)
    b = torch.randn(4096, generator=g).to(dev)
    ref = x.double() @ w.double().t() + b.double()
    e_fp32 = (torch.nn.functional.linear(x, w, b).double() - ref).pow(2).mean().sqrt().item()
    lin = SplitLinear(w, b)
    got = lin(x)
    assert got.dtype == torch.float32
    assert (got.double() - ref).pow(2).mean().sqrt().item() <= 2.0 * e_fp32
    assert split_gemm.overflowed(dev) == 0
    x2 = x.clone()
    x2[3, 5] = 1e5
    lin(x2)
    assert split_gemm.overflowed(dev) == 1 and split_gemm.overflowed(dev) == 0
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        lin(x)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            y = lin(x)
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, got)


def test_decoder_linear_helpers_are_f_linear_off_the_gpu():
    """BartStepDecoder._lin / _mod (what the fused paths call for every projection): plain F.linear for anything that is not an fp32
    activation on the GPU, whatever SEAL_SPLIT_GEMM says (default: on)"""
    import torch.nn.functional as F
    from seal_amd import split_gemm
    from seal_amd.bart_decoder import BartStepDecoder
    from tests.helpers import tiny_bart
    assert split_gemm.ENABLED is (os.environ.get("SEAL_SPLIT_GEMM", "1") == "1")
    m = tiny_bart(60)
    dec = BartStepDecoder(m)
    x = torch.randn(5, dec.d)
    for L in dec.layers:
        assert torch.equal(dec._lin(x, L["qkv_w"], L["qkv_b"]), F.linear(x, L["qkv_w"], L["qkv_b"]))
        for name in ("so", "cq", "co", "fc1"):
            assert torch.equal(dec._mod(x, L[name]), L[name](x))
        h = L["act"](dec._mod(x, L["fc1"]))
        assert torch.equal(dec._mod(h, L["fc2"]), L["fc2"](h))
    assert torch.equal(dec.lm_head(x), F.linear(x, dec.lm_w, dec.lm_b.view(-1)))
    if not split_gemm.ENABLED:
        assert BartStepDecoder.split_gemm is False


def test_split_policy_follows_the_measured_shapes():
    """``SplitLinears.wants``: the products that measured faster through the split on an MI355X (module text of seal_amd/split_gemm.py)"""
    w = lambda n, k: torch.empty(n, k, device="meta")
    yes = [(600, 3072, 1024, True), (600, 4096, 1024, True), (600, 1024, 4096, True), (600, 50265, 1024, True), (300, 50265, 1024, False),
           (300, 4096, 1024, True), (300, 3072, 1024, True), (3200, 1024, 1024, False), (600, 3072, 1024, False), (4096, 50265, 1024, False)]
    no = [(600, 1024, 1024, True), (300, 1024, 1024, True), (40, 50265, 1024, True), (40, 4096, 1024, True), (300, 4096, 1024, False),
          (300, 3072, 1024, False), (600, 1024, 1024, False), (600, 1022, 1022, True)]
    for rows, n, k, planes in yes:
        assert SplitLinears.wants(w(n, k), rows, planes), (rows, n, k, planes)
    for rows, n, k, planes in no:
        assert not SplitLinears.wants(w(n, k), rows, planes), (rows, n, k, planes)


def test_model_forward_through_split_linears_scores_like_fp32(monkeypatch):
    from seal_amd import split_gemm
    monkeypatch.setattr(split_gemm, "MIN_ROWS", 0)
    monkeypatch.setattr(split_gemm, "MIN_MACS_WITH_SPLIT_PASS", 0.0)
    """every nn.Linear / lm_head product of a (tiny) BART forward through the emulated split product: the running log-probability sums of
    teacher-forced hypotheses move by no more than fp32's own distance from float64 (tools/split_gemm_e2e_cpu.py does the same at BART-large
    geometry: 8.4e-6 against fp32's 5.5e-6, tolerance 1e-4)"""
    import copy
    import torch.nn.functional as F
    from tests.helpers import tiny_bart
    vocab = 200
    m = tiny_bart(vocab, d_model=128, heads=2, max_positions=64)
    g = torch.Generator().manual_seed(3)
    enc = torch.randint(4, vocab, (6, 9), generator=g)
    dec = torch.randint(4, vocab, (6, 8), generator=g)
    tgt = torch.randint(4, vocab, (6, 8), generator=g)

    def sums(model):
        with torch.no_grad():
            lp = torch.log_softmax(model(input_ids=enc, decoder_input_ids=dec).logits, dim=-1)
        fin = torch.isfinite(lp.gather(-1, tgt[..., None])[..., 0])
        return torch.where(fin, lp.gather(-1, tgt[..., None])[..., 0], torch.zeros((), dtype=lp.dtype)).cumsum(-1).double()
    a = sums(m)
    split, orig, used = SplitLinears(), F.linear, []

    def split_linear(x, w, b=None):
        if x.dtype != torch.float32 or w.shape[1] % 4:
            return orig(x, w, b)
        used.append(w.shape)
        return split(x.reshape(-1, x.shape[-1]), w, b).view(*x.shape[:-1], w.shape[0])
    torch.nn.functional.linear = split_linear
    try:
        b = sums(m)
    finally:
        torch.nn.functional.linear = orig
    ref = sums(copy.deepcopy(m).double())
    assert len(used) > 20
    e_fp32, e_split = (a - ref).abs().max().item(), (b - ref).abs().max().item()
    assert e_split <= max(3 * e_fp32, 2e-6), (e_split, e_fp32)


@pytest.mark.gpu
def test_planes_from_the_producing_kernels_on_the_gpu():
    """sealnn_add_layernorm_planes: fp32 output == sealnn_add_layernorm's, planes == the split of that output, bit for bit;
    sealnn_gelu_planes == the split of torch's gelu (erf form) to the last bit of the hi plane (erff may differ by an ulp: lo within 2^-10 of hi's ulp)"""
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(9)
    rows, d = 77, 1024
    x, y = torch.randn(rows, d, generator=g).to(dev), (torch.randn(rows, d, generator=g) * 3).to(dev)
    gamma, beta = (torch.rand(d, generator=g) + 0.5).to(dev), torch.randn(d, generator=g).to(dev)
    want = torch.empty_like(x)
    check(lib().sealnn_add_layernorm(st, x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, d, 1e-5, want.data_ptr()))
    out, planes = torch.empty_like(x), torch.empty(rows, 3 * d, dtype=torch.float16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().sealnn_add_layernorm_planes(st, x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, d, 1e-5, out.data_ptr(),
                                            planes.data_ptr(), flag.data_ptr()))
    assert torch.equal(out, want) and torch.equal(planes.cpu(), split_planes_reference(want.cpu())) and int(flag.item()) == 0
    h = (torch.randn(rows, 4096, generator=g) * 2).to(dev)
    hp = torch.empty(rows, 3 * 4096, dtype=torch.float16, device=dev)
    check(lib().sealnn_gelu_planes(st, h.data_ptr(), rows, 4096, hp.data_ptr(), flag.data_ptr()))
    ref = split_planes_reference(torch.nn.functional.gelu(h).cpu())
    got = hp.cpu()
    assert torch.equal(got[:, :4096], got[:, 4096:8192])
    back = got[:, :4096].double() + got[:, 8192:].double() * 2.0 ** -LO_SHIFT
    want_back = ref[:, :4096].double() + ref[:, 8192:].double() * 2.0 ** -LO_SHIFT
    assert (back - want_back).abs().max().item() <= 1e-6 and int(flag.item()) == 0


def test_deferred_product_is_the_finished_one():
    """``SplitLinear(...)(x, defer=True)`` hands back the raw accumulators + (alpha, bias); alpha is a power of two, so alpha * acc is exact
    and ``value()`` is the product with its epilogue applied (CPU emulation of both)"""
    import math
    g = torch.Generator().manual_seed(11)
    x = _activations(33, 256, 2)
    w, b = torch.randn(96, 256, generator=g) * 0.05, torch.randn(96, generator=g)
    lin = SplitLinear(w, b)
    d = lin(x, defer=True)
    assert math.log2(d.alpha) == int(math.log2(d.alpha)) and d.acc.dtype == torch.float32 and d.bias is lin.bias
    assert torch.allclose(d.value(), lin(x), rtol=1e-6, atol=1e-6)
    assert torch.equal(lin.from_planes(split_planes_reference(x), defer=True).acc, d.acc)


@pytest.mark.gpu
def test_kernels_that_apply_the_epilogue_themselves_on_the_gpu(monkeypatch):
    """sealnn_*_acc(raw accumulators, bias, alpha) == the plain kernel on alpha * acc + bias, bit for bit -- outputs, caches, planes; and a
    tree forward of the decoder with the epilogues deferred == the same with torch.addmm's (split_gemm.DEFER_EPILOGUE off) to 1e-4"""
    from seal_amd import split_gemm
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(13)
    alpha = 2.0 ** -13
    rows, H, T, d = 45, 16, 9, 1024
    # self-attention step: q / k / v read from the accumulators, k / v written to the caches
    acc = (torch.randn(rows, 3 * d, generator=g) * 3000).to(dev)
    bias = torch.randn(3 * d, generator=g).to(dev)
    qkv = acc * alpha + bias
    t = torch.tensor([4], dtype=torch.int64, device=dev)
    caches = [(torch.randn(rows, H, T, 64, generator=g)).to(dev) for _ in range(2)]
    anc = torch.randint(0, rows, (T, rows), generator=g, dtype=torch.int32).to(dev)
    outs = []
    for use_acc in (False, True):
        kc, vc, an, out = caches[0].clone(), caches[1].clone(), anc.clone(), torch.empty(rows, d, device=dev)
        if use_acc:
            check(lib().sealnn_self_attn_step_acc(st, acc.data_ptr(), bias.data_ptr(), alpha, kc.data_ptr(), vc.data_ptr(), t.data_ptr(), rows, H, T,
                                                  0.125, out.data_ptr(), an.data_ptr()))
        else:
            check(lib().sealnn_self_attn_step(st, qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), t.data_ptr(), rows, H, T, 0.125, out.data_ptr(),
                                              an.data_ptr()))
        outs.append((out, kc, vc, an))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # tree self-attention
    n, A = 60, 5
    tacc = (torch.randn(n, 3 * d, generator=g) * 3000).to(dev)
    tqkv = tacc * alpha + bias
    depth = torch.randint(0, A, (n,), generator=g)
    tanc = torch.full((n, A), -1, dtype=torch.int32)
    for i in range(n):
        for j in range(int(depth[i])):
            tanc[i, j] = int(torch.randint(0, n, (1,), generator=g))
        tanc[i, int(depth[i])] = i
    tanc = tanc.to(dev)
    o1, o2 = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
    check(lib().sealnn_tree_self_attn(st, tqkv.data_ptr(), tanc.data_ptr(), n, A, H, 0.125, o1.data_ptr()))
    check(lib().sealnn_tree_self_attn_acc(st, tacc.data_ptr(), bias.data_ptr(), alpha, tanc.data_ptr(), n, A, H, 0.125, o2.data_ptr()))
    assert torch.equal(o1, o2)
    # add + LayerNorm (+ planes), gelu planes
    x = torch.randn(rows, d, generator=g).to(dev)
    yacc, yb = (torch.randn(rows, d, generator=g) * 20000).to(dev), torch.randn(d, generator=g).to(dev)
    y = yacc * alpha + yb
    gamma, beta = (torch.rand(d, generator=g) + 0.5).to(dev), torch.randn(d, generator=g).to(dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    w1, p1, w2, p2 = torch.empty_like(x), torch.empty(rows, 3 * d, dtype=torch.float16, device=dev), torch.empty_like(x), torch.empty(rows, 3 * d, dtype=torch.float16, device=dev)
    check(lib().sealnn_add_layernorm_planes(st, x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, d, 1e-5, w1.data_ptr(), p1.data_ptr(), flag.data_ptr()))
    check(lib().sealnn_add_layernorm_acc(st, x.data_ptr(), yacc.data_ptr(), yb.data_ptr(), alpha, gamma.data_ptr(), beta.data_ptr(), rows, d, 1e-5, w2.data_ptr(),
                                         p2.data_ptr(), flag.data_ptr()))
    assert torch.equal(w1, w2) and torch.equal(p1, p2)
    w3 = torch.empty_like(x)
    check(lib().sealnn_add_layernorm_acc(st, x.data_ptr(), yacc.data_ptr(), yb.data_ptr(), alpha, gamma.data_ptr(), beta.data_ptr(), rows, d, 1e-5, w3.data_ptr(),
                                         None, None))
    assert torch.equal(w1, w3)
    hacc, hb = (torch.randn(rows, 4096, generator=g) * 15000).to(dev), torch.randn(4096, generator=g).to(dev)
    h = hacc * alpha + hb
    g1, g2 = torch.empty(rows, 3 * 4096, dtype=torch.float16, device=dev), torch.empty(rows, 3 * 4096, dtype=torch.float16, device=dev)
    check(lib().sealnn_gelu_planes(st, h.data_ptr(), rows, 4096, g1.data_ptr(), flag.data_ptr()))
    check(lib().sealnn_gelu_planes_acc(st, hacc.data_ptr(), hb.data_ptr(), alpha, rows, 4096, g2.data_ptr(), flag.data_ptr()))
    assert torch.equal(g1, g2) and int(flag.item()) == 0
    # the decoder's tree forward at a size where every projection goes through the split: deferred against addmm's epilogue
    from transformers import BartConfig, BartForConditionalGeneration
    from seal_amd.bart_decoder import BartStepDecoder
    torch.manual_seed(0)
    cfg = BartConfig(vocab_size=4000, d_model=1024, encoder_layers=1, decoder_layers=2, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=4096, decoder_ffn_dim=4096, max_position_embeddings=64)
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg).eval()
    dec = BartStepDecoder(model)
    B, S, N, A = 4, 12, 3200, 6
    ids = torch.randint(3, 4000, (B, S), generator=g).to(dev)
    mask = torch.ones(B, S, dtype=torch.long, device=dev)
    enc = dec.encode(ids, mask)
    prepared = dec.teacher_prepare(enc, mask)
    tok = torch.randint(3, 4000, (N,), generator=g).to(dev)
    depth = (torch.arange(N) % A)
    anc = torch.full((N, A), -1, dtype=torch.long)
    for i in range(N):
        k = int(depth[i])
        anc[i, :k + 1] = torch.arange(i - k, i + 1)
    qidx = (torch.arange(N) // (N // B)).clamp(max=B - 1).to(dev)
    res = []
    for defer in (True, False):
        monkeypatch.setattr(split_gemm, "DEFER_EPILOGUE", defer)
        res.append(dec.tree_logits(tok, depth.to(dev), anc.to(dev), qidx, enc, mask, prepared=prepared))
    # (the library may pick another GEMM kernel for a product without an epilogue: same arithmetic, another summation order)
    assert torch.isfinite(res[0]).all() and (res[0] - res[1]).abs().max().item() <= 1e-4 and split_gemm.overflowed(dev) == 0


def test_hand_configuration_tables_are_well_formed():
    """``split_gemm.HAND_CONFIGS`` / ``HAND_CONFIGS_PAIRS``: row tiers ascending, every configuration one ``sealnn_hgemm_nt`` accepts -- a known tile, stages that
    fit its LDS, K groups the tile has, split-K slices that divide the K steps of the form (3K / 64, resp. 2K / 64 for pairs) into whole K groups, and no more
    slabs than the kernels that add them take (16); ``hand_config`` answers the first tier that holds the rows, None beyond the last and with HAND_GEMM off"""
    from seal_amd import split_gemm
    for table, pairs in ((split_gemm.HAND_CONFIGS, False), (split_gemm.HAND_CONFIGS_PAIRS, True)):
        assert table
        for (n, k3), tiers in table.items():
            assert k3 % 3 == 0 and [r for r, _ in tiers] == sorted({r for r, _ in tiers}), (n, k3)
            steps = (k3 // 3 * 2 if pairs else k3) // 64
            for rows, cfg in tiers:
                tile, stages, kg, slices = cfg & 0x7f, (cfg >> 8) & 0xf, (cfg >> 12) & 0xf, (cfg >> 16) & 0x1fff
                assert cfg >> 29 == 0 and tile in (1, 2, 3, 4, 5, 6, 7) and 1 <= slices <= 16, (n, k3, rows, hex(cfg))
                assert steps % slices == 0 and (steps // slices) % kg == 0, (n, k3, rows, hex(cfg))
                if tile >= 5:
                    assert kg == 1 and 2 <= stages <= (2 if tile == 5 else 3)
                else:
                    assert (kg == 1 and 1 <= stages <= 3) or (stages == 2 and kg <= {1: 1, 2: 4, 3: 2, 4: 2}[tile])
                assert split_gemm.hand_config(rows, n, k3, pairs) == cfg and split_gemm.hand_config(1, n, k3, pairs) == tiers[0][1]
            assert split_gemm.hand_config(tiers[-1][0] + 1, n, k3, pairs) is None
    try:
        split_gemm.HAND_GEMM = False
        assert split_gemm.hand_config(600, 1024, 3072) is None and split_gemm.hand_config(600, 1024, 3072, True) is None
    finally:
        split_gemm.HAND_GEMM = True


def test_pair_planes_of_a_weight_interleave_its_hi_and_lo_planes():
    """``SplitLinear.pair_planes``: [N, 2K], per 32 columns the hi plane's 32 values then the lo plane's (blocks 0 and 1 of the three-block planes; block 2 is
    block 0 times 2^-11 and is not carried); a K that 32 does not divide is refused"""
    from seal_amd import split_gemm
    g = torch.Generator().manual_seed(2)
    w = torch.randn(24, 96, generator=g) * 0.05
    lin = split_gemm.SplitLinear(w, torch.zeros(24))
    wp = lin.pair_planes()
    assert wp.shape == (24, 192) and wp.dtype == torch.float16 and wp is lin.pair_planes()
    v = wp.view(24, 3, 2, 32)
    assert torch.equal(v[:, :, 0].reshape(24, 96), lin.planes[:, :96]) and torch.equal(v[:, :, 1].reshape(24, 96), lin.planes[:, 96:192])
    assert torch.equal((lin.planes[:, :96].float() * 2.0 ** -11).half(), lin.planes[:, 192:])
    with pytest.raises(ValueError):
        split_gemm.SplitLinear(torch.randn(8, 48, generator=g), None).pair_planes()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(600, 1024, 3072), (300, 1024, 12288), (37, 200, 192), (640, 192, 256), (321, 4096 + 40, 512)])
def test_hand_written_gemm_against_torch(M, N, K):
    """``sealnn_hgemm_nt`` (hgemm_kernels.hip: LDS-DMA staging, swizzled LDS, MFMA 16x16x32 f16) in every instantiated configuration -- tiles,
    LDS stages, K groups, split-K slabs: EXACT on one-hot operands (a permuted fragment or a transposed tile cannot pass), within fp32
    accumulation noise of torch's product on random ones; ragged heights / widths (rows beyond M, N are clamped on load, not stored)."""
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = torch.randn(N, K, generator=g).half().to(dev)
    ref = torch.mm(a.float(), w.float().t())
    a1 = torch.zeros(M, K, dtype=torch.float16, device=dev)
    a1[torch.arange(M, device=dev), (torch.arange(M, device=dev) * 7 + 3) % K] = 1.0
    ref1 = torch.mm(a1.float(), w.float().t())
    n_cfg = 0
    for tile in (1, 2, 3, 4):
        for stages, kg in ((1, 1), (2, 1), (3, 1), (2, 2), (2, 4)):
            if kg > {1: 1, 2: 4, 3: 2, 4: 2}[tile]:
                continue
            for slices in (1, 2, 4):
                if (K // 64) % slices or (K // 64 // slices) % kg:
                    continue
                cfg = tile | (stages << 8) | (kg << 12) | (slices << 16)
                for x, want, exact in ((a, ref, False), (a1, ref1, True)):
                    c = torch.full((slices, M, N), float("nan"), device=dev)
                    check(lib().sealnn_hgemm_nt(st, x.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, N, cfg))
                    got = c.sum(0)
                    if exact:
                        assert torch.equal(got, want), (tile, stages, kg, slices)
                    else:
                        assert float((got - want).abs().max() / want.abs().max()) < 1e-4, (tile, stages, kg, slices)
                n_cfg += 1
    # the tall tiles (320 x 128 / 64 / 96, eight waves, hand-counted waits on inline-assembly fragment reads; 96: a ragged last piece per stage)
    for tile, stage_opts in ((5, (2,)), (6, (2, 3)), (7, (2, 3))):
        for stages in stage_opts:
            for slices in (1, 2, 4):
                if (K // 64) % slices:
                    continue
                cfg = tile | (stages << 8) | (1 << 12) | (slices << 16)
                for x, want, exact in ((a, ref, False), (a1, ref1, True)):
                    c = torch.full((slices, M, N), float("nan"), device=dev)
                    check(lib().sealnn_hgemm_nt(st, x.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, N, cfg))
                    got = c.sum(0)
                    if exact:
                        assert torch.equal(got, want), (tile, stages, slices)
                    else:
                        assert float((got - want).abs().max() / want.abs().max()) < 1e-4, (tile, stages, slices)
                n_cfg += 1
    with pytest.raises(Exception):
        check(lib().sealnn_hgemm_nt(st, a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, N, 5 | (3 << 8) | (1 << 12) | (1 << 16)))    # 3 stages of 320 x 128 do not fit the LDS
    c = torch.empty(M, N, device=dev)
    check(lib().sealnn_hgemm_nt(st, a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, N, 0))          # the shape-picked configuration
    assert float((c - ref).abs().max() / ref.abs().max()) < 1e-4 and n_cfg >= 8
    with pytest.raises(Exception):
        check(lib().sealnn_hgemm_nt(st, a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, 100, N, 0))     # K not a multiple of 64


@pytest.mark.gpu
def test_gelu_adds_the_slabs_of_a_split_k_fc1():
    """``sealnn_gelu_planes_acc_slabs`` (fc1 as a split-K product of the tall tile: GELU adds the slabs as it reads them) == ``sealnn_gelu_planes_acc``
    on the slabs summed beforehand in the same order, bit for bit"""
    from seal_amd._lib import check, lib
    from seal_amd import split_gemm
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(5)
    rows, d = 77, 4096
    for n_slabs in (1, 2, 3, 16):
        slabs = (torch.randn(n_slabs, rows, d, generator=g) * 3).to(dev)
        bias = torch.randn(d, generator=g).to(dev)
        tot = slabs[0].clone()
        for s_ in range(1, n_slabs):
            tot = tot + slabs[s_]
        want = torch.empty(rows, 3 * d, dtype=torch.float16, device=dev)
        got = torch.empty_like(want)
        flag = split_gemm._flag(dev).data_ptr()
        check(lib().sealnn_gelu_planes_acc(st, tot.data_ptr(), bias.data_ptr(), 0.5, rows, d, want.data_ptr(), flag))
        check(lib().sealnn_gelu_planes_acc_slabs(st, slabs.data_ptr(), n_slabs, slabs.stride(0), bias.data_ptr(), 0.5, rows, d, got.data_ptr(), flag))
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), n_slabs
    with pytest.raises(Exception):
        check(lib().sealnn_gelu_planes_acc_slabs(st, slabs.data_ptr(), 17, slabs.stride(0), bias.data_ptr(), 0.5, rows, d, got.data_ptr(), flag))


@pytest.mark.gpu
def test_pair_planes_and_the_three_products_of_a_k_step():
    """the PAIRS operand (hi / lo per 32 columns, [rows, 2K]): every kernel that writes planes writes the same 16-bit values in its _pairs form as in the
    three-block form; ``sealnn_hgemm_nt`` with the PAIRS bit on those operands == the three-block product -- EXACTLY on one-hot activations (a misplaced
    column or plane cannot pass), to fp32 summation noise on random ones, in 4-wave and tall tiles, with split-K slabs -- and == fp64 ``F.linear`` to
    split-GEMM accuracy"""
    from seal_amd import split_gemm
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(21)
    flag = split_gemm._flag(dev).data_ptr()

    def as_pairs(three, k):
        hi, lo = three[:, :k].reshape(three.shape[0], -1, 32), three[:, 2 * k:].reshape(three.shape[0], -1, 32)
        assert torch.equal(three[:, :k], three[:, k:2 * k])
        return torch.stack((hi, lo), 2).reshape(three.shape[0], 2 * k)
    # the writers: split_planes, gelu, add + LayerNorm (behind slabs)
    rows, d = 77, 1024
    x = (torch.randn(rows, d, generator=g) * 3).to(dev)
    three, two = torch.empty(rows, 3 * d, dtype=torch.float16, device=dev), torch.empty(rows, 2 * d, dtype=torch.float16, device=dev)
    check(lib().sealnn_split_planes(st, x.data_ptr(), rows, d, three.data_ptr(), flag))
    check(lib().sealnn_split_planes_pairs(st, x.data_ptr(), rows, d, two.data_ptr(), flag))
    assert torch.equal(as_pairs(three, d).view(torch.int16), two.view(torch.int16))
    slabs, bias = (torch.randn(3, rows, d, generator=g)).to(dev), torch.randn(d, generator=g).to(dev)
    check(lib().sealnn_gelu_planes_acc_slabs(st, slabs.data_ptr(), 3, slabs.stride(0), bias.data_ptr(), 0.5, rows, d, three.data_ptr(), flag))
    check(lib().sealnn_gelu_planes_acc_slabs_pairs(st, slabs.data_ptr(), 3, slabs.stride(0), bias.data_ptr(), 0.5, rows, d, two.data_ptr(), flag))
    assert torch.equal(as_pairs(three, d).view(torch.int16), two.view(torch.int16))
    gamma, beta = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    o3, o2 = torch.empty(rows, d, device=dev), torch.empty(rows, d, device=dev)
    check(lib().sealnn_add_layernorm_acc_slabs(st, x.data_ptr(), slabs.data_ptr(), 3, slabs.stride(0), bias.data_ptr(), 0.5, gamma.data_ptr(), beta.data_ptr(),
                                               rows, d, 1e-5, o3.data_ptr(), three.data_ptr(), flag))
    check(lib().sealnn_add_layernorm_acc_slabs_pairs(st, x.data_ptr(), slabs.data_ptr(), 3, slabs.stride(0), bias.data_ptr(), 0.5, gamma.data_ptr(),
                                                     beta.data_ptr(), rows, d, 1e-5, o2.data_ptr(), two.data_ptr(), flag))
    assert torch.equal(o3, o2) and torch.equal(as_pairs(three, d).view(torch.int16), two.view(torch.int16))
    # the product
    for rows, n, k in ((600, 1024, 1024), (300, 4096, 1024), (37, 200, 256), (321, 1024, 4096)):
        w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
        lin = split_gemm.SplitLinear(w, torch.zeros(n, device=dev))
        wp = lin.pair_planes()
        assert torch.equal(wp.view(n, -1, 2, 32)[:, :, 0].reshape(n, k), lin.planes[:, :k]) and torch.equal(wp.view(n, -1, 2, 32)[:, :, 1].reshape(n, k), lin.planes[:, k:2 * k])
        x = torch.randn(rows, k, generator=g).to(dev)
        x1 = torch.zeros(rows, k, device=dev)
        x1[torch.arange(rows, device=dev), (torch.arange(rows, device=dev) * 7 + 3) % k] = 1.0
        ref = torch.nn.functional.linear(x.double(), w.double()).float()
        n_cfg = 0
        for xx, exact in ((x, False), (x1, True)):
            a3, a2 = torch.empty(rows, 3 * k, dtype=torch.float16, device=dev), torch.empty(rows, 2 * k, dtype=torch.float16, device=dev)
            check(lib().sealnn_split_planes(st, xx.data_ptr(), rows, k, a3.data_ptr(), flag))
            check(lib().sealnn_split_planes_pairs(st, xx.data_ptr(), rows, k, a2.data_ptr(), flag))
            c3 = torch.empty(rows, n, device=dev)
            check(lib().sealnn_hgemm_nt(st, a3.data_ptr(), lin.planes.data_ptr(), c3.data_ptr(), rows, n, 3 * k, n, 2 | (2 << 8) | (1 << 12) | (1 << 16)))
            for tile, stages, kg in ((1, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 4), (3, 3, 1), (4, 2, 2), (129, 2, 1), (5, 2, 1), (6, 3, 1), (7, 3, 1), (7, 2, 1)):
                for slices in (1, 2, 4):
                    if (2 * k // 64) % slices or (2 * k // 64 // slices) % kg:
                        continue
                    c = torch.full((slices, rows, n), float("nan"), device=dev)
                    check(lib().sealnn_hgemm_nt(st, a2.data_ptr(), wp.data_ptr(), c.data_ptr(), rows, n, 2 * k, n,
                                                tile | (stages << 8) | (kg << 12) | (slices << 16) | split_gemm.PAIRS_BIT))
                    got = c.sum(0)
                    if exact:
                        assert torch.equal(got, c3), (tile, stages, kg, slices)
                    else:
                        assert float((got - c3).abs().max()) <= 6e-6 * float(c3.abs().max()), (tile, stages, kg, slices)
                        assert float((got * lin.alpha - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
                    n_cfg += 1
        assert n_cfg >= 16
    assert split_gemm.overflowed(dev) == 0


@pytest.mark.gpu
def test_a_finished_product_through_the_hand_written_kernel():
    """a product nobody defers (the encoder's q / k / v for torch's attention, the cross-attention K / V) at a height the hand-written kernel has a
    configuration for: ``sealnn_hgemm_nt`` + ``sealnn_finish_product`` == ``Deferred.value()`` of the same slabs bit for bit, no library GEMM is
    issued, and the result is the fp32 ``F.linear`` to split-GEMM accuracy; ``HAND_GEMM`` off: the library's addmm, same accuracy"""
    from seal_amd import split_gemm
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    for rows, n, k in ((320, 3072, 1024), (300, 2048, 1024), (600, 2048, 1024), (77, 4096, 1024)):
        w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        x = torch.randn(rows, k, generator=g).to(dev)
        lin = split_gemm.SplitLinear(w, b)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double()).float()
        before = split_gemm.LIBRARY_GEMMS[0]
        got = lin(x)
        assert split_gemm.LIBRARY_GEMMS[0] == before and split_gemm.hand_config(rows, n, 3 * k) is not None
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        d = lin(x, defer=True, slabs_ok=True)
        assert isinstance(d, split_gemm.Deferred) and torch.equal(d.value(), got)
        try:
            split_gemm.HAND_GEMM = False
            lib_out = lin(x)
        finally:
            split_gemm.HAND_GEMM = True
        assert split_gemm.LIBRARY_GEMMS[0] == before + 1 and float((lib_out - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    st = torch.cuda.current_stream(dev).cuda_stream
    with pytest.raises(Exception):
        check(lib().sealnn_finish_product(st, got.data_ptr(), 1, 0, b.data_ptr(), 1.0, 4, 6, got.data_ptr()))       # n not a multiple of 4


@pytest.mark.gpu
def test_split_k_slabs_are_summed_by_the_kernel_that_reads_them(monkeypatch):
    """the decode step's fc2 product through the hand-written kernel: 4 split-K slabs that ``sealnn_add_layernorm_acc_slabs`` adds in slab
    order as it reads them == ``sealnn_add_layernorm_acc`` on the slabs summed beforehand in the same order, bit for bit (outputs and planes);
    and the graph-captured step decoder with the hand-written product == the same with the library's (``split_gemm.HAND_GEMM`` off) to 1e-5"""
    from seal_amd import split_gemm
    from seal_amd._lib import check, lib
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(17)
    rows, d, S = 77, 1024, 4
    alpha = 2.0 ** -12
    x = torch.randn(rows, d, generator=g).to(dev)
    slabs = (torch.randn(S, rows, d, generator=g) * 9000).to(dev)
    tot = slabs[0]
    for s in range(1, S):
        tot = tot + slabs[s]
    yb = torch.randn(d, generator=g).to(dev)
    gamma, beta = (torch.rand(d, generator=g) + 0.5).to(dev), torch.randn(d, generator=g).to(dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    o1, p1, o2, p2 = torch.empty_like(x), torch.empty(rows, 3 * d, dtype=torch.float16, device=dev), torch.empty_like(x), torch.empty(rows, 3 * d, dtype=torch.float16, device=dev)
    check(lib().sealnn_add_layernorm_acc(st, x.data_ptr(), tot.contiguous().data_ptr(), yb.data_ptr(), alpha, gamma.data_ptr(), beta.data_ptr(), rows, d, 1e-5,
                                         o1.data_ptr(), p1.data_ptr(), flag.data_ptr()))
    check(lib().sealnn_add_layernorm_acc_slabs(st, x.data_ptr(), slabs.data_ptr(), S, rows * d, yb.data_ptr(), alpha, gamma.data_ptr(), beta.data_ptr(), rows, d,
                                               1e-5, o2.data_ptr(), p2.data_ptr(), flag.data_ptr()))
    assert torch.equal(o1, o2) and torch.equal(p1, p2)
    # the attention step kernels between two hand-written products: slabs in, split planes out == the plain kernel on the finished operand,
    # then sealnn_split_planes over its output
    H, T, Bq, Kb, S_enc = 16, 9, 5, 9, 24
    R = Bq * Kb
    qs = (torch.randn(S, R, 3 * d, generator=g) * 3000).to(dev)
    qb = torch.randn(3 * d, generator=g).to(dev)
    qtot = qs[0]
    for s in range(1, S):
        qtot = qtot + qs[s]
    qkv = qtot * alpha + qb
    t = torch.tensor([4], dtype=torch.int64, device=dev)
    caches = [(torch.randn(R, H, T, 64, generator=g)).to(dev) for _ in range(2)]
    anc = torch.randint(0, R, (T, R), generator=g, dtype=torch.int32).to(dev)
    k1, v1, a1, o1 = caches[0].clone(), caches[1].clone(), anc.clone(), torch.empty(R, d, device=dev)
    check(lib().sealnn_self_attn_step(st, qkv.data_ptr(), k1.data_ptr(), v1.data_ptr(), t.data_ptr(), R, H, T, 0.125, o1.data_ptr(), a1.data_ptr()))
    pl1 = torch.empty(R, 3 * d, dtype=torch.float16, device=dev)
    check(lib().sealnn_split_planes(st, o1.data_ptr(), R, d, pl1.data_ptr(), flag.data_ptr()))
    k2, v2, a2, o2, pl2 = caches[0].clone(), caches[1].clone(), anc.clone(), torch.empty(R, d, device=dev), torch.empty(R, 3 * d, dtype=torch.float16, device=dev)
    check(lib().sealnn_self_attn_step_x(st, qs.data_ptr(), S, R * 3 * d, qb.data_ptr(), alpha, k2.data_ptr(), v2.data_ptr(), t.data_ptr(), R, H, T, 0.125,
                                        o2.data_ptr(), pl2.data_ptr(), flag.data_ptr(), a2.data_ptr()))
    assert torch.equal(o1, o2) and torch.equal(pl1, pl2) and torch.equal(k1, k2) and torch.equal(v1, v2) and torch.equal(a1, a2)
    ck = torch.randn(Bq, H, 64, S_enc, generator=g).to(dev)
    cv = torch.randn(Bq, H, S_enc, 64, generator=g).to(dev)
    cb = torch.zeros(Bq, S_enc, device=dev)
    cb[:, -3:] = torch.finfo(torch.float32).min
    cq_s = (torch.randn(S, R, d, generator=g) * 3000).to(dev)
    cqb = torch.randn(d, generator=g).to(dev)
    ctot = cq_s[0]
    for s in range(1, S):
        ctot = ctot + cq_s[s]
    cqv = (ctot * alpha + cqb).contiguous()
    c1, c2, cp1, cp2 = torch.empty(R, d, device=dev), torch.empty(R, d, device=dev), torch.empty(R, 3 * d, dtype=torch.float16, device=dev), torch.empty(R, 3 * d, dtype=torch.float16, device=dev)
    check(lib().sealnn_cross_attn_step(st, cqv.data_ptr(), ck.data_ptr(), cv.data_ptr(), cb.data_ptr(), Bq, Kb, H, S_enc, 0.125, c1.data_ptr()))
    check(lib().sealnn_split_planes(st, c1.data_ptr(), R, d, cp1.data_ptr(), flag.data_ptr()))
    check(lib().sealnn_cross_attn_step_x(st, cq_s.data_ptr(), S, R * d, cqb.data_ptr(), alpha, ck.data_ptr(), cv.data_ptr(), cb.data_ptr(), Bq, Kb, H, S_enc, 0.125,
                                         c2.data_ptr(), cp2.data_ptr(), flag.data_ptr()))
    assert torch.equal(c1, c2) and torch.equal(cp1, cp2) and int(flag.item()) == 0
    # the step decoder: the hand-written products (d x d projections, qkv, fc2) against the library's
    from transformers import BartConfig, BartForConditionalGeneration
    from seal_amd.bart_decoder import BartStepDecoder
    torch.manual_seed(0)
    cfg = BartConfig(vocab_size=3000, d_model=1024, encoder_layers=1, decoder_layers=2, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=4096, decoder_ffn_dim=4096, max_position_embeddings=64)
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg).eval()
    B, K, S_in, T = 20, 15, 12, 4            # (300 rows: every product of the step is a split product, so the PAIRS form applies)
    ids = torch.randint(3, 3000, (B, S_in), generator=g).to(dev)
    mask = torch.ones(B, S_in, dtype=torch.long, device=dev)
    outs, used, paired = [], [], []
    for hand, pairs in ((True, True), (True, False), (False, False)):         # (pairs: every plane of the step as hi / lo pairs, three products per K step)
        monkeypatch.setattr(split_gemm, "HAND_GEMM", hand)
        monkeypatch.setattr(split_gemm, "PAIRS", pairs)
        dec = BartStepDecoder(model)
        enc = dec.encode(ids, mask)
        dec.start(enc, mask, K, T)
        tok = torch.full((B * K,), 2, dtype=torch.long, device=dev)
        steps = []
        for t in range(T - 1):
            lg = dec.step(tok).clone()
            steps.append(lg)
            tok = lg.argmax(-1) if not outs else outs[0][t].argmax(-1)            # (the same tokens in every arm)
        outs.append(torch.stack(steps))
        used.append(split_gemm.hand_config(B * K, 1024, 3 * 4096) is not None)
        paired.append(any(getattr(z, "pairs", False) for z in dec._static_cache.values()))
    assert used == [True, True, False] and paired == [True, False, False]
    assert torch.isfinite(outs[0]).all() and split_gemm.overflowed(dev) == 0
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-4 and (outs[0] - outs[2]).abs().max().item() <= 2e-4


@pytest.mark.gpu
def test_shared_first_step_through_the_hand_written_kernel(monkeypatch):
    """the shared first step of a decode (``batch`` rows: 40 in the bench) as pair planes through ``sealnn_hgemm_nt`` -- no library GEMM but, for a
    vocabulary without a configuration, the output projection -- against the same step through the library's fp32 GEMMs and against HF's forward:
    logits of the first step and of the steps after it (which read the cache slot it wrote) within 2e-4 / 1e-4"""
    from transformers import BartConfig, BartForConditionalGeneration
    from seal_amd import split_gemm
    from seal_amd.bart_decoder import BartStepDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = BartConfig(vocab_size=3000, d_model=1024, encoder_layers=1, decoder_layers=2, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=4096, decoder_ffn_dim=4096, max_position_embeddings=64)
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg).eval()
    g = torch.Generator().manual_seed(9)
    B, K, S_in, T = 40, 15, 12, 4
    ids = torch.randint(3, 3000, (B, S_in), generator=g).to(dev)
    mask = torch.ones(B, S_in, dtype=torch.long, device=dev)
    mask[3, 8:] = 0
    outs, lib_gemms = [], []
    for by_hand in (True, False):
        monkeypatch.setattr(BartStepDecoder, "first_step_by_hand", by_hand)
        dec = BartStepDecoder(model)
        enc = dec.encode(ids, mask)
        dec.start(enc, mask, K, T)
        before = split_gemm.LIBRARY_GEMMS[0]
        assert dec._first_step_by_hand(enc.view(-1, 1024)[:B], B) is by_hand
        tok = torch.full((B * K,), 2, dtype=torch.long, device=dev)
        steps = []
        for t in range(T - 1):
            lg = dec.step(tok, beams_identical=(t == 0)).clone()
            steps.append(lg)
            tok = lg.argmax(-1) if not outs else outs[0][t].argmax(-1)
        outs.append(torch.stack(steps))
        lib_gemms.append(split_gemm.LIBRARY_GEMMS[0] - before)
    assert torch.isfinite(outs[0]).all() and (outs[0] - outs[1]).abs().max().item() <= 2e-4 and split_gemm.overflowed(dev) == 0
    # HF's cache-free forward on the tokens the first arm chose
    rows = torch.full((B * K, 1), 2, dtype=torch.long, device=dev)
    ids_rep, am_rep = ids.repeat_interleave(K, 0), mask.repeat_interleave(K, 0)
    with torch.no_grad():
        for t in range(T - 1):
            want = model(input_ids=ids_rep, attention_mask=am_rep, decoder_input_ids=rows).logits[:, -1, :]
            assert (outs[0][t] - want).abs().max().item() <= 1e-4, t
            rows = torch.cat([rows, outs[0][t].argmax(-1)[:, None]], 1)


@pytest.mark.gpu
def test_output_projection_finished_in_the_store(monkeypatch):
    """a hand step's output projection with alpha, final_logits_bias and the per-query logit bias applied in the kernel's store
    (``sealnn_hgemm_nt_ep``) == the raw accumulators + ``torch.add`` behind them, BIT FOR BIT (the same accumulators, the same one rounding), through a
    change of the logit bias between two decodes, with -inf entries, at BART-large's vocabulary and width"""
    from transformers import BartConfig, BartForConditionalGeneration
    from seal_amd import split_gemm
    from seal_amd.bart_decoder import BartStepDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = BartConfig(vocab_size=50265, d_model=1024, encoder_layers=1, decoder_layers=1, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=4096, decoder_ffn_dim=4096, max_position_embeddings=64)
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg).eval()
    g = torch.Generator().manual_seed(4)
    B, K, S_in, T = 20, 15, 10, 4
    ids = torch.randint(3, 50265, (B, S_in), generator=g).to(dev)
    mask = torch.ones(B, S_in, dtype=torch.long, device=dev)
    biases = []
    for _ in range(2):
        lb = torch.randn(B, 50265, generator=g)
        lb[:, 1] = float("-inf")
        lb[3, 100:200] = float("-inf")
        biases.append(lb.to(dev))
    outs = []
    for in_store in (True, False):
        monkeypatch.setattr(BartStepDecoder, "lm_head_finished_in_store", in_store)
        dec = BartStepDecoder(model)
        got = []
        for lb in (biases[0], biases[1], None):
            enc = dec.encode(ids, mask)
            dec.start(enc, mask, K, T)
            dec.logit_bias = lb
            tok = torch.full((B * K,), 2, dtype=torch.long, device=dev)
            for t in range(T - 1):
                lg = dec.step(tok, beams_identical=(t == 0)).clone()
                got.append(lg)
                tok = torch.where(torch.isfinite(lg), lg, torch.full_like(lg, -1e30)).argmax(-1) if not outs else outs[0][len(got) - 1][1]
                got[-1] = (lg, tok)
        assert (getattr(dec._st, "lm_epilogue", None) == "in the store") is in_store
        outs.append(got)
    for (a, _), (b, _) in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert split_gemm.overflowed(dev) == 0


@pytest.mark.gpu
def test_encoder_through_the_split_gemm_matches_hf_encoder(monkeypatch):
    """``BartStepDecoder.encode`` with the encoder's linear layers through the split GEMM (the layer written out: q / k / v as one product, torch's
    fused attention, LayerNorms) against HF's own ``BartEncoder`` forward in fp32, at BART-large width with padded inputs: within 2e-5 on O(1)
    activations (the split's error is below an fp32 GEMM's own); with the switch off the HF layers run and the output is HF's."""
    from transformers import BartConfig, BartForConditionalGeneration
    from seal_amd import bart_decoder, split_gemm
    from seal_amd.bart_decoder import BartStepDecoder
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = BartConfig(vocab_size=3000, d_model=1024, encoder_layers=2, decoder_layers=1, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=4096, decoder_ffn_dim=4096, max_position_embeddings=64)
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg).eval()
    g = torch.Generator().manual_seed(5)
    B, S = 60, 30
    ids = torch.randint(3, 3000, (B, S), generator=g).to(dev)
    lens = torch.randint(8, S + 1, (B,), generator=g).to(dev)
    mask = (torch.arange(S, device=dev)[None, :] < lens[:, None]).long()
    ids = torch.where(mask.bool(), ids, torch.ones_like(ids))
    with torch.no_grad():
        ref = model.model.encoder(input_ids=ids, attention_mask=mask).last_hidden_state
    dec = BartStepDecoder(model)
    monkeypatch.setattr(bart_decoder, "ENCODER_SPLIT", True)
    a = dec.encode(ids, mask)
    monkeypatch.setattr(bart_decoder, "ENCODER_SPLIT", False)
    b = dec.encode(ids, mask)
    valid = mask.bool()
    assert torch.isfinite(a[valid]).all() and float((a[valid] - ref[valid]).abs().max()) < 2e-5, float((a[valid] - ref[valid]).abs().max())
    assert float((b[valid] - ref[valid]).abs().max()) < 1e-5 and split_gemm.overflowed(dev) == 0


def test_weight_cache_accepts_inference_tensors():
    """``SplitLinears._of`` validates its cache entry by the weight's version counter; a tensor made under ``torch.inference_mode``
    (the fused q/k/v weights the rescoring entry points concatenate there) has none -- reading ``_version`` raises -- and is taken
    as never modified instead (round-5 advisor finding, reproduced on CPU)."""
    from seal_amd.split_gemm import SplitLinears, tensor_version
    with torch.inference_mode():
        w = torch.cat([torch.randn(64, 32), torch.randn(64, 32)], 0)
        b = torch.zeros(128)
        assert w.is_inference() and tensor_version(w) == -1
        sl = SplitLinears()
        first = sl._of(w, b)
        assert sl._of(w, b) is first                       # a hit, not a rebuild and not a RuntimeError
    p = torch.nn.Parameter(torch.randn(64, 32))
    v0 = tensor_version(p)
    with torch.no_grad():
        p.add_(1.0)
    assert tensor_version(p) == v0 + 1


@pytest.mark.gpu
def test_stand_alone_rescore_keys_at_bart_width_inside_inference_mode():
    """``rescore_keys`` on a FRESH d_model = 1024 model (no step decoder attached yet): the entry point runs under ``torch.inference_mode``,
    builds ``BartStepDecoder`` there, so the decoder's concatenated q/k/v and cross k/v weights -- and the encoder's -- are inference tensors,
    and at this width every product takes the split path, which looks its weights up in the version-validated cache.  The scores must
    equal one HF row per key (the reference's batching, keys.py:64-141)."""
    import numpy as np
    from transformers import BartConfig, BartForConditionalGeneration
    from seal_amd.keys import rescore_keys
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = BartConfig(vocab_size=3000, d_model=1024, encoder_layers=1, decoder_layers=2, encoder_attention_heads=16, decoder_attention_heads=16,
                     encoder_ffn_dim=4096, decoder_ffn_dim=4096, max_position_embeddings=64)
    with torch.device(dev):
        model = BartForConditionalGeneration(cfg).eval()
    assert not hasattr(model, "_seal_step_decoder")
    rng = np.random.default_rng(3)
    B = 8
    inputs = [[0] + rng.integers(4, 2990, size=int(n)).tolist() + [2] for n in rng.integers(20, 30, size=B)]
    keys = []
    for _ in range(B):                                    # ~190 distinct prefixes per query: > 128 rows, every product over the MAC threshold
        kk = []
        for _ in range(24):
            base = rng.integers(4, 2990, size=8).tolist()
            kk += [base[:i] for i in range(1, 9)]
        keys.append([(-1.0, k) for k in kk])
    a = rescore_keys(model, inputs, keys, batch_size=B, share_prefixes=True)
    dec = model._seal_step_decoder
    assert dec.layers[0]["qkv_w"].is_inference()          # the situation the test is about
    assert len(dec.split_gemm._by_weight) > 0, "no product took the split path: the test does not reach the cache"
    b = rescore_keys(model, inputs, keys, batch_size=B, share_prefixes=False)
    for qa, qb in zip(a, b):
        assert [k for _, k in qa] == [k for _, k in qb]
        for (sa, _), (sb, _) in zip(qa, qb):
            assert abs(sa - sb) <= 1e-4 * max(1.0, abs(sb))
