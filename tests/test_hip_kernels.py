"""GPU parity of the single HIP kernels, called through the C ABI (include/mtts.h), against fp64 PyTorch on the CPU.

fp32-input MFMA is an exact fp32 FMA chain, so the only difference from a CPU fp32 GEMM is summation order;
tolerances are a few fp32 ulps of the accumulated magnitude."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import sub

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    return sub("_hip")


@pytest.fixture(params=[0, 6, 2], ids=["fp32-mfma", "bf16x6", "f16x3s"])
def terms(request):
    """GEMM arithmetic: native fp32 MFMA, the three-term bf16 split, or the two-term fp16 split with scaled residual
    (all fp32-equivalent; see gemm_f32.hip)."""
    return request.param


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(out, ref, tol):
    out = out.detach().cpu().double()
    err = (out - ref).abs().max().item()
    mag = ref.abs().max().item()
    assert err <= tol * max(mag, 1.0), f"max abs err {err:.3e} (ref magnitude {mag:.3e})"


def conv_ref(a, w, bias, B, T, pad, stride=1):
    """a [B*T, C] channels-last -> conv1d -> [B*T_out, N] (fp64)."""
    C = a.shape[1]
    x = a.double().view(B, T, C).transpose(1, 2)
    y = F.conv1d(x, w.double(), None if bias is None else bias.double(), stride=stride, padding=pad)
    return y.transpose(1, 2).reshape(-1, w.shape[0])


@pytest.mark.parametrize("B,T,C,N", [(3, 100, 384, 384), (2, 77, 200, 100), (1, 5, 64, 1), (4, 160, 1536, 384), (1, 130, 96, 288)])
def test_linear_bias_residual(hip, terms, B, T, C, N):
    a, w, b, r = rnd(B * T, C, seed=1), rnd(N, C, seed=2, scale=C ** -0.5), rnd(N, seed=3), rnd(B * T, N, seed=4)
    ref = F.linear(a.double(), w.double(), b.double()) + r.double()
    out = hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, res=r.cuda(), terms=terms)
    close(out, ref, 2e-6 * math.sqrt(C))


def test_large_grid_uses_128_row_tiles(hip, terms):
    """M x N big enough for the 128x128 block tile (small problems above run on the 64x128 variant)."""
    B, T, C, N = 8, 1000, 96, 1152
    a, w, b = rnd(B * T, C, seed=41), rnd(N, C, seed=42, scale=C ** -0.5), rnd(N, seed=43)
    ref = F.linear(a.double(), w.double(), b.double())
    out = hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, terms=terms)
    close(out, ref, 2e-5)
    # implicit conv on the same tile shape, ragged mask
    w3 = rnd(N, C, 3, seed=44, scale=(3 * C) ** -0.5)
    lens = torch.tensor([T - 13 * i for i in range(B)])
    mask = (torch.arange(T)[None] < lens[:, None]).float().reshape(-1)
    out = hip.gemm_f32(a.cuda(), w3.cuda(), b.cuda(), B=B, T_in=T, a_mask=mask.cuda(), terms=terms)
    close(out, conv_ref(a * mask[:, None], w3, b, B, T, 1), 2e-5)


def test_identity_asymmetric(hip, terms):
    """A = I with an asymmetric W catches a transposed accumulator map."""
    n = 128
    a = torch.eye(n)
    w = torch.arange(n * n, dtype=torch.float32).view(n, n) / 1000.0
    out = hip.gemm_f32(a.cuda(), w.cuda(), None, B=1, T_in=n, terms=terms)
    if terms == 2:      # 22 significand bits per operand: not bit-exact, but far below any transposition error
        assert (out.cpu() - w.t()).abs().max().item() < 1e-5
    else:
        assert torch.equal(out.cpu(), w.t().contiguous())


@pytest.mark.parametrize("k,C,N,B,T", [(3, 64, 96, 2, 77), (5, 96, 160, 3, 50), (3, 200, 384, 2, 130), (5, 1152, 288, 1, 128)])
def test_conv_same_with_mask(hip, terms, k, C, N, B, T):
    a, w, b = rnd(B * T, C, seed=5), rnd(N, C, k, seed=6, scale=(C * k) ** -0.5), rnd(N, seed=7)
    lens = torch.tensor([T - 7 * i for i in range(B)])
    mask = (torch.arange(T)[None] < lens[:, None]).float().reshape(-1)
    ref = conv_ref(a * mask[:, None], w, b, B, T, k // 2)
    out = hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, a_mask=mask.cuda(), terms=terms)
    close(out, ref, 2e-6 * math.sqrt(C * k))
    # the same with relu / silu and an output mask
    for act, fn in ((1, torch.relu), (2, F.silu)):
        out = hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, a_mask=mask.cuda(), act=act, out_mask=mask.cuda(), terms=terms)
        close(out, fn(ref) * mask[:, None].double(), 2e-6 * math.sqrt(C * k))


def test_conv_stride2(hip, terms):
    B, T, C, N = 2, 50, 64, 64
    a, w, b = rnd(B * T, C, seed=8), rnd(N, C, 3, seed=9, scale=(3 * C) ** -0.5), rnd(N, seed=10)
    ref = conv_ref(a, w, b, B, T, 1, stride=2)
    out = hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, T_out=25, in_stride=2, terms=terms)
    close(out, ref, 2e-5)


def test_layernorm_prologue_and_snake(hip, terms):
    B, T, C, N = 2, 90, 384, 1536
    a = rnd(B * T, C, seed=11) * 2 + 0.3
    w, b = rnd(N, C, seed=12, scale=C ** -0.5), rnd(N, seed=13)
    alpha, beta = rnd(N, seed=14, scale=0.2), rnd(N, seed=15, scale=0.2)
    mean, rstd = hip.row_stats(a.cuda())
    ad = a.double()
    mu = ad.mean(1)
    var = ((ad - mu[:, None]) ** 2).mean(1)
    close(mean, mu, 1e-6)
    close(rstd, 1 / torch.sqrt(var + 1e-5), 1e-6)
    h = F.linear((ad - mu[:, None]) / torch.sqrt(var + 1e-5)[:, None], w.double(), b.double())
    ae, ib = torch.exp(alpha), 1.0 / (torch.exp(beta) + 1e-9)
    ref = h + ib.double() * torch.sin(h * ae.double()) ** 2
    out = hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, a_mean=mean, a_rstd=rstd, act=3, p0=ae.cuda(), p1=ib.cuda(), terms=terms)
    close(out, ref, 1e-5)


def test_layernorm_stats_travel_through_epilogue(hip, terms):
    """GEMM 1 writes x = a.W1^T + res and leaves (mean, M2) partials of its 64-column slices; GEMM 2 merges them in its
    prologue as the LayerNorm of x.  Must equal the two-pass statistics of row_stats on x."""
    for B, T in ((2, 100), (8, 1000)):      # 64-row and 128-row block tiles
        C = 384
        a, r = rnd(B * T, C, seed=51), rnd(B * T, C, seed=52) * 2 + 0.5
        w1, b1 = rnd(C, C, seed=53, scale=C ** -0.5), rnd(C, seed=54)
        w2, b2 = rnd(1152, C, seed=55, scale=C ** -0.5), rnd(1152, seed=56)
        x, stats = hip.gemm_f32(a.cuda(), w1.cuda(), b1.cuda(), B=B, T_in=T, res=r.cuda(), stats_out=True, terms=terms)
        xd = x.cpu().double()
        assert torch.allclose(stats[:, :, 0].cpu().double(), xd.view(B * T, 6, 64).mean(-1), atol=1e-5)
        mean, rstd = hip.row_stats(x)
        y_ref = hip.gemm_f32(x, w2.cuda(), b2.cuda(), B=B, T_in=T, a_mean=mean, a_rstd=rstd, terms=terms)
        y = hip.gemm_f32(x, w2.cuda(), b2.cuda(), B=B, T_in=T, a_part=stats, terms=terms)
        mu = xd.mean(1, keepdim=True)
        var = ((xd - mu) ** 2).mean(1, keepdim=True)
        ref = F.linear((xd - mu) / torch.sqrt(var + 1e-5), w2.double(), b2.double())
        close(y, ref, 1e-5)
        assert (y - y_ref).abs().max().item() < 2e-5


def test_two_term_split_is_opt_in_and_looser(hip):
    """terms=3 (two bf16 terms, three products): ~2^-17 per product -- documented, not the default."""
    B, T, C, N = 2, 300, 384, 384
    a, w, b = rnd(B * T, C, seed=61), rnd(N, C, seed=62, scale=C ** -0.5), rnd(N, seed=63)
    ref = F.linear(a.double(), w.double(), b.double())
    e3 = (hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, terms=3).cpu().double() - ref).abs().max().item()
    e6 = (hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, terms=6).cpu().double() - ref).abs().max().item()
    e0 = (hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, terms=0).cpu().double() - ref).abs().max().item()
    e2 = (hip.gemm_f32(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, terms=2).cpu().double() - ref).abs().max().item()
    assert e6 < 3 * e0 + 1e-6      # fp32-equivalent
    assert e2 < 3 * e0 + 1e-6      # fp32-equivalent
    # fp16 split: tiny and huge operands (subnormal residuals are avoided by the 2^11 scaling; beyond 65504 saturates, no inf)
    small = hip.gemm_f32((a * 1e-4).cuda(), (w * 1e-3).cuda(), None, B=B, T_in=T, terms=2).cpu().double()
    ref_s = F.linear((a * 1e-4).double(), (w * 1e-3).double())
    assert (small - ref_s).abs().max().item() < 1e-5 * ref_s.abs().max().item()
    big = a.clone(); big[0, 0] = 1e6
    assert torch.isfinite(hip.gemm_f32(big.cuda(), w.cuda(), None, B=B, T_in=T, terms=2)).all()
    assert e3 < 2e-4 and e3 > e6   # visibly looser, still small


# ------------------------------------------------------------------------------------------------ P16-operand GEMM
@pytest.mark.parametrize("B,T,C,N,bm", [(3, 100, 384, 384, 0), (2, 77, 224, 100, 0), (8, 1000, 96, 1152, 128), (4, 160, 1536, 384, 64),
                                        (1, 5, 64, 4, 0)])
def test_p16_linear_bias_residual(hip, B, T, C, N, bm):
    """Operands pre-split into fp16 head/residual images, tiles by LDS-DMA (csrc/gemm_p16.hip): same fp32-equivalent result."""
    a, w, b, r = rnd(B * T, C, seed=1), rnd(N, C, seed=2, scale=C ** -0.5), rnd(N, seed=3), rnd(B * T, N, seed=4)
    ref = F.linear(a.double(), w.double(), b.double()) + r.double()
    o = hip.gemm_p16(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, res=r.cuda(), force_bm=bm)
    close(o["out"], ref, 2e-6 * math.sqrt(C))


@pytest.mark.parametrize("k,C,N,B,T", [(3, 64, 96, 2, 77), (5, 96, 160, 3, 50), (3, 224, 384, 2, 130), (5, 1152, 288, 1, 128)])
def test_p16_conv_same_with_mask(hip, k, C, N, B, T):
    """Implicit conv: the tap shift and the zero padding at the sequence ends are per-lane DMA source addresses; the row
    mask is folded into the P16 image by its producer (here the conversion pass)."""
    a, w, b = rnd(B * T, C, seed=5), rnd(N, C, k, seed=6, scale=(k * C) ** -0.5), rnd(N, seed=7)
    lens = [T - 7 * i for i in range(B)]
    mask = torch.zeros(B, T)
    for i, n in enumerate(lens):
        mask[i, :n] = 1
    mask = mask.view(-1)
    ref = conv_ref(a * mask[:, None], w, b, B, T, k // 2) * mask[:, None].double()
    o = hip.gemm_p16(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, a_mask=mask.cuda(), out_mask=mask.cuda())
    close(o["out"], ref, 2e-5)


def test_p16_conv_stride2(hip):
    B, T, C, N = 2, 50, 64, 64
    a, w, b = rnd(B * T, C, seed=8), rnd(N, C, 3, seed=9, scale=(3 * C) ** -0.5), rnd(N, seed=10)
    ref = conv_ref(a, w, b, B, T, 1, stride=2)
    o = hip.gemm_p16(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, T_out=25, in_stride=2)
    close(o["out"], ref, 2e-5)


@pytest.mark.parametrize("B,T", [(2, 90), (8, 1000)])
def test_p16_layernorm_in_epilogue_snake_and_p16_output(hip, B, T):
    """LayerNorm applied after the product as rstd * (x.W - mean * rowsum(W)) from the producer's partial moments, SnakeBeta,
    and the result written as a P16 image for the next GEMM (decoded here): all against fp64."""
    C, N = 384, 1536
    a = rnd(B * T, C, seed=11) * 2 + 0.3
    w, b = rnd(N, C, seed=12, scale=C ** -0.5), rnd(N, seed=13)
    alpha, beta = rnd(N, seed=14, scale=0.2), rnd(N, seed=15, scale=0.2)
    ad = a.double()
    part = torch.stack([ad.view(-1, 6, 64).mean(-1), ((ad.view(-1, 6, 64) - ad.view(-1, 6, 64).mean(-1, keepdim=True)) ** 2).sum(-1)], -1)
    mu = ad.mean(1)
    var = ((ad - mu[:, None]) ** 2).mean(1)
    h = F.linear((ad - mu[:, None]) / torch.sqrt(var + 1e-5)[:, None], w.double(), b.double())
    ae, ib = torch.exp(alpha), 1.0 / (torch.exp(beta) + 1e-9)
    ref = h + ib.double() * torch.sin(h * ae.double()) ** 2
    o = hip.gemm_p16(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, a_part=part.float().cuda(), act=3, p0=ae.cuda(), p1=ib.cuda(),
                     want_p16=True)
    close(o["out"], ref, 1e-5)
    close(o["out16"], ref, 1e-5)                       # 22 significand bits survive the image
    mean, rstd = hip.row_stats(a.cuda())
    o2 = hip.gemm_p16(a.cuda(), w.cuda(), b.cuda(), B=B, T_in=T, a_mean=mean, a_rstd=rstd, act=3, p0=ae.cuda(), p1=ib.cuda())
    close(o2["out"], ref, 1e-5)


def test_p16_stats_out_and_unscaled_residual(hip):
    """The epilogue's 64-column partial moments, and the unscaled residual image the attention kernel reads (lscale = 1)."""
    B, T, C = 2, 100, 384
    a, r = rnd(B * T, C, seed=51), rnd(B * T, C, seed=52) * 2 + 0.5
    w1, b1 = rnd(C, C, seed=53, scale=C ** -0.5), rnd(C, seed=54)
    o = hip.gemm_p16(a.cuda(), w1.cuda(), b1.cuda(), B=B, T_in=T, res=r.cuda(), stats_out=True, want_p16=True, lscale=1.0)
    ref = F.linear(a.double(), w1.double(), b1.double()) + r.double()
    close(o["out"], ref, 1e-5)
    close(o["out16"], ref, 1e-5)
    xd = o["out"].cpu().double().view(B * T, 6, 64)
    assert torch.allclose(o["stats"][:, :, 0].cpu().double(), xd.mean(-1), atol=1e-5)
    assert torch.allclose(o["stats"][:, :, 1].cpu().double(), ((xd - xd.mean(-1, keepdim=True)) ** 2).sum(-1), rtol=1e-4, atol=1e-4)



@pytest.mark.parametrize("B,T,H,D,mode", [(2, 320, 6, 64, 0), (3, 130, 6, 48, 1), (2, 24, 2, 32, 0), (1, 77, 2, 24, 1), (1, 640, 2, 64, 0),
                                           (8, 1024, 12, 64, 0), (8, 1000, 12, 48, 1)])   # the last two run the 128-query block variant
def test_attention(hip, oracle, B, T, H, D, mode):
    qkv = rnd(B * T, 3 * H * D, seed=20)
    lens = torch.tensor([T - 11 * i for i in range(B)])
    mask = (torch.arange(T)[None] < lens[:, None]).float()
    q, k, v = [z.view(B, T, H, D).transpose(1, 2).double() for z in qkv.split(H * D, dim=1)]
    scale = 1.0 / math.sqrt(D)
    if mode == 0:
        ref = oracle.sdpa_reference(q, k, v, mask.double().view(B, 1, 1, T).expand(B, H, 1, T), scale)
    else:
        ref = oracle.sdpa_reference(q, k, v, (mask[:, None, :, None] * mask[:, None, None, :]).bool(), scale)
    ref = ref.transpose(1, 2).reshape(B * T, H * D)
    out = hip.attention_f32(qkv.cuda(), mask.reshape(-1).cuda(), B, T, H, D, scale, mode)
    if mode == 1:   # padded query rows are "don't care" (zeroed downstream by x_mask)
        keep = mask.reshape(-1).bool()
        out, ref = out.cpu()[keep], ref[keep]
    close(out, ref, 5e-6)


@pytest.mark.parametrize("B,T,H,mode", [(2, 320, 6, 0), (1, 640, 2, 0), (3, 130, 2, 1), (32, 640, 6, 0), (4, 161, 6, 0), (2, 192, 3, 0), (3, 65, 2, 0)])
def test_attention_p16_io(hip, oracle, B, T, H, mode):
    """Same kernel with q|k|v read from a P16 image and the output written as one (the decoder's transformer blocks).  T in
    65..192 runs the whole-sequence form (one workgroup per utterance and head, every key staged once): the half-length level."""
    D = 64
    qkv = rnd(B * T, 3 * H * D, seed=22)
    lens = torch.tensor([T - 3 * i for i in range(B)])
    mask = (torch.arange(T)[None] < lens[:, None]).float()
    out = hip.attention_p16(qkv.cuda(), mask.reshape(-1).cuda(), B, T, H, D, 0.125, mode)
    ref = hip.attention_f32(qkv.cuda(), mask.reshape(-1).cuda(), B, T, H, D, 0.125, mode)      # itself checked against fp64 above
    if mode == 1:
        keep = mask.reshape(-1).bool().cuda()
        out, ref = out[keep], ref[keep]
    assert (out - ref).abs().max().item() < 2e-6
    if B <= 3:
        q, k, v = [z.view(B, T, H, D).transpose(1, 2).double() for z in qkv.split(H * D, dim=1)]
        if mode == 0:
            r64 = oracle.sdpa_reference(q, k, v, mask.double().view(B, 1, 1, T).expand(B, H, 1, T), 0.125)
        else:
            r64 = oracle.sdpa_reference(q, k, v, (mask[:, None, :, None] * mask[:, None, None, :]).bool(), 0.125)
        r64 = r64.transpose(1, 2).reshape(B * T, H * D)
        o = out.cpu()
        if mode == 1:
            r64 = r64[mask.reshape(-1).bool()]
        close(o, r64, 5e-6)


def test_attention_forced_rescale(hip, oracle):
    """One key per tile dominates so that the running max jumps at every tile boundary (online-softmax rescale path)."""
    B, T, H, D = 1, 256, 1, 64
    qkv = rnd(B * T, 3 * D, seed=21, scale=0.5)
    q, k, v = qkv[:, :D].clone(), qkv[:, D:2 * D].clone(), qkv[:, 2 * D:].clone()
    for tile in range(4):
        k[tile * 64 + 13] = q[5] * (4.0 + 3.0 * tile)
    qkv = torch.cat([q, k, v], 1)
    mask = torch.ones(B * T)
    ref = oracle.sdpa_reference(q.double()[None, None], k.double()[None, None], v.double()[None, None], None, 0.125)[0, 0]
    out = hip.attention_f32(qkv.cuda(), mask.cuda(), B, T, H, D, 0.125, 0)
    close(out, ref, 5e-6)


@pytest.mark.parametrize("B,T,C", [(2, 100, 384), (3, 24, 64), (1, 33, 384)])
def test_groupnorm_mish(hip, B, T, C):
    y = rnd(B * T, C, seed=30) * 1.5 + 0.2
    g, b = 1 + 0.1 * rnd(C, seed=31), 0.1 * rnd(C, seed=32)
    lens = torch.tensor([T - 5 * i for i in range(B)])
    mask = (torch.arange(T)[None] < lens[:, None]).float().reshape(-1)
    x = y.double().view(B, T, C).transpose(1, 2)
    ref = (F.mish(F.group_norm(x, 8, g.double(), b.double(), eps=1e-5)) * mask.view(B, 1, T).double()).transpose(1, 2).reshape(-1, C)
    out = hip.groupnorm_mish(y.cuda(), g.cuda(), b.cuda(), mask.cuda(), B, T)
    close(out, ref, 3e-6)


@pytest.mark.parametrize("B,T,C,act,film,mask", [(3, 37, 96, 0, True, True), (2, 128, 256, 0, True, False), (2, 50, 192, 2, False, True),
                                                   (1, 1, 96, 0, True, True)])
def test_channel_layernorm_film(hip, B, T, C, act, film, mask):
    """layernorm_kernel: channel LayerNorm (biased variance, reference text_encoder.py:19-27) + SiLU (ConvSiluNorm) or the
    DurationPredictor's per-utterance FiLM `x * gamma_b + beta_b` (text_encoder.py:102-109) + row mask, against fp64."""
    x = rnd(B * T, C, seed=1, scale=3.0) + 0.7
    g, b = 1.0 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    fm = (rnd(B, 2 * C, seed=4) + 0.5) if film else None
    mk = (rnd(B * T, seed=5) > -0.3).float() if mask else None
    xd = x.double()
    mean = xd.mean(1, keepdim=True)
    var = ((xd - mean) ** 2).mean(1, keepdim=True)
    ref = (xd - mean) * torch.rsqrt(var + 1e-5) * g.double() + b.double()
    if act == 2:
        ref = F.silu(ref)
    if film:
        fb = fm.double().repeat_interleave(T, dim=0)
        ref = ref * fb[:, :C] + fb[:, C:]
    if mask:
        ref = ref * mk.double()[:, None]
    out = hip.channel_layernorm(x.cuda(), g.cuda(), b.cuda(), B, T, act=act, film=None if fm is None else fm.cuda(),
                                mask=None if mk is None else mk.cuda())
    close(out, ref, 3e-6)
