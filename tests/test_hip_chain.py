"""GPU parity of the transformer-block chain kernel (csrc/tblock_chain.hip) through the C ABI (mtts_tblock_chain) against fp64
PyTorch on the CPU: out-projection + residual, LayerNorm, FeedForward with SnakeBeta + residual, LayerNorm and the following
block's q|k|v projection in ONE launch (reference matcha/models/components/transformer.py:249-301,104-120,61-77)."""
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


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def ln0(v):
    mu = v.mean(1, keepdim=True)
    var = ((v - mu) ** 2).mean(1, keepdim=True)
    return (v - mu) / torch.sqrt(var + 1e-5)


def chain_ref(att, x, w_out, b_out, w1, b1, p0, p1, w2, b2, w_qkv, b_qkv):
    d = lambda t: None if t is None else t.double()
    x1 = d(x) if att is None else d(x) + F.linear(d(att), d(w_out), d(b_out))
    h = F.linear(ln0(x1), d(w1), d(b1))
    h = h + d(p1) * torch.sin(h * d(p0)) ** 2
    x2 = x1 + F.linear(h, d(w2), d(b2))
    qkv = None if w_qkv is None else F.linear(ln0(x2), d(w_qkv), d(b_qkv))
    return x1, x2, qkv


def make_case(M, C, inner, n_qkv, seed):
    att = rnd(M, inner, seed=seed + 1) if inner else None
    x = rnd(M, C, seed=seed + 2) * 2 + 0.3
    w_out = rnd(C, inner, seed=seed + 3, scale=inner ** -0.5) if inner else None
    b_out = rnd(C, seed=seed + 4) if inner else None
    w1, b1 = rnd(4 * C, C, seed=seed + 5, scale=C ** -0.5), rnd(4 * C, seed=seed + 6)
    p0 = torch.exp(rnd(4 * C, seed=seed + 7, scale=0.2))
    p1 = 1.0 / (torch.exp(rnd(4 * C, seed=seed + 8, scale=0.2)) + 1e-9)
    w2, b2 = rnd(C, 4 * C, seed=seed + 9, scale=(4 * C) ** -0.5), rnd(C, seed=seed + 10)
    w_qkv = rnd(n_qkv, C, seed=seed + 11, scale=C ** -0.5) if n_qkv else None
    b_qkv = rnd(n_qkv, seed=seed + 12) if n_qkv else None
    return att, x, w_out, b_out, w1, b1, p0, p1, w2, b2, w_qkv, b_qkv


def close(out, ref, tol, what):
    err = (out.detach().cpu().double() - ref).abs().max().item()
    mag = max(ref.abs().max().item(), 1.0)
    assert err <= tol * mag, f"{what}: max abs err {err:.3e} (ref magnitude {mag:.3e})"


CASES = [
    # M, C, inner, n_qkv, qb, ch
    (200, 384, 384, 1152, 64, 128),       # production width, a partly filled last workgroup
    (161, 384, 384, 1152, 32, 128),
    (130, 384, 384, 1152, 48, 256),       # 256-wide hidden chunks, 48-row workgroups
    (97, 384, 384, 1152, 32, 256),
    (64, 384, 384, 0, 64, 128),           # last block of a run: no q|k|v
    (70, 384, 0, 0, 64, 128),             # FeedForward alone
    (75, 128, 128, 384, 64, 128),         # narrow estimators of the test suite
    (33, 128, 128, 384, 32, 128),
    (90, 256, 192, 576, 64, 128),         # q|k|v width that does not fill the last pass (36 tiles of 48)
    (50, 256, 128, 384, 32, 128),
]


@pytest.mark.parametrize("M,C,inner,n_qkv,qb,ch", CASES)
def test_chain_vs_fp64(hip, M, C, inner, n_qkv, qb, ch):
    case = make_case(M, C, inner, n_qkv, seed=100 + M)
    att, x = case[0], case[1]
    x1, x2, qkv = chain_ref(*case)
    dev = torch.device("cuda")
    x_out, qkv_out = hip.tblock_chain(None if att is None else att.to(dev), x.to(dev), *case[2:10], w_qkv=case[10], b_qkv=case[11],
                                      qb=qb, ch=ch)
    # 22 significand bits per operand, fp32 accumulation, three GEMMs deep: the same bar as the single P16 GEMM tests
    close(x_out, x2, 2e-5, "x_out")
    if n_qkv:
        close(qkv_out, qkv, 3e-5, "qkv")


def test_chain_masked_rows_and_exact_row_independence(hip):
    """x_out_mask zeroes whole rows of the output image (the masked copy the convs read) without touching the q|k|v of those
    rows, and a row's result does not depend on which workgroup / row tile it lands in (bitwise)."""
    M, C = 150, 384
    case = make_case(M, C, 384, 1152, seed=7)
    dev = torch.device("cuda")
    mask = (torch.arange(M) % 5 != 0).float()
    args = [None if t is None else t for t in case]
    a, x = args[0].to(dev), args[1].to(dev)
    plain, q_plain = hip.tblock_chain(a, x, *args[2:10], w_qkv=args[10], b_qkv=args[11], qb=64, ch=128)
    masked, q_masked = hip.tblock_chain(a, x, *args[2:10], w_qkv=args[10], b_qkv=args[11], out_mask=mask.to(dev), qb=64, ch=128)
    keep = mask.bool()
    assert torch.equal(masked[keep.to(dev)], plain[keep.to(dev)])
    assert masked[(~keep).to(dev)].abs().max().item() == 0.0
    assert torch.equal(q_masked, q_plain)
    # the same rows shifted by 19 positions: other row tiles, other lanes
    shift = 19
    a2, x2 = torch.roll(a, shift, 0), torch.roll(x, shift, 0)
    rolled, q_rolled = hip.tblock_chain(a2, x2, *args[2:10], w_qkv=args[10], b_qkv=args[11], qb=64, ch=128)
    assert torch.equal(torch.roll(rolled, -shift, 0), plain)
    assert torch.equal(torch.roll(q_rolled, -shift, 0), q_plain)
    # and with 32-row workgroups
    small, q_small = hip.tblock_chain(a, x, *args[2:10], w_qkv=args[10], b_qkv=args[11], qb=32, ch=128)
    assert torch.equal(small, plain) and torch.equal(q_small, q_plain)


def test_prefetch_workgroups_do_not_change_results(hip):
    """The chain launch's prefetch workgroups (MTTS_CHAIN_PF, read once per process; default 8) only touch the weight stream:
    0, 8 and 24 of them give bitwise the same outputs (one subprocess per setting), a partly filled last workgroup included."""
    import hashlib, os, subprocess, sys
    from conftest import ROOT
    code = (
        "import sys, hashlib, importlib, torch\n"
        f"sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})\n"
        "hip = importlib.import_module('matcha-tts-24k_amd._hip')\n"
        "from test_hip_chain import make_case\n"
        "case = make_case(700, 384, 384, 1152, seed=3)\n"
        "dev = torch.device('cuda')\n"
        "xo, q = hip.tblock_chain(case[0].to(dev), case[1].to(dev), *case[2:10], w_qkv=case[10], b_qkv=case[11], qb=48, ch=256)\n"
        "print('HASH', hashlib.sha256(xo.cpu().numpy().tobytes() + q.cpu().numpy().tobytes()).hexdigest())\n")
    hashes = {}
    for pf in ("0", "8", "24"):
        env = dict(os.environ, MTTS_CHAIN_PF=pf)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        hashes[pf] = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("HASH")][0]
    assert len(set(hashes.values())) == 1, hashes


PAIR_CASES = [
    # M, C, inner, n_qkv, qb, ch
    (500, 384, 384, 1152, 48, 256),       # 11 row tiles -> 32 workgroups (two groups of 16, the second partly idle)
    (130, 384, 384, 1152, 32, 256),
    (300, 384, 384, 0, 48, 256),          # last block of a run: no q|k|v
    (90, 256, 192, 576, 64, 128),         # 8 hidden chunks, q|k|v width that does not fill its last pass
    (70, 128, 128, 384, 32, 128),
]


@pytest.mark.parametrize("M,C,inner,n_qkv,qb,ch", PAIR_CASES)
def test_chain_pair_form_vs_fp64(hip, M, C, inner, n_qkv, qb, ch):
    """The pair form (two workgroups of one XCD per row tile, FF2 partial sums exchanged through the L2): same bar as the
    single-workgroup form, bitwise repeatable, and equal to that form up to the order of one addition per element."""
    case = make_case(M, C, inner, n_qkv, seed=300 + M)
    x1, x2, qkv = chain_ref(*case)
    dev = torch.device("cuda")
    args = dict(w_qkv=case[10], b_qkv=case[11], qb=qb, ch=ch)
    x_out, qkv_out = hip.tblock_chain(case[0].to(dev), case[1].to(dev), *case[2:10], pair=True, **args)
    close(x_out, x2, 2e-5, "x_out")
    if n_qkv:
        close(qkv_out, qkv, 3e-5, "qkv")
    again, q_again = hip.tblock_chain(case[0].to(dev), case[1].to(dev), *case[2:10], pair=True, **args)
    assert torch.equal(again, x_out) and (not n_qkv or torch.equal(q_again, qkv_out))
    single, q_single = hip.tblock_chain(case[0].to(dev), case[1].to(dev), *case[2:10], **args)
    assert (single - x_out).abs().max().item() <= 4e-6 * max(1.0, x2.abs().max().item())


def test_chain_pair_form_masked_rows(hip):
    M, C = 200, 384
    case = make_case(M, C, 384, 0, seed=11)
    dev = torch.device("cuda")
    mask = (torch.arange(M) % 3 != 0).float()
    plain, _ = hip.tblock_chain(case[0].to(dev), case[1].to(dev), *case[2:10], qb=48, ch=256, pair=True)
    masked, _ = hip.tblock_chain(case[0].to(dev), case[1].to(dev), *case[2:10], out_mask=mask.to(dev), qb=48, ch=256, pair=True)
    keep = mask.bool().to(dev)
    assert torch.equal(masked[keep], plain[keep]) and masked[~keep].abs().max().item() == 0.0
