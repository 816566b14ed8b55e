"""A/B of the P16 GEMM's prologue/epilogue options on the decoder's shapes; run under rocprofv3 --kernel-trace and read
the gemm_p16_kernel durations in launch order (tools/p16_ablate.sh)."""
import importlib, sys, torch
sys.path.insert(0, ".")
hip = importlib.import_module("matcha-tts-24k_amd._hip")
torch.manual_seed(0)
M, C = 20480, 384
a = torch.randn(M, C, device="cuda")
part = torch.stack([a.view(M, 6, 64).mean(-1), ((a.view(M, 6, 64) - a.view(M, 6, 64).mean(-1, keepdim=True)) ** 2).sum(-1)], -1).contiguous()
for N in (1152, 384, 1536):
    w = torch.randn(N, C, device="cuda") * C ** -0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    p0 = torch.rand(N, device="cuda") + 0.5
    cases = [
        ("plain f32 out", dict()),
        ("bias", dict(bias=b)),
        ("LN", dict(a_part=part)),
        ("p16 out only", dict(want_f32=False, want_p16=True)),
        ("LN + p16 only", dict(a_part=part, want_f32=False, want_p16=True)),
        ("res + f32 + p16 + stats", dict(res=r, want_p16=True, stats_out=True)),
        ("LN + snake + p16 only", dict(a_part=part, act=3, p0=p0, p1=p0, want_f32=False, want_p16=True)),
    ]
    for name, kw in cases:
        bias = kw.pop("bias", None)
        for _ in range(3):
            hip.gemm_p16(a, w, bias, B=32, T_in=640, **kw)
        print(f"CASE N={N} {name}")
torch.cuda.synchronize()
