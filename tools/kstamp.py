#!/usr/bin/env python3
"""In-kernel stamps of the P16 GEMM (diagnostic build, -DMTTS_KSTAMP): where a workgroup's lifetime goes.
  build (CPU box):  python tools/kstamp.py --build          -> tools/ab/kstamp.so (gemm_p16.hip recompiled with the stamps)
  run (GPU box):    MTTS_HIP_LIB=$PWD/tools/ab/kstamp.so python tools/kstamp.py
Per shape: median cycles from kernel start to the first tile landed, per k-step of the main loop, of the epilogue; the
workgroup lifetime and the launch span in microseconds; the in-kernel clock (s_memtime / s_memrealtime)."""
import argparse
import ctypes as C
import importlib
import os
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"


def build(extra=(), name="kstamp", nostamp=False):
    pk = ROOT / PKG
    obj = f"/tmp/gemm_p16_{name}.o"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
           "-DMTTS_KSTAMP", *extra, "-c", str(pk / "csrc" / "gemm_p16.hip"), "-o", obj]
    if nostamp:                                                # an A/B library without the stamps (bench.py under MTTS_HIP_LIB)
        cmd.remove("-DMTTS_KSTAMP")
    subprocess.run(cmd, check=True)
    others = [str(pk / "build" / f"{n}.o") for n in ("gemm_f32", "attention_f32", "norm_glue", "vocos", "model")]
    out = ROOT / "tools" / "ab" / f"{name}.so"
    out.parent.mkdir(exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", str(out)], check=True)
    print("built", out)


def fmt_stamps(s, nk):
    """s: int64 tensor [WGs, 8] of one launch -> the table columns."""
    med = statistics.median
    first = (s[:, 1] - s[:, 0]).tolist()
    loop = ((s[:, 2] - s[:, 1]).double() / max(nk - 1, 1)).tolist()           # stamp 1 sits after the first tile's barrier
    epi = (s[:, 3] - s[:, 2]).tolist()
    life = (s[:, 3] - s[:, 0]).double()
    rt = (s[:, 5] - s[:, 4]).double().clamp_min(1)                            # 100 MHz ticks
    ghz = float((life / rt).median()) * 0.1
    span_us = float(s[:, 5].max() - s[:, 4].min()) / 100.0
    return (f"{len(s):5d} {nk:3d} | {med(first):10.0f} {med(loop):10.0f} {med(epi):9.0f} | "
            f"{float(life.median()) / (ghz * 1e3):6.1f} {span_us:7.1f} {ghz:5.2f} | park {med((s[:, 6] - s[:, 2]).tolist()):5.0f} "
            f"chunk0 {med((s[:, 7] - s[:, 6]).tolist()):6.0f} rest {med((s[:, 3] - s[:, 7]).tolist()):6.0f}"
            f" | s7-s0 {med((s[:, 7] - s[:, 0]).tolist()):6.0f} (setup, -DMTTS_KSTAMP_SETUP builds only)")


def insitu(args, lib, torch, dev):
    """Stamps of launches INSIDE a synthesise() step: the n-th P16 GEMM launch of the step, n = --first .. --first + --count."""
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    lib.mtts_debug_set_kstamp_nth.argtypes = [C.c_void_p, C.c_int]
    lib.mtts_debug_set_kstamp_nth.restype = None
    lib.mtts_debug_kstamp_info.argtypes = [C.POINTER(C.c_int)]
    lib.mtts_debug_kstamp_info.restype = None
    hp = hparams.prod_v20(n_spks=1)
    model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(synthetic.make_state_dict(hp, seed=7), strict=True)
    model = model.to(dev).eval()
    x, x_len, _ = synthetic.make_inputs(hp, args.batch, 128, seed=1234)
    x, x_len = x.to(dev), x_len.to(dev)
    for _ in range(2):
        model.synthesise(x, x_len, 10, speaker=0)
    stamps = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
    info = (C.c_int * 8)()
    print(f"{'n':>4s} {'M':>6s} {'N':>5s} {'K':>5s} {'BM':>3s} st ks flags | {'WGs':>5s} {'nk':>3s} | {'first tile':>10s} {'cyc/k-step':>10s} {'epilogue':>9s} | {'WG us':>6s} {'span us':>7s} {'GHz':>5s}")
    seen = {}
    for n in range(args.first, args.first + args.count):
        stamps.zero_()
        torch.cuda.synchronize()
        lib.mtts_debug_set_kstamp_nth(stamps.data_ptr(), n)
        model.synthesise(x, x_len, 10, speaker=0)
        torch.cuda.synchronize()
        lib.mtts_debug_kstamp_info(info)
        M, N, K, bm, st, ks, taps, flags = list(info)
        key = (M, N, K, flags)
        seen[key] = seen.get(key, 0) + 1
        if seen[key] > args.per_shape:
            continue
        s = stamps.cpu().view(-1, 8)
        s = s[s[:, 3] != 0]
        if len(s) == 0:
            print(n, "no stamps", list(info))
            continue
        fl = "".join(c for c, b in zip("LrIFgtsS", (1, 2, 4, 8, 16, 32, 64, 128)) if flags & b)
        print(f"{n:4d} {M:6d} {N:5d} {K:5d} {bm:3d} {st:2d} {ks:2d} {fl:6s}| " + fmt_stamps(s, K // 32), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--insitu", action="store_true", help="stamp launches inside synthesise() (flags: L LayerNorm, r residual image, I image out, F fp32 out, g GN statistics, t Block1D tail, s SnakeBeta, S row moments out)")
    ap.add_argument("--first", type=int, default=300)
    ap.add_argument("--count", type=int, default=40)
    ap.add_argument("--per-shape", type=int, default=2)
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--flags", default="", help="extra compile flags for --build, e.g. -DMTTS_EPI_ROLL")
    ap.add_argument("--name", default="kstamp", help="tools/ab/<name>.so")
    ap.add_argument("--nostamp", action="store_true", help="--build without -DMTTS_KSTAMP")
    ap.add_argument("--thrash", action="store_true", help="run the other shapes' kernels between the repeats (cold instruction cache)")
    args = ap.parse_args()
    if args.build:
        return build(args.flags.split(), args.name, args.nostamp)
    import torch
    hip = importlib.import_module(PKG + "._hip")
    lib = hip.load()
    lib.mtts_debug_set_kstamp.argtypes = [C.c_void_p]
    lib.mtts_debug_set_kstamp.restype = None
    dev = torch.device("cuda")
    B = args.batch
    if args.insitu:
        return insitu(args, lib, torch, dev)
    shapes = [  # name, T, C, ntaps, N, kind
        ("L0 conv k3 384->384", 322, 384, 3, 384, "plain"),
        ("L0 qkv 384->1152 (LN)", 322, 384, 1, 1152, "ln"),
        ("L0 out 384->384 (+res, image)", 322, 384, 1, 384, "res"),
        ("L0 ff1 384->1536 (LN, snake, image)", 322, 384, 1, 1536, "ff1"),
        ("L0 ff2 1536->384 (+res, image)", 322, 1536, 1, 384, "res"),
        ("L1 conv k3 384->384", 161, 384, 3, 384, "plain"),
        ("L1 qkv", 161, 384, 1, 1152, "ln"),
        ("L1 out", 161, 384, 1, 384, "res"),
        ("L1 ff1", 161, 384, 1, 1536, "ff1"),
        ("L1 ff2", 161, 1536, 1, 384, "res"),
    ]
    print(f"{'shape':38s} {'WGs':>5s} {'nk':>3s} | {'first tile':>10s} {'cyc/k-step':>10s} {'epilogue':>9s} | {'WG us':>6s} {'span us':>7s} {'GHz':>5s}")
    calls = []
    for name, T, Cc, nt, N, kind in shapes:
        a = torch.randn(B * T, Cc, device=dev)
        w = torch.randn(N, Cc, nt, device=dev) * (Cc * nt) ** -0.5 if nt > 1 else torch.randn(N, Cc, device=dev) * Cc ** -0.5
        bias = torch.randn(N, device=dev)
        kw = dict(B=B, T_in=T)
        if kind in ("ln", "ff1"):
            kw.update(a_mean=torch.zeros(B * T, device=dev), a_rstd=torch.ones(B * T, device=dev))
        if kind == "ff1":
            kw.update(act=3, p0=torch.ones(N, device=dev), p1=torch.ones(N, device=dev), want_f32=False, want_p16=True)
        if kind == "res":
            kw.update(res=torch.randn(B * T, N, device=dev), want_p16=True, stats_out=True)
        calls.append((a, w, bias, kw))
    for i, (name, T, Cc, nt, N, kind) in enumerate(shapes):
        a, w, bias, kw = calls[i]
        stamps = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
        for rep in range(3):                                   # the last launch is the one read
            if args.thrash:                                    # the other shapes' kernels in between, as inside a decoder block
                for j, (a2, w2, b2, kw2) in enumerate(calls):
                    if j != i:
                        hip.gemm_p16(a2, w2, b2, **kw2)
            stamps.zero_()
            lib.mtts_debug_set_kstamp(stamps.data_ptr())
            hip.gemm_p16(a, w, bias, **kw)
        torch.cuda.synchronize()
        s = stamps.cpu().view(-1, 8)
        live = s[:, 3] != 0
        s = s[live]
        nk = nt * Cc // 32
        first = (s[:, 1] - s[:, 0]).tolist()
        loop = ((s[:, 2] - s[:, 1]).double() / max(nk - 1, 1)).tolist()       # stamp 1 sits after the first tile's barrier
        epi = (s[:, 3] - s[:, 2]).tolist()
        life = (s[:, 3] - s[:, 0]).double()
        rt = (s[:, 5] - s[:, 4]).double().clamp_min(1)                        # 100 MHz ticks
        ghz = float((life / rt).median()) * 0.1
        span_us = float(s[:, 5].max() - s[:, 4].min()) / 100.0
        med = statistics.median
        print(f"{name:38s} {len(s):5d} {nk:3d} | {med(first):10.0f} {med(loop):10.0f} {med(epi):9.0f} | "
              f"{float(life.median()) / (ghz * 1e3):6.1f} {span_us:7.1f} {ghz:5.2f} | park {med((s[:, 6] - s[:, 2]).tolist()):5.0f} "
              f"chunk0 {med((s[:, 7] - s[:, 6]).tolist()):6.0f} rest {med((s[:, 3] - s[:, 7]).tolist()):6.0f}", flush=True)


if __name__ == "__main__":
    main()
