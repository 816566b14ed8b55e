#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernel on the prod shapes of the path (B=32, T_pad=640).  Run on the GPU box:
    python tools/gemm_shapes.py [--reps 20]
Times come from HIP events around back-to-back launches with pre-packed weights (mtts_gemm_f32 with d_w = NULL)."""
import argparse
import ctypes as C
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hip = importlib.import_module("matcha-tts-24k_amd._hip")

SHAPES = [  # name, B, T_in, T_out, C, ntaps, stride, N, mask, norm, act, res, count per NFE
    ("L0 conv1 down0 (x|mu)", 32, 640, 640, 200, 3, 1, 384, 1, 0, 0, 0, 1),
    ("L0 conv k3 C384 mask", 32, 640, 640, 384, 3, 1, 384, 1, 0, 0, 0, 2),
    ("L0 conv k3 C384", 32, 640, 640, 384, 3, 1, 384, 0, 0, 0, 0, 2),
    ("L0 conv1 up1 C768", 32, 640, 640, 768, 3, 1, 384, 1, 0, 0, 0, 1),
    ("L0 res 1x1 C768", 32, 640, 640, 768, 1, 1, 384, 1, 0, 0, 0, 1),
    ("L0 res 1x1 C200", 32, 640, 640, 200, 1, 1, 384, 1, 0, 0, 0, 1),
    ("L0 qkv", 32, 640, 640, 384, 1, 1, 1152, 0, 1, 0, 0, 4),
    ("L0 attn out", 32, 640, 640, 384, 1, 1, 384, 0, 0, 0, 1, 4),
    ("L0 ff1 snake", 32, 640, 640, 384, 1, 1, 1536, 0, 1, 3, 0, 4),
    ("L0 ff2", 32, 640, 640, 1536, 1, 1, 384, 0, 0, 0, 1, 4),
    ("L0->1 down s2", 32, 640, 320, 384, 3, 2, 384, 1, 0, 0, 0, 1),
    ("L0 final proj", 32, 640, 640, 384, 1, 1, 100, 0, 0, 0, 1, 1),
    ("L1 conv k3 C384 mask", 32, 320, 320, 384, 3, 1, 384, 1, 0, 0, 0, 4),
    ("L1 conv k3 C384", 32, 320, 320, 384, 3, 1, 384, 0, 0, 0, 0, 4),
    ("L1 conv1 up0 C768", 32, 320, 320, 768, 3, 1, 384, 1, 0, 0, 0, 1),
    ("L1 res 1x1 C384", 32, 320, 320, 384, 1, 1, 384, 1, 0, 0, 0, 3),
    ("L1 qkv", 32, 320, 320, 384, 1, 1, 1152, 0, 1, 0, 0, 8),
    ("L1 attn out", 32, 320, 320, 384, 1, 1, 384, 0, 0, 0, 1, 8),
    ("L1 ff1 snake", 32, 320, 320, 384, 1, 1, 1536, 0, 1, 3, 0, 8),
    ("L1 ff2", 32, 320, 320, 1536, 1, 1, 384, 0, 0, 0, 1, 8),
    ("L1 up convT phase", 32, 320, 320, 384, 2, 1, 384, 1, 0, 0, 0, 2),
]


FUSED = False
TERMS = -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--terms", type=int, default=-1, help="0 fp32 MFMA, 6 / 3 split-bf16 MFMA, -1 library default")
    ap.add_argument("--fused-stats", action="store_true", help="LayerNorm stats via epilogue partials (as the decoder runs them)")
    args = ap.parse_args()
    global FUSED, TERMS
    FUSED = args.fused_stats
    TERMS = args.terms
    lib = hip.load()
    dev = torch.device("cuda")
    tot_t = tot_f = 0.0
    print(f"{'shape':26s} {'M':>6s} {'N':>5s} {'K':>5s} {'us':>8s} {'TFLOP/s':>8s}  x/NFE")
    for name, B, Ti, To, Cc, nt, st, N, mk, nm, act, res, cnt in SHAPES:
        a = torch.randn(B * Ti, Cc, device=dev)
        w = torch.randn(N, Cc, nt, device=dev) * (Cc * nt) ** -0.5 if nt > 1 else torch.randn(N, Cc, device=dev) * Cc ** -0.5
        bias = torch.randn(N, device=dev)
        mask = torch.ones(B * Ti, device=dev) if mk else None
        mean = torch.zeros(B * Ti, device=dev) if nm else None
        rstd = torch.ones(B * Ti, device=dev) if nm else None
        p0 = torch.ones(N, device=dev) if act == 3 else None
        p1 = torch.ones(N, device=dev) if act == 3 else None
        r = torch.randn(B * To, N, device=dev) if res else None
        out = torch.empty(B * To, N, device=dev)
        part_in = bool(nm) and FUSED and Cc % 64 == 0
        st_out = bool(res) and FUSED and N % 64 == 0
        part = torch.zeros(B * Ti, max(Cc // 64, 1), 2, device=dev)
        part[:, :, 1] = 64.0
        stats = torch.empty(B * To, max(N // 64, 1), 2, device=dev)
        packed = torch.empty(lib.mtts_gemm_packed_bytes(N, Cc, nt), dtype=torch.uint8, device=dev)
        taps = (C.c_int * nt)(*[j - nt // 2 for j in range(nt)])
        s = hip.stream_ptr()

        def launch(wptr):
            hip.check(lib.mtts_gemm_f32(hip.ptr(a), Cc, B, Ti, Cc, nt, taps, st, To, hip.ptr(mask), hip.ptr(mean) if not part_in else None, hip.ptr(rstd) if not part_in else None, hip.ptr(part) if part_in else None, part.shape[1] if part_in else 0, wptr,
                                        packed.data_ptr(), hip.ptr(bias), N, act, hip.ptr(p0), hip.ptr(p1), hip.ptr(r), N if res else 0,
                                        None, 1.0, hip.ptr(out), N, hip.ptr(stats) if st_out else None, TERMS, s))
        launch(hip.ptr(w.contiguous()))
        for _ in range(3):
            launch(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            launch(None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        fl = 2.0 * B * To * N * nt * Cc
        print(f"{name:26s} {B*To:6d} {N:5d} {nt*Cc:5d} {us:8.1f} {fl/us/1e6:8.1f}  {cnt}")
        tot_t += us * cnt
        tot_f += fl * cnt
    print(f"weighted per NFE: {tot_t/1e3:.2f} ms, {tot_f/tot_t/1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
