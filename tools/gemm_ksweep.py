#!/usr/bin/env python3
"""Fixed cost vs K slope of the GEMM kernel: time(K) for M=20480 (and 10240), N in {384, 1536}."""
import ctypes as C, importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hip = importlib.import_module("matcha-tts-24k_amd._hip")
lib = hip.load()
dev = torch.device("cuda")
for (B, T, N, res) in ((32, 640, 384, 1), (32, 640, 1536, 0), (32, 320, 384, 1)):
    for K in (32, 64, 128, 256, 384, 768, 1536):
        a = torch.randn(B * T, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; bias = torch.randn(N, device=dev)
        r = torch.randn(B * T, N, device=dev) if res else None
        out = torch.empty(B * T, N, device=dev)
        packed = torch.empty(lib.mtts_gemm_packed_bytes(N, K, 1), dtype=torch.uint8, device=dev)
        taps = (C.c_int * 1)(0); s = hip.stream_ptr()
        def launch(wp):
            hip.check(lib.mtts_gemm_f32(hip.ptr(a), K, B, T, K, 1, taps, 1, T, None, None, None, None, 0, wp, packed.data_ptr(), hip.ptr(bias), N, 0, None, None,
                                        hip.ptr(r), N if res else 0, None, 1.0, hip.ptr(out), N, None, -1, s))
        launch(hip.ptr(w))
        for _ in range(5): launch(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): launch(None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"M {B*T:6d} N {N:5d} K {K:5d}  {us:8.1f} us  {2.0*B*T*N*K/us/1e6:7.1f} TFLOP/s")
