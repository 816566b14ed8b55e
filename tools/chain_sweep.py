#!/usr/bin/env python3
"""Time the transformer-block chain kernel alone (mtts_tblock_chain_timed) over row counts and workgroup shapes.
    python tools/chain_sweep.py [--rows 64,512,2048,5152,10304] [--cfg 64:128,32:128,48:256,32:256] [--no-qkv]"""
import argparse, importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hip = importlib.import_module("matcha-tts-24k_amd._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="64,512,2048,5152,10304")
    ap.add_argument("--cfg", default="64:128,32:128,48:256,32:256")
    ap.add_argument("--no-qkv", action="store_true")
    ap.add_argument("--repeat", type=int, default=20)
    ap.add_argument("--pair", action="store_true", help="the pair form (two workgroups per row tile)")
    a = ap.parse_args()
    C, inner, nq = 384, 384, 0 if a.no_qkv else 1152
    g = torch.Generator().manual_seed(1)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    w_out, b_out = r(C, inner, sc=inner ** -0.5), r(C)
    w1, b1 = r(4 * C, C, sc=C ** -0.5), r(4 * C)
    p0, p1 = torch.exp(r(4 * C, sc=0.2)), 1.0 / (torch.exp(r(4 * C, sc=0.2)) + 1e-9)
    w2, b2 = r(C, 4 * C, sc=(4 * C) ** -0.5), r(C)
    wq, bq = (r(nq, C, sc=C ** -0.5), r(nq)) if nq else (None, None)
    flop_row = 2.0 * (C * inner + 8 * C * C + C * nq)
    print(f"{'rows':>6s} {'qb':>3s} {'ch':>4s} {'WGs':>5s} {'us':>8s} {'TF/s':>7s} {'GB/s per WG (weights)':>22s}")
    for cfg in a.cfg.split(","):
        qb, ch = (int(v) for v in cfg.split(":"))
        for M in (int(v) for v in a.rows.split(",")):
            att, x = r(M, inner).cuda(), (r(M, C) * 2 + 0.3).cuda()
            _, _, ms = hip.tblock_chain(att, x, w_out, b_out, w1, b1, p0, p1, w2, b2, w_qkv=wq, b_qkv=bq, qb=qb, ch=ch, repeat=a.repeat, pair=a.pair)
            wbytes = 4.0 * (C * inner + 8 * C * C + C * nq)
            print(f"{M:6d} {qb:3d} {ch:4d} {(M + qb - 1) // qb:5d} {ms * 1e3:8.1f} {flop_row * M / ms / 1e9:7.1f} {wbytes / ms / 1e6:22.1f}")


if __name__ == "__main__":
    main()
