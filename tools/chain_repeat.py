#!/usr/bin/env python3
"""GPU diagnostic: the chain launch alone (mtts_tblock_chain), the same inputs N times -- are the outputs bitwise repeatable, and
if not, WHERE do they differ (rows -> workgroup / row tile, channels -> wave / tile)?
    python tools/chain_repeat.py [--rows 10304] [--qb 48] [--ch 256] [--runs 100] [--thrash]"""
import argparse, importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
PKG = "matcha-tts-24k_amd"


def describe(name, d, qb):
    rows = torch.nonzero(d.amax(dim=1) > 0).flatten()
    cols = torch.nonzero(d.amax(dim=0) > 0).flatten()
    wgs = sorted(set((rows // qb).tolist()))
    return (f"{name}: {int((d > 0).sum())} values, rows {int(rows.min())}..{int(rows.max())} ({rows.numel()}) in workgroups {wgs[:8]}"
            f"{'...' if len(wgs) > 8 else ''} ({len(wgs)}), row-in-tile {sorted(set((rows % qb).tolist()))[:16]}, "
            f"cols {int(cols.min())}..{int(cols.max())} ({cols.numel()}) col/16 {sorted(set((cols // 16).tolist()))[:24]}, max {float(d.max()):.3e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10304)
    ap.add_argument("--qb", type=int, default=48)
    ap.add_argument("--ch", type=int, default=256)
    ap.add_argument("--runs", type=int, default=100)
    ap.add_argument("--thrash", action="store_true", help="sweep 1 GiB between launches (weights and activations leave the L2s)")
    args = ap.parse_args()
    hip = importlib.import_module(PKG + "._hip")
    from test_hip_chain import make_case
    dev = torch.device("cuda")
    case = make_case(args.rows, 384, 384, 1152, seed=5)
    att, x = case[0].to(dev), case[1].to(dev)
    big = torch.empty(256 << 20, dtype=torch.float32, device=dev) if args.thrash else None
    first = None
    bad = 0
    for r in range(args.runs):
        if big is not None:
            big.add_(1.0)
        xo, qkv = hip.tblock_chain(att, x, *case[2:10], w_qkv=case[10], b_qkv=case[11], qb=args.qb, ch=args.ch)
        if first is None:
            first = (xo.clone(), qkv.clone())
            continue
        dx, dq = (xo - first[0]).abs(), (qkv - first[1]).abs()
        if bool((dx > 0).any()) or bool((dq > 0).any()):
            bad += 1
            msg = []
            if bool((dx > 0).any()):
                msg.append(describe("x_out", dx, args.qb))
            if bool((dq > 0).any()):
                msg.append(describe("qkv", dq, args.qb))
            print(f"run {r}: DIFFERS  " + "  |  ".join(msg), flush=True)
    print(f"result: {bad} of {args.runs - 1} repeats differ (rows {args.rows}, qb {args.qb}, ch {args.ch}, thrash {args.thrash})", flush=True)


if __name__ == "__main__":
    main()
