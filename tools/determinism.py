#!/usr/bin/env python3
"""GPU diagnostic: is the full-size synthesis (B=32, Tx=128, euler/N) bitwise repeatable?  Prints, per repeat, where it differs from
the first run (utterances, frame range, size of the difference).  Environment switches (MTTS_CHAIN, MTTS_ATTN_WHOLE, ...) are read
by the library, so A/B them from the shell:
    MTTS_CHAIN=0 python tools/determinism.py --runs 6"""
import argparse, importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--spks", type=int, default=10)
    args = ap.parse_args()
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dev = torch.device("cuda")
    hp = hparams.prod_v20(n_spks=args.spks)
    sd = synthetic.make_state_dict(hp, seed=7)
    m = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m.decoder.solver = "euler"
    x, x_len, spk = synthetic.make_inputs(hp, args.batch, 128, seed=1234)
    z = synthetic.cpu_noise((args.batch, 100, 640)).to(dev)
    x, x_len, spk = x.to(dev), x_len.to(dev), spk.to(dev)
    first, bad = None, 0
    for r in range(args.runs):
        out = m.synthesise(x, x_len, args.steps, speaker=spk if args.spks > 1 else 0, z=z)["mel"]
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            print(f"run 0: shape {tuple(out.shape)} finite {bool(torch.isfinite(out).all())} flags {m.hip.range_flags().tolist()}", flush=True)
            continue
        d = (out - first).abs()
        if not bool((d > 0).any()):
            print(f"run {r}: identical", flush=True)
            continue
        bad += 1
        per_b = d.amax(dim=(1, 2))
        ub = torch.nonzero(per_b > 0).flatten().tolist()
        desc = []
        for b in ub[:6]:
            fr = torch.nonzero(d[b].amax(dim=0) > 0).flatten()
            desc.append(f"b={b} frames {int(fr.min())}..{int(fr.max())} ({fr.numel()}) max {float(per_b[b]):.3e}")
        print(f"run {r}: DIFFERS in {len(ub)} utterances, {int((d > 0).sum())} values; " + "; ".join(desc), flush=True)
    print(f"result: {bad} of {args.runs - 1} repeats differ", flush=True)


if __name__ == "__main__":
    main()
