#!/usr/bin/env python3
"""GPU diagnostic of the chain launch inside the model: determinism across runs, range flags, agreement with the four-launch path.
    python tools/chain_debug.py [--batch 32] [--qb 32] [--ch 128] [--runs 4]"""
import argparse, importlib, os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--qb", default="")
    ap.add_argument("--ch", default="128")
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dev = torch.device("cuda")
    hp = hparams.prod_v20(n_spks=1)
    sd = synthetic.make_state_dict(hp, seed=7)

    def make(env):
        for k, v in env.items():
            os.environ[k] = v
        m = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        m.hip
        for k in env:
            del os.environ[k]
        m.decoder.solver = "euler"
        m.range_policy = "ignore"
        return m
    env = {"MTTS_CHAIN_MIN_ROWS": "0", "MTTS_CHAIN_CH": args.ch}
    if args.qb:
        env["MTTS_CHAIN_QB"] = args.qb
    fused = make(env)
    plain = make({"MTTS_CHAIN": "0"})
    x, x_len, _ = synthetic.make_inputs(hp, args.batch, 128, seed=1234)
    x, x_len = x.to(dev), x_len.to(dev)
    ref = plain.synthesise(x, x_len, args.steps, speaker=0)["mel"]
    print("plain flags", plain.hip.range_flags().tolist(), "finite", bool(torch.isfinite(ref).all()))
    first = None
    for r in range(args.runs):
        out = fused.synthesise(x, x_len, args.steps, speaker=0)["mel"]
        flags = fused.hip.range_flags().tolist()
        err = (out - ref).abs().max().item()
        same = None if first is None else bool(torch.equal(out, first))
        worst = (out - ref).abs().amax(dim=(1, 2))
        print(f"run {r}: flags {flags} max|fused - plain| {err:.3e} finite {bool(torch.isfinite(out).all())} same_as_run0 {same} "
              f"worst utterances {torch.topk(worst, min(3, args.batch)).indices.tolist()}")
        if first is None:
            first = out.clone()


if __name__ == "__main__":
    main()
