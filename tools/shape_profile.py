#!/usr/bin/env python3
"""Per-shape table of one bench step on the GPU box: every launch of the library's event pass grouped by
(class, algorithmic FLOPs, compulsory bytes) = one GEMM / attention shape.
    python tools/shape_profile.py [--batch 32] [--steps 3] [--solver euler] [--n-timesteps 10]
Environment switches of the library (MTTS_FOLD, MTTS_FOLD_ALIGN, MTTS_GEMM_BM, ...) apply as usual."""
import argparse
import collections
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--solver", default="euler")
    ap.add_argument("--n-timesteps", type=int, default=10)
    args = ap.parse_args()
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dev = torch.device("cuda")
    hp = hparams.prod_v20(n_spks=1)
    model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(synthetic.make_state_dict(hp, seed=7), strict=True)
    model = model.to(dev).eval()
    model.decoder.solver = args.solver
    x, x_len, _ = synthetic.make_inputs(hp, args.batch, args.tokens, seed=1234)
    x, x_len = x.to(dev), x_len.to(dev)
    for _ in range(2):
        model.synthesise(x, x_len, args.n_timesteps, speaker=0)
    hip = model.hip
    hip.prof_enable(True)
    hip.prof_reset()
    torch.cuda.synchronize()
    for _ in range(args.steps):
        model.synthesise(x, x_len, args.n_timesteps, speaker=0)
    torch.cuda.synchronize()
    recs = hip.prof_records()
    hip.prof_enable(False)
    groups = collections.OrderedDict()
    for k, ms, fl, by in recs:
        g = groups.setdefault((k, fl, by), [0, 0.0])
        g[0] += 1
        g[1] += ms
    names = {0: "gemm", 1: "attn", 2: "elem"}
    tot = {0: 0.0, 1: 0.0, 2: 0.0}
    print(f"{'class':5s} {'GFLOP':>9s} {'MB':>8s} {'n/step':>7s} {'us':>8s} {'TFLOP/s':>8s} {'ms/step':>8s}")
    rows = sorted(groups.items(), key=lambda kv: -kv[1][1])
    for (k, fl, by), (n, ms) in rows:
        tot[k] += ms / args.steps
        if k == 2 and ms / args.steps < 0.02:
            continue
        us = ms * 1e3 / n
        print(f"{names[k]:5s} {fl/1e9:9.3f} {by/1e6:8.2f} {n/args.steps:7.1f} {us:8.1f} {(fl/us/1e6 if us else 0):8.1f} {ms/args.steps:8.3f}")
    print("ms/step by class:", {names[k]: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
