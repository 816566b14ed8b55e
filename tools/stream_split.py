#!/usr/bin/env python3
"""Experiment: one batch of B utterances as S independent sub-batches on S HIP streams (utterances are independent, so the
sub-batches' kernels can fill each other's ramp-up / tail / inter-kernel gaps).  python tools/stream_split.py [--batch 32]"""
import os
os.environ.setdefault("MTTS_CHAIN_PAIR", "0")       # several contexts share the card here: a pair-form chain launch needs the chip to itself
import argparse
import importlib
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--splits", default="1,2,4")
    ap.add_argument("--threads", type=int, default=1, help="1: one host thread AND one model (= one mtts_ctx) per stream; 0: one thread issues all streams")
    args = ap.parse_args()
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dev = torch.device("cuda")
    hp = hparams.prod_v20(n_spks=1)
    sd = synthetic.make_state_dict(hp, seed=7)

    def make_model():
        m = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        m.decoder.solver = "euler"
        return m

    # a context is single-threaded (include/mtts.h "Threading"; the entry points refuse concurrent use): with host threads every
    # stream gets a model of its own
    n_models = max(int(v) for v in args.splits.split(",")) if args.threads else 1
    models = [make_model() for _ in range(n_models)]
    x, x_len, _ = synthetic.make_inputs(hp, args.batch, 128, seed=1234)
    x, x_len = x.to(dev), x_len.to(dev)
    for S in [int(v) for v in args.splits.split(",")]:
        streams = [torch.cuda.Stream() for _ in range(S)]
        per = args.batch // S

        def part(i):
            with torch.cuda.stream(streams[i]):
                models[i if args.threads else 0].synthesise(x[i * per:(i + 1) * per], x_len[i * per:(i + 1) * per], 10, speaker=0)

        def step():
            if args.threads and S > 1:
                th = [threading.Thread(target=part, args=(i,)) for i in range(S)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            else:
                for i in range(S):
                    part(i)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        print(f"B={args.batch} split into {S} x {per} on {S} stream(s), threads={args.threads}: {el*1e3:.2f} ms/step "
              f"{args.batch*320/el:.0f} frames/s", flush=True)


if __name__ == "__main__":
    main()
