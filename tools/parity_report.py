#!/usr/bin/env python3
"""Mel error of the HIP path against the golden vectors (reference's own code) for each GEMM arithmetic mode.
Run on the GPU box:  MTTS_GEMM_TERMS=<0|6|3> python tools/parity_report.py"""
import importlib, os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
hp_m = importlib.import_module("matcha-tts-24k_amd.hparams"); syn = importlib.import_module("matcha-tts-24k_amd.synthetic")
inf = importlib.import_module("matcha-tts-24k_amd.inference")
dev = torch.device("cuda")
hp = hp_m.prod_v20(1); sd = syn.make_state_dict(hp)
m = inf.MatchaTTSInfer(**hp.as_reference_kwargs()); m.load_state_dict(sd); m = m.to(dev).eval()
g = np.load(ROOT / "tests/golden/prod_synth.npz")
x, xl, _ = syn.make_inputs(hp, 1, 128); z = syn.cpu_noise((1, 100, 640)).to(dev)
out = [f"GEMM arithmetic mode (terms) = {m.hip.gemm_terms()}"]
for solver, steps, key in (("euler", 2, "mel_euler2"), ("euler", 10, "mel_euler10"), ("midpoint", 4, "mel_midpoint4")):
    m.decoder.solver = solver
    mel = m.synthesise(x.to(dev), xl.to(dev), steps, speaker=0, z=z)["mel"].cpu()
    ref = torch.from_numpy(g[key])
    out.append(f"  {solver}/{steps}: max-abs {float((mel-ref).abs().max()):.3e}  mean-abs {float((mel-ref).abs().mean()):.3e}  (|mel| max {float(ref.abs().max()):.1f})")
print("\n".join(out))
