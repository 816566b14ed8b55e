#!/usr/bin/env python3
"""Per-instantiation table of one bench step (round-2 verdict item 5a): every launch of the library's event pass grouped by the
kernel instantiation its launcher chose (mtts_prof_tags) -- launches per step, mean microseconds, GFLOP per launch, TFLOP/s,
fraction of the arithmetic's matrix peak -- joined, when a tools/profile_gpu.sh directory is given, with rocprofv3's own mean
duration, the MFMA-busy fraction and the HBM bytes per launch of the same instantiation.

    python tools/kernel_table.py [--steps 3] [--batch 32] [--prof gpurun_out/prof_<tag>] [--out profiles/<tag>_kernel_table.md]
"""
import argparse
import collections
import csv
import glob
import importlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"
PEAK = {2: 2500.0 / 3, 16: 2500.0, 17: 2500.0, 1: 2500.0, 6: 2500.0 / 6, 3: 2500.0 / 3, 0: 157.3}


def short(name):
    return name.split("(")[0].replace("void ", "").replace("mtts::", "").strip()


def rocprof(prof_dir):
    """{instantiation: {"avg_us":, "busy":, "read_mb":, "write_mb":}} from a tools/profile_gpu.sh output directory."""
    out = collections.defaultdict(dict)
    d = Path(prof_dir)
    for f in glob.glob(str(d / "trace" / "*" / "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])]["avg_us"] = float(r["AverageNs"]) / 1e3
            out[short(r["Name"])]["calls"] = int(r["Calls"])
    for sub, counter, key, scale in (("fetch", "FETCH_SIZE", "read_mb", 2.0 * 1024 / 1e6), ("write", "WRITE_SIZE", "write_mb", 1024 / 1e6)):
        agg = collections.defaultdict(list)
        for f in glob.glob(str(d / sub / "*" / "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k][key] = sum(v) / len(v) * scale            # FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md), KiB -> MB
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(str(d / "mfma" / "*" / "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if v.get("GRBM_GUI_ACTIVE", 0) > 0:
            out[k]["busy"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--prof", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--join", default=None, help="a <table>.json written by an earlier run: join it with --prof on a machine without a GPU")
    a = ap.parse_args()
    if a.join:
        rows, rp = json.loads(Path(a.join).read_text()), rocprof(a.prof)
        f = lambda v, fmt: "" if v is None else format(v, fmt)
        lines = ["| instantiation | launches/step | mean us (events) | GFLOP/launch | TFLOP/s | frac of peak | % of kernel time | mean us (rocprofv3) | MFMA busy | HBM read MB | HBM write MB |",
                 "|---|---|---|---|---|---|---|---|---|---|---|"]
        for r in rows:
            q = rp.get(r["instantiation"], {})
            r.update({k: q.get(k) for k in ("avg_us", "busy", "read_mb", "write_mb")})
            lines.append(f"| `{r['instantiation']}` | {r['launches_per_step']:.0f} | {r['mean_us']:.1f} | {r['gflop_per_launch']:.3f} | {r['tflops']:.1f} | {r['frac']:.3f} | "
                         f"{r['time_pct']:.1f} | {f(r.get('avg_us'), '.1f')} | {f(r.get('busy'), '.3f')} | {f(r.get('read_mb'), '.2f')} | {f(r.get('write_mb'), '.2f')} |")
        text = "\n".join(lines) + "\n"
        print(text)
        if a.out:
            Path(a.out).write_text(text)
            Path(a.out).with_suffix(".json").write_text(json.dumps(rows, indent=1))
        return
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dev = torch.device("cuda")
    hp = hparams.prod_v20(n_spks=1)
    model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(synthetic.make_state_dict(hp, seed=7), strict=True)
    model = model.to(dev).eval()
    model.decoder.solver = "euler"
    x, x_len, _ = synthetic.make_inputs(hp, a.batch, 128, seed=1234)
    x, x_len = x.to(dev), x_len.to(dev)
    for _ in range(2):
        model.synthesise(x, x_len, 10, speaker=0)
    hip = model.hip
    hip.prof_enable(True)
    hip.prof_reset()
    torch.cuda.synchronize()
    for _ in range(a.steps):
        model.synthesise(x, x_len, 10, speaker=0)
    torch.cuda.synchronize()
    recs, tags = hip.prof_records(), hip.prof_tags()
    hip.prof_enable(False)
    peak = PEAK[hip.gemm_terms()]
    groups = collections.OrderedDict()
    for (k, ms, fl, by), tag in zip(recs, tags):
        g = groups.setdefault(tag if tag != "-" else {0: "(untagged GEMM)", 1: "(untagged attention)", 2: "(streaming / glue kernels)"}[k], [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += ms
        g[2] += fl
        g[3] += by
    rp = rocprof(a.prof) if a.prof else {}
    total_ms = sum(g[1] for g in groups.values())
    lines = [f"# Per-instantiation table, batch {a.batch}, euler/10, {a.steps} steps (event pass: {total_ms / a.steps:.2f} ms of kernels per step; "
             f"peak for this arithmetic {peak:.0f} TFLOP/s fp32-equivalent)", "",
             "| instantiation | launches/step | mean us (events) | GFLOP/launch | TFLOP/s | frac of peak | % of kernel time | mean us (rocprofv3) | MFMA busy | HBM read MB | HBM write MB |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    rows = []
    for tag, (n, ms, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        tf_s = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        r = rp.get(tag, {})
        rows.append({"instantiation": tag, "launches_per_step": n / a.steps, "mean_us": ms * 1e3 / n, "gflop_per_launch": fl / n / 1e9,
                     "tflops": tf_s, "frac": tf_s / peak, "time_pct": 100.0 * ms / total_ms, **{k: r.get(k) for k in ("avg_us", "busy", "read_mb", "write_mb")}})
        f = lambda v, fmt: "" if v is None else format(v, fmt)
        lines.append(f"| `{tag}` | {n / a.steps:.0f} | {ms * 1e3 / n:.1f} | {fl / n / 1e9:.3f} | {tf_s:.1f} | {tf_s / peak:.3f} | {100.0 * ms / total_ms:.1f} | "
                     f"{f(r.get('avg_us'), '.1f')} | {f(r.get('busy'), '.3f')} | {f(r.get('read_mb'), '.2f')} | {f(r.get('write_mb'), '.2f')} |")
    text = "\n".join(lines) + "\n"
    print(text)
    if a.out:
        Path(a.out).write_text(text)
        Path(a.out).with_suffix(".json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
