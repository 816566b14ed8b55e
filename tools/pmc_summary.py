#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh run into profiles/ (tracked): per-kernel time from the rocprofv3 kernel trace and
HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly
half of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.

    python tools/pmc_summary.py gpurun_out/prof_<tag> <tag>
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("mtts::", "")
    return n.strip()


def counters(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(str(d / "*" / "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    src, tag = Path(sys.argv[1]), sys.argv[2]
    out = ROOT / "profiles"
    out.mkdir(exist_ok=True)
    stats = glob.glob(str(src / "trace" / "*" / "*_kernel_stats.csv"))[0]
    shutil.copy(stats, out / f"{tag}_kernel_stats.csv")
    times = {}
    for r in csv.DictReader(open(stats)):
        times[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    fetch, write = counters(src / "fetch", "FETCH_SIZE"), counters(src / "write", "WRITE_SIZE")
    rows, g_bytes, g_n = [], 0.0, 0
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        n = len(fetch[k])
        rd = 2.0 * sum(fetch[k]) / n * 1024.0
        wr = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0]))) * 1024.0
        calls, avg_us, pct = times.get(k, (0, 0.0, 0.0))
        rows.append({"kernel": k, "launches_in_pmc_run": n, "hbm_read_mb_per_launch": round(rd / 1e6, 2),
                     "hbm_write_mb_per_launch": round(wr / 1e6, 2), "avg_us_kernel_trace": round(avg_us, 1), "time_pct": pct,
                     "hbm_gb_per_s": round((rd + wr) / max(avg_us, 1e-9) / 1e3, 1)})
        if k.startswith(("gemm_f32_kernel", "gemm_p16_kernel", "tblock_chain_kernel")):     # the launches bench.py counts as GEMM
            g_bytes += (rd + wr) * n
            g_n += n
    # MFMA-pipe busy fraction per kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024)
    mfma = []
    mf_files = glob.glob(str(src / "mfma" / "*" / "*_counter_collection.csv"))
    if mf_files:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in mf_files:
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
            gui, busy = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            if busy > 0 and gui > 0:
                mfma.append({"kernel": k, "mfma_busy_fraction": round(busy / (gui / 8.0 * 1024.0), 4)})
    summary = {"tag": tag, "mfma_busy": mfma, "command": "python3 bench.py --steps N --warmup 1 --no-cpu-baseline --no-events (under rocprofv3)",
               "corrections": "FETCH_SIZE x2 (gfx950), KiB -> bytes x1024", "gemm_hbm_mb_per_launch": round(g_bytes / max(g_n, 1) / 1e6, 2),
               "gemm_launches": g_n, "kernels": rows[:16]}
    (out / f"{tag}_pmc_traffic.json").write_text(json.dumps(summary, indent=1))
    (out / "pmc_traffic_latest.json").write_text(json.dumps(summary, indent=1))
    for name in ("trace_bench.json",):
        if (src / name).exists():
            shutil.copy(src / name, out / f"{tag}_{name}")
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
