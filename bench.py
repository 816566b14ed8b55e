#!/usr/bin/env python3
"""Headline benchmark: valid mel-frames/s of end-to-end synthesise() (BASELINE.json metric).

One step = one pass of the hot path over one batch of synthetic input: ids on the device -> text encoder ->
durations -> alignment -> CFM decoder ODE solve (euler, n_timesteps=10) -> denormalised mel on the device.
Workload at N=1 = BASELINE.json configs[1]: batch 32 random phoneme sequences of 128 tokens, n_spks=1, fp32,
prod v20 architecture with random-init weights (no checkpoints exist offline), 5 fine frames per token, i.e.
T_pad = 640 decoder frames and 320 valid mel frames per utterance, strict reference padding.

N>1: one rank per GPU; utterances shard data-parallel, every rank synthesises its own 32 utterances (weak scaling) and the
finished mels are all-gathered over RCCL inside the timed step.  Either launch it under torch.distributed.run (RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or plainly as `python bench.py --gpus N`: the parent then starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process before it has
touched the GPU (never an exec after device init), relays the child's output and exits with its return code.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline] [--no-events]
prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _self_launch_if_needed():
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks as a child torch.distributed.run job.
    Runs before torch is imported in this process, so the parent never initialises a device."""
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _self_launch_if_needed()

import torch
import torch.distributed as dist

PKG = "matcha-tts-24k_amd"
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / f16 MFMA, dense
# GEMM arithmetic modes (csrc/gemm_f32.hip): MFMA products executed per fp32-equivalent multiply-accumulate
GEMM_MODES = {0: ("f32", "v_mfma_f32_32x32x2_f32 on fp32 operands", 1, PEAK_F32_MFMA_TFLOPS),
              1: ("f16 operands, fp32 accumulate (OPT-IN reduced precision, MTTS_GEMM_TERMS=1: not the headline arithmetic)",
                  "v_mfma_f32_16x16x32_f16 / 32x32x16_f16, 1 product per MAC", 1, PEAK_16BIT_MFMA_TFLOPS),
              2: ("f32 via 2-term f16 split (fp32 accumulate)", "v_mfma_f32_16x16x32_f16 / 32x32x16_f16, 3 products per MAC", 3, PEAK_16BIT_MFMA_TFLOPS),
              6: ("f32 via 3-term bf16 split (fp32 accumulate)", "v_mfma_f32_32x32x16_bf16, 6 products per MAC", 6, PEAK_16BIT_MFMA_TFLOPS),
              3: ("f32 via 2-term bf16 split (fp32 accumulate, ~2^-17 per product)", "v_mfma_f32_32x32x16_bf16, 3 products per MAC", 3, PEAK_16BIT_MFMA_TFLOPS),
              16: ("f16 storage, fp32 accumulate (BASELINE configs[2] arithmetic: 2-byte activation / weight planes; text encoder fp32-equivalent)",
                   "v_mfma_f32_16x16x32_f16 / 32x32x16_f16, 1 product per MAC", 1, PEAK_16BIT_MFMA_TFLOPS),
              17: ("bf16 storage, fp32 accumulate (BASELINE configs[2] dtype: 2-byte bfloat16 activation / weight planes; text encoder fp32-equivalent)",
                   "v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16, 1 product per MAC", 1, PEAK_16BIT_MFMA_TFLOPS)}
BATCH, N_TOKENS, N_STEPS_ODE, SOLVER = 32, 128, 10, "euler"


def algorithmic_flops(hp, batch, t_pad, tx, nfe):
    """GEMM/conv/attention MACs x 2 (30.08 GFLOP per NFE per utterance at T=640; encoder+DP 4.08)."""
    C = hp.decoder.channels[0]
    T = t_pad

    def R(i, o):
        return 4 * i * o + 3 * o * o

    def TB(tl):
        return 4 * C * C + 8 * C * C + 2 * tl * C

    dec = 2 * (R(2 * hp.n_feats, C) * T + R(C, C) * 3 * T / 2 + R(2 * C, C) * 3 * T / 2 + 4 * TB(T) * T + 8 * TB(T / 2) * T / 2
               + 3 * C * C * T / 2 + 3 * C * C * T / 2 + 4 * C * C * T / 2 + 3 * C * C * T + 3 * C * C * T + C * hp.n_feats * T)
    e = hp.encoder
    H = e.n_channels + hp.spk_emb_dim
    enc = 2 * tx * (e.prenet_layers * e.prenet_kernel_size * e.n_channels ** 2 + e.n_channels ** 2
                    + e.n_layers * (4 * H * H + 2 * tx * H + 2 * e.kernel_size * H * e.filter_channels)
                    + H * e.n_channels + e.n_channels * hp.n_feats
                    + e.dp_kernel_size * (H * e.dp_filter_channels + (e.dp_n_layers - 1) * e.dp_filter_channels ** 2) + e.dp_filter_channels)
    return batch * (nfe * dec + enc)


def pmc_traffic():
    """HBM bytes per GEMM launch from rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in separate runs,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot collect PMC counters itself: the value
    is READ from the committed summary profiles/pmc_traffic_latest.json (written by tools/profile_gpu.sh on an earlier box) and
    is labelled as such in `traffic_source`; null if absent."""
    f = ROOT / "profiles" / "pmc_traffic_latest.json"
    if not f.exists():
        return None, None
    try:
        d = json.loads(f.read_text())
        return d["gemm_hbm_mb_per_launch"], f"profiles/pmc_traffic_latest.json ({d.get('label', 'committed rocprofv3 --pmc summary')}; NOT measured in this run)"
    except Exception:
        return None, None


def cpu_baseline(hp, sd, synthetic):
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores at B=32 (the bench workload) and B=1, as
    BASELINE.md section 4 asks: warm-up, then the median of 3 (B=32, ~10 s each) / 5 (B=1) runs."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import matcha_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    threads = min(avail, 16)            # a 1-GPU box's CPU share is 16 cores; oversubscribing torch's pool is slower
    torch.set_num_threads(threads)
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass

    def timed(b, reps):
        x, x_len, _ = synthetic.make_inputs(hp, b, N_TOKENS, seed=1234)
        ts, frames = [], 0
        with torch.inference_mode():
            for _ in range(reps):
                t0 = time.perf_counter()
                out = O.synthesise(sd, hp, x, x_len, N_STEPS_ODE, speaker=0, solver=SOLVER)
                ts.append(time.perf_counter() - t0)
                frames = int(out["mel_lengths"].sum())
        ts.sort()
        return frames, ts[len(ts) // 2], ts

    with torch.inference_mode():        # warm the thread pool and the allocator
        x, x_len, _ = synthetic.make_inputs(hp, 2, N_TOKENS, seed=1234)
        O.synthesise(sd, hp, x, x_len, 1, speaker=0, solver=SOLVER)
    f1, m1, t1 = timed(1, 5)
    f32_, m32, t32 = timed(BATCH, 3)
    return {"value": round(f32_ / m32, 1), "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "value_b1": round(f1 / m1, 1), "cpu_model": cpu_model,
            "sample": f"oracle/matcha_oracle.py synthesise(Tx={N_TOKENS}, {SOLVER}/{N_STEPS_ODE}), torch CPU fp32, {threads} threads: "
                      f"B={BATCH} median of 3 runs ({', '.join(f'{t:.2f}' for t in t32)} s for {f32_} valid frames) = value; "
                      f"B=1 median of 5 runs ({', '.join(f'{t:.2f}' for t in t1)} s for {f1} frames) = value_b1"}


def main():
    global BATCH, SOLVER, N_STEPS_ODE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP-event pass (roofline = null)")
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (default 32 = BASELINE configs[1]; other values are labelled)")
    ap.add_argument("--with-vocoder", action="store_true",
                    help="NON-DEFAULT: append the Vocos-24k head (random-init weights) to every step: mel -> waveform (configs[3] style)")
    ap.add_argument("--arithmetic", default=None, choices=["f32", "f16-storage", "bf16-storage"],
                    help="f32 (default): fp32-equivalent split arithmetic = configs[1]; f16-storage / bf16-storage: BASELINE configs[2]'s "
                         "16-bit storage mode with fp16 / bfloat16 planes (per-rank shape of the 8-way job: 32 utterances per GPU), labelled "
                         "with its measured mel error")
    ap.add_argument("--solver", default=SOLVER)
    ap.add_argument("--n-timesteps", type=int, default=N_STEPS_ODE)
    args = ap.parse_args()
    if args.arithmetic == "f16-storage":
        os.environ["MTTS_GEMM_TERMS"] = "16"
    elif args.arithmetic == "bf16-storage":
        os.environ["MTTS_GEMM_TERMS"] = "17"
    default_cfg = (args.batch, args.solver, args.n_timesteps) == (BATCH, SOLVER, N_STEPS_ODE) and not args.with_vocoder
    BATCH, SOLVER, N_STEPS_ODE = args.batch, args.solver, args.n_timesteps

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:      # unreachable from the command line (_self_launch_if_needed), kept for importers
        raise SystemExit("--gpus N>1 needs one rank per GPU: run `python bench.py --gpus N` or launch under torch.distributed.run")
    if os.environ.get("MTTS_BENCH_DRYRUN") == "1":
        # launch rehearsal without a device (tests/test_bench_launch.py): the ranks rendezvous over gloo, agree on a value and
        # rank 0 prints one line; no part of the product path runs
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            got = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            got = 0
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "max_rank": got, "gpus_arg": args.gpus}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    # MTTS_DIST_BACKEND=gloo: rehearsal of the N-rank job on fewer cards (ranks share devices, collectives staged through host
    # memory by dp.py) -- labelled in the output, never the scaling measurement, which runs one rank per GPU over RCCL
    backend = os.environ.get("MTTS_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and n_dev < world:
        raise SystemExit(f"--gpus {world} needs {world} devices for RCCL (found {n_dev}); set MTTS_DIST_BACKEND=gloo to rehearse")
    if world > n_dev:           # rehearsal: ranks share a card, and the pair form of the chain launch needs the chip to itself
        os.environ.setdefault("MTTS_CHAIN_PAIR", "0")
    local = local % max(n_dev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    dp = importlib.import_module(PKG + ".dp")

    hp = hparams.prod_v20(n_spks=1)
    sd = synthetic.make_state_dict(hp, seed=7)
    model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.decoder.solver = SOLVER

    # this rank's shard of the global batch (different utterances per rank; same per-rank work: weak scaling)
    x_all, len_all, _ = synthetic.make_inputs(hp, BATCH * world, N_TOKENS, seed=1234)
    sl = dp.shard_slice(BATCH * world, world, rank)
    x, x_len = x_all[sl].to(dev), len_all[sl].to(dev)

    vocoder = None
    if args.with_vocoder:
        vocoder = inference.load_vocoder("vocos", state_dict=synthetic.make_vocos_state_dict(seed=11))

    def step():
        mel = model.synthesise(x, x_len, n_timesteps=N_STEPS_ODE, speaker=0)["mel"]
        if vocoder is not None:
            vocoder(mel)                     # [B, 256 * T_valid] samples at 24 kHz; the metric stays mel frames
        if world > 1:
            mel = dp.all_gather_mels(mel, world)
        return mel

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mel = step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    valid_per_utt = mel.shape[-1]
    frames = valid_per_utt * BATCH * world * args.steps
    t_pad = 2 * valid_per_utt

    terms = model.hip.gemm_terms()
    roofline = None
    if rank == 0 and not args.no_events:
        # same steps again with a HIP event pair around every kernel launch, recorded on the launch stream
        hip = model.hip
        hip.prof_enable(True)
        hip.prof_reset()
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for _ in range(args.steps):
            model.synthesise(x, x_len, n_timesteps=N_STEPS_ODE, speaker=0)
        torch.cuda.synchronize()
        p_el = time.perf_counter() - p0
        n_g, ms_g, fl_g, by_g = hip.prof_read(0)
        n_a, ms_a, fl_a, by_a = hip.prof_read(1)
        n_e, ms_e, _, _ = hip.prof_read(2)
        hip.prof_enable(False)
        hip.prof_reset()
        achieved = fl_g / (ms_g * 1e-3) / 1e12
        _, instr, products, hw_peak = GEMM_MODES[terms]
        peak = hw_peak / products      # fp32-equivalent peak: every algorithmic MAC costs `products` MFMA MACs
        roofline = {
            "bound": "mfma", "kernel": "gemm_p16_kernel + tblock_chain_kernel + gemm_f32_kernel (GEMM / implicit conv1d / transformer-block chain, all instantiations)",
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            # the whole step against the same peak: every executed GEMM + attention FLOP over the TIMED step (launch gaps, attention,
            # streaming kernels, the text encoder and the host included), not only the time inside GEMM kernels
            "frac_whole_step": round((fl_g + fl_a) / args.steps / (el / args.steps) / 1e12 / peak, 4),
            "traffic": pmc_traffic()[0], "traffic_source": pmc_traffic()[1],
            "peak_basis": f"{hw_peak:.0f} TFLOP/s dense ({instr}) / {products}",
            "mfma_executed_tflops": round(achieved * products, 1),
            "algorithmic_mb_per_launch": round(by_g / max(n_g, 1) / 1e6, 2),
            "launches_per_step": n_g // args.steps, "avg_launch_us": round(ms_g * 1e3 / max(n_g, 1), 2),
            "gflop_per_launch": round(fl_g / max(n_g, 1) / 1e9, 3),
            "gemm_ms_per_step": round(ms_g / args.steps, 3),
            "attention": {"tflops": round(fl_a / (ms_a * 1e-3) / 1e12, 2), "ms_per_step": round(ms_a / args.steps, 3),
                          "launches_per_step": n_a // args.steps},
            "elementwise_ms_per_step": round(ms_e / args.steps, 3), "elementwise_launches_per_step": n_e // args.steps,
            # two FLOP counts: what the kernels executed (folded padding: valid rows + one row for all padded frames, rounded up to
            # whole tiles) -- `achieved` / `frac` above come from this one -- and SURVEY 8d's strict-padding count of the reference
            "executed_tflop_per_step": round((fl_g + fl_a) / args.steps / 1e12, 3),
            "strict_padding_tflop_per_step": round(algorithmic_flops(hp, BATCH, t_pad, N_TOKENS, N_STEPS_ODE * {'euler': 1, 'midpoint': 2, 'rk4': 4}[SOLVER]) / 1e12, 3),
            "rows_per_utterance": {"reference_T_pad": t_pad, "held": model.decoder.fold_plan(t_pad, valid_per_utt) or t_pad},
            "whole_path_tflops_strict_padding_equivalent": round(algorithmic_flops(hp, BATCH, t_pad, N_TOKENS, N_STEPS_ODE * {'euler': 1, 'midpoint': 2, 'rk4': 4}[SOLVER]) * args.steps / el / 1e12, 2),
            "ms_per_step_with_events": round(p_el / args.steps * 1e3, 2),
        }

    precision = None
    if rank == 0 and terms in (1, 16, 17):
        # reduced-precision arithmetic: its mel error against this library's fp32-equivalent path (itself 4e-5 from the
        # reference goldens, tests/test_hip_path.py) on the first utterances of the batch, same ids and the same noise
        nb = min(4, BATCH)
        z = model.decoder.noise(torch.empty(nb, hp.n_feats, t_pad, device=dev))
        got = model.synthesise(x[:nb], x_len[:nb], n_timesteps=N_STEPS_ODE, speaker=0, z=z)["mel"]
        ref_model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
        ref_model.load_state_dict(sd, strict=True)
        ref_model = ref_model.to(dev).eval()
        ref_model._rt.hip = importlib.import_module(PKG + "._hip").HipModel(hp, terms=2)
        ref_model.decoder.solver = SOLVER
        want = ref_model.synthesise(x[:nb], x_len[:nb], n_timesteps=N_STEPS_ODE, speaker=0, z=z)["mel"]
        precision = {"mel_max_abs_error_vs_fp32_path": round(float((got - want).abs().max()), 5),
                     "mel_mean_abs_error_vs_fp32_path": round(float((got - want).abs().mean()), 6),
                     "mel_abs_max": round(float(want.abs().max()), 2), "utterances_compared": nb,
                     "note": "outside the 1e-3 bar of the fp32 path by design (BASELINE.md section 4: reported, not gated)"}
        del ref_model

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(hp, sd, synthetic)

    if rank == 0:
        line = {
            "metric": "mel-frames/s (100-bin, 24 kHz) end-to-end synthesise(), n_timesteps=10",
            "value": round(frames / el, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": GEMM_MODES[terms][0], "data": "synthetic",
            "config": {"workload": ("" if default_cfg else f"NON-DEFAULT batch={BATCH} {SOLVER}/{N_STEPS_ODE}{' + Vocos head' if args.with_vocoder else ''} variant of ") +
                                   "configs[1]: batch=32 random phoneme seqs len=128, n_spks=1, euler n_timesteps=10, fp32, "
                                   "prod v20 architecture, random-init weights, T_pad=640 / 320 valid frames per utterance "
                                   "(reference 2x padding kept in the results; the estimator folds the identical padded frames, DESIGN.md section 4), "
                                   "noise from the device seed-42 generator",
                       "per_gpu_batch": BATCH, "global_batch": BATCH * world, "n_tokens": N_TOKENS, "parallelism": f"dp{world}" + ("" if backend == "nccl" or world == 1 else f" REHEARSAL over {backend} on {n_dev} device(s)"),
                       "n_feats": hp.n_feats, "note": "the reference fork uses 100 mel bins (Vocos-24k), not 80"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if terms in (16, 17):
            line["config"]["workload"] = ("per-rank shape of configs[2] (batch=256 sharded 8-way = 32 utterances per GPU, n_timesteps=10, 16-bit "
                                          "storage / fp32 accumulate) -- NOT the fp32 headline; " + line["config"]["workload"])
        if precision is not None:
            line["precision"] = precision
        print(json.dumps(line))
    if world > 1:
        dist.barrier()              # the other ranks wait here while rank 0 finishes its per-kernel event pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
