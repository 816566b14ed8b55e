#!/usr/bin/env python3
"""BASELINE config #5: the reference's load test (reference psr/load_test.py:28-62,87-120) as an in-process closed loop.

The reference drives its HTTP server with locust users that POST one of 33 texts, wait for the audio, then sleep for the
audio's duration (a listener) before the next request.  Here the same closed loop runs in one process against
``FrameBudgetBatcher`` (dynamic batching in front of ``MatchaTTSInfer.synthesise``) with the reference handler's work per
request (reference server.py:93-127): synthesise (midpoint, 4 steps = the server defaults) -> Vocos head -> peak normalise ->
trailing-silence trim -> waveform on the host.  The phonemizer and the MP3/OGG encoders are CPU stages outside the path
(SURVEY.md section 8): a request is a random id sequence of round(2.3 * characters) tokens (3 ids per
voiced phoneme; the phonemizer is not runnable offline), characters = the lengths of the reference's 33 TEXT_SAMPLES.

    python tools/load_sim.py [--levels 8:0.1,32:0.1,128:1,512:0.25,...] [--min-requests 200] [--no-vocoder] [--max-batch 32]
A level is users:time_scale -- `users` closed-loop threads whose listening pause is the audio's duration x time_scale, i.e. the
request rate of users / time_scale real listeners (a process cannot hold thousands of threads; below saturation the latency
does not depend on how the rate is produced).  Per level: a warm-up that is NOT measured (the first seconds: HIP-graph
captures, workspace growth, lazily configured kernels), then at least --min-requests measured requests.  One JSON line per
level (requests, p50 / p95 latency, p50 / p95 latency per audio-second = the real-time factor a client sees, audio-seconds
served per second, mean batch, worker-busy fraction) and a final line with the saturation point: the largest equivalent user
count whose p95 latency per audio-second stays under --bound (the reference's one published figure for this shape is "10
users" on an i9 + RTX 3050, reference psr/PSR_README.md:22).
"""
import argparse
import importlib
import json
import random
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "matcha-tts-24k_amd"

# (language, characters) of the reference's TEXT_SAMPLES (psr/load_test.py:28-62), in file order
TEXT_SAMPLE_CHARS = [("en-us", 430), ("en-us", 236), ("en-us", 162), ("en-us", 434), ("en-us", 142), ("en-us", 91), ("en-us", 428),
                     ("en-us", 26), ("en-us", 152), ("en-us", 330), ("en-us", 131), ("ro", 356), ("fr-fr", 388), ("en-us", 61),
                     ("en-us", 58), ("en-us", 57), ("en-us", 64), ("en-us", 78), ("en-us", 192), ("en-us", 184), ("en-gb", 75),
                     ("en-gb", 69), ("en-gb", 159), ("ro", 78), ("ro", 52), ("ro", 183), ("ro", 84), ("ro", 77), ("fr-fr", 60),
                     ("fr-fr", 63), ("fr-fr", 179), ("fr-fr", 86), ("fr-fr", 84)]
TOKENS_PER_CHAR = 2.3
MAX_TOKENS = 3000


def percentile(v, q):
    if not v:
        return None
    s = sorted(v)
    return s[min(len(s) - 1, int(round(q * (len(s) - 1))))]


def run_level(batcher, inference, hp, users, min_seconds, seed, time_scale=1.0, min_requests=200, max_seconds=60.0, warm_seconds=3.0):
    t_begin = time.monotonic()
    state = {"measure_from": t_begin + warm_seconds, "stop": False}
    lat, lat_per_s, audio_s, lock = [], [], [], threading.Lock()
    serving = importlib.import_module(PKG + ".serving")
    voices_of = {"en-us": [{"id": "0(50)+1(50)"}]}            # the load test's extra "Voice mix" voice (load_test.py:17)
    for v in inference.VOICES:
        voices_of.setdefault(v["lang"], []).append(v)

    def user(uid):
        rng = random.Random(seed * 1000 + uid)
        time.sleep(rng.random() * min(1.0, 0.002 * users + 0.2))      # locust spawns users over a ramp, not in one instant
        while not state["stop"]:
            lang, chars = rng.choice(TEXT_SAMPLE_CHARS)
            voice = rng.choice(voices_of.get(lang) or inference.VOICES)
            n_tok = max(1, min(MAX_TOKENS, round(chars * TOKENS_PER_CHAR)))
            ids = [rng.randrange(hp.n_vocab) for _ in range(n_tok)]
            t0 = time.monotonic()
            p = serving.request_params(voice=voice["id"], speed=1.0)            # the handler's mapping (server.py:96-115)
            res = batcher.submit(ids, speaker=p.speaker, voice_mix=p.voice_mix, solver=p.solver, n_timesteps=p.n_timesteps,
                                 scale_correction=p.scale_correction, length_scale=p.length_scale).result()
            t1 = time.monotonic()
            dur = (len(res["audio"]) if "audio" in res else res["mel_length"] * inference.STD_RES_HOP_LENGTH) / inference.SAMPLE_RATE
            if t0 >= state["measure_from"]:                  # requests that started inside the measured window
                with lock:
                    lat.append(t1 - t0)
                    audio_s.append(dur)
                    lat_per_s.append((t1 - t0) / max(dur, 1e-3))
            end = time.monotonic() + dur * time_scale        # the listener (reference load_test.py:120 gevent.sleep(duration))
            while not state["stop"] and time.monotonic() < end:
                time.sleep(min(0.05, max(end - time.monotonic(), 0.0)))

    threads = [threading.Thread(target=user, args=(u,), daemon=True) for u in range(users)]
    for t in threads:
        t.start()
    time.sleep(warm_seconds)
    b0, busy0, t0 = batcher.batches_run, batcher.busy_s, time.monotonic()
    while True:
        time.sleep(0.25)
        el = time.monotonic() - t0
        with lock:
            n = len(lat)
        if (el >= min_seconds and n >= min_requests) or el >= max_seconds:
            break
    state["stop"] = True
    el = time.monotonic() - t0
    nb, busy = max(batcher.batches_run - b0, 1), batcher.busy_s - busy0
    for t in threads:
        t.join(timeout=30)
    with lock:
        n = len(lat)
        return {"users": users, "time_scale": time_scale, "equivalent_users": round(users / time_scale), "seconds": round(el, 1),
                "requests": n, "requests_per_s": round(n / el, 2), "mean_batch": round(n / nb, 2),
                "p50_latency_s": percentile(lat, 0.5), "p95_latency_s": percentile(lat, 0.95),
                "p50_latency_per_audio_s": percentile(lat_per_s, 0.5), "p95_latency_per_audio_s": percentile(lat_per_s, 0.95),
                "audio_s_per_s": round(sum(audio_s) / el, 1), "mean_audio_s": round(sum(audio_s) / max(n, 1), 2),
                "worker_busy_fraction": round(min(busy / el, 1.0), 3)}


def build(dev, with_vocoder=True, n_spks=15, max_batch=32, max_tokens=16384, max_wait_ms=2.0):
    hparams = importlib.import_module(PKG + ".hparams")
    synthetic = importlib.import_module(PKG + ".synthetic")
    inference = importlib.import_module(PKG + ".inference")
    batcher_mod = importlib.import_module(PKG + ".batcher")
    hp = hparams.prod_v20(n_spks=n_spks)                     # 15 voices (reference inference.py:16-32)
    model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(synthetic.make_state_dict(hp, seed=7), strict=True)
    model = model.to(dev).eval()
    vocoder = inference.load_vocoder("vocos", state_dict=synthetic.make_vocos_state_dict(seed=11)) if with_vocoder else None
    b = batcher_mod.FrameBudgetBatcher(model, max_batch=max_batch, max_tokens=max_tokens, max_wait_ms=max_wait_ms, vocoder=vocoder)
    return hp, inference, model, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="8:0.1,32:0.1,128:0.25,256:0.25,512:0.25,512:0.125,512:0.0625,512:0.03125")
    ap.add_argument("--seconds", type=float, default=10.0, help="minimum measured window per level")
    ap.add_argument("--max-seconds", type=float, default=45.0)
    ap.add_argument("--min-requests", type=int, default=200)
    ap.add_argument("--bound", type=float, default=0.25, help="p95 latency per audio-second that still counts as served")
    ap.add_argument("--no-vocoder", action="store_true")
    ap.add_argument("--max-batch", type=int, default=32)
    ap.add_argument("--max-wait-ms", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda")
    hp, inference, model, batcher = build(dev, not args.no_vocoder, max_batch=args.max_batch, max_wait_ms=args.max_wait_ms)
    rows = []
    try:
        # warm-up outside every window: first-use costs (workspace growth, lazily configured kernels, graph captures of the common
        # single-request buckets) are not request latency
        for n in (60, 130, 210, 400, 560, 1000):
            batcher.submit([1] * n, solver=inference.DEFAULT_ODE_SOLVER, n_timesteps=inference.DEFAULT_NUM_STEPS).result()
        for lv in args.levels.split(","):
            u, ts = lv.split(":")
            r = run_level(batcher, inference, hp, int(u), args.seconds, args.seed, time_scale=float(ts), min_requests=args.min_requests,
                          max_seconds=args.max_seconds)
            r.update(config="configs[4]: closed-loop users (reference psr/load_test.py), 33 TEXT_SAMPLES lengths x 2.3 tokens/char, "
                            f"midpoint/4, dynamic batching max_batch={args.max_batch}, per-request padding, "
                            + ("Vocos head + trim on device" if not args.no_vocoder else "mel only"),
                     data="synthetic ids, random-init weights", n_gpus=1)
            rows.append(r)
            print(json.dumps(r), flush=True)
        ok = [r for r in rows if r["p95_latency_per_audio_s"] is not None and r["p95_latency_per_audio_s"] <= args.bound]
        best = max(ok, key=lambda r: r["equivalent_users"]) if ok else None
        print(json.dumps({"saturation": {"bound_p95_latency_per_audio_s": args.bound,
                                         "max_equivalent_users_within_bound": best["equivalent_users"] if best else 0,
                                         "at": {k: best[k] for k in ("users", "time_scale", "requests_per_s", "audio_s_per_s", "mean_batch",
                                                                      "p95_latency_per_audio_s", "worker_busy_fraction")} if best else None,
                                         "reference_published": "10 users on i9 + RTX 3050 (reference psr/PSR_README.md:22)"}}), flush=True)
    finally:
        batcher.close()


if __name__ == "__main__":
    main()
