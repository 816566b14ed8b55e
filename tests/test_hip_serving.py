"""BASELINE configs #4 and #5 exercised end to end on the GPU.

#4: multi-speaker (n_spks=10) + per-utterance speaker conditioning, n_timesteps=32, mel -> Vocos-24k waveform on the device
    (reference inference.py:57-76,233-265 with the reference's vocos head).  The mel is checked against the oracle at a size the
    oracle affords, the full-size batch through size-independent properties, the waveform against the (parity unpinned, see
    oracle/vocos_oracle.py) restated head.
#5: the closed-loop load shape of reference psr/load_test.py over the dynamic batcher (tools/load_sim.py), short run."""
import importlib.util
import sys

import pytest
import torch

from conftest import ROOT, sub

pytestmark = pytest.mark.gpu
MEL_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    return torch.device("cuda")


@pytest.fixture(scope="module")
def cfg4(hparams, synthetic, dev):
    inf = sub("inference")
    hp = hparams.prod_v20(n_spks=10)
    sd = synthetic.make_state_dict(hp, seed=7)
    m = inf.MatchaTTSInfer(**hp.as_reference_kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    voc_sd = synthetic.make_vocos_state_dict(seed=11)
    vocoder = inf.load_vocoder("vocos", state_dict=voc_sd)
    return hp, sd, m, vocoder, voc_sd, inf


def maxabs(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def test_config4_small_vs_oracle(cfg4, synthetic, oracle, dev):
    """n_spks=10, a different speaker per utterance, euler/32 and midpoint/16 (32 evaluations each), ragged Tx <= 24: mel vs
    the oracle; then the chained Vocos head vs its restatement."""
    hp, sd, model, vocoder, voc_sd, inf = cfg4
    lengths = [24, 17, 9]
    x, x_len, _ = synthetic.make_inputs(hp, 3, 24, seed=404, lengths=lengths)
    spk = torch.tensor([7, 2, 9])
    for solver, steps in (("euler", 32), ("midpoint", 16)):
        model.decoder.solver = solver
        with torch.inference_mode():
            ref = oracle.synthesise(sd, hp, x, x_len, steps, speaker=spk, solver=solver)
        z = synthetic.cpu_noise((3, hp.n_feats, ref["t_pad"])).to(dev)
        out = model.synthesise(x.to(dev), x_len.to(dev), steps, speaker=spk.to(dev), z=z)
        assert torch.equal(out["mel_lengths"].cpu(), ref["mel_lengths"])
        assert maxabs(out["mel"], ref["mel"]) < MEL_TOL, solver
    import vocos_oracle
    b, t = 1, int(out["mel_lengths"][1])
    mel1 = out["mel"][b:b + 1, :, :t]
    wav = inf.to_waveform(mel1, vocoder)
    with torch.inference_mode():
        ref_audio = vocos_oracle.decode(voc_sd, mel1.cpu())
        peak = ref_audio.abs().max()
        if peak > 1.0:
            ref_audio = ref_audio / peak * 0.95
    assert wav.shape == ref_audio.squeeze().shape
    assert maxabs(wav, ref_audio.squeeze()) < 2e-3 * max(1.0, float(ref_audio.abs().max()))


def test_config4_full_size_properties(cfg4, synthetic, dev):
    """B=32, Tx=128, speakers b mod 10, euler/32 -> mel -> waveform on the device: finite, deterministic,
    rows independent of their batch neighbours (row b == the utterance alone, same speaker and noise slice), the speaker
    matters, and the waveform has hop * (T - 1) samples per utterance."""
    hp, sd, model, vocoder, voc_sd, inf = cfg4
    x, x_len, spk = synthetic.make_inputs(hp, 32, 128, seed=1234)
    assert spk.tolist() == [b % 10 for b in range(32)]
    z = synthetic.cpu_noise((32, 100, 640)).to(dev)
    model.decoder.solver = "euler"
    out = model.synthesise(x.to(dev), x_len.to(dev), 32, speaker=spk.to(dev), z=z)["mel"]
    assert out.shape == (32, 100, 320) and torch.isfinite(out).all()
    assert torch.equal(out, model.synthesise(x.to(dev), x_len.to(dev), 32, speaker=spk.to(dev), z=z)["mel"])
    for b in (3, 29):
        solo = model.synthesise(x[b:b + 1].to(dev), x_len[b:b + 1].to(dev), 32, speaker=spk[b:b + 1].to(dev), z=z[b:b + 1])["mel"]
        assert maxabs(out[b:b + 1], solo) < 2e-4
    other = model.synthesise(x[3:4].to(dev), x_len[3:4].to(dev), 32, speaker=torch.tensor([4], device=dev), z=z[3:4])["mel"]
    assert maxabs(out[3:4], other) > 1e-2                          # speaker 3 vs speaker 4 on the same text and noise
    audio = vocoder(out)
    assert audio.shape == (32, 256 * 319) and torch.isfinite(audio).all()
    wav = inf.to_waveform(out[:1], vocoder)
    assert wav.device.type == "cpu" and wav.dim() == 1 and float(wav.abs().max()) <= 1.0 + 1e-6


def test_per_utterance_scale_factors_equal_separate_calls(cfg4, synthetic, dev):
    """A batch that mixes voices carries one scale_correction / length_scale per utterance (mtts_durations_per_utterance): the
    integer durations equal those of separate calls with the scalar factors (reference inference.py:129-143)."""
    hp, sd, model, vocoder, voc_sd, inf = cfg4
    lengths = [40, 33, 21]
    x, x_len, _ = synthetic.make_inputs(hp, 3, 40, seed=5, lengths=lengths)
    sc, ls = [1.08, 1.03, 1.05], [1.0, 0.8, 1.6]
    model.decoder.solver = "euler"
    out = model.synthesise(x.to(dev), x_len.to(dev), 1, speaker=0, scale_correction=sc, length_scale=ls, debug=True)
    for b, n in enumerate(lengths):
        solo = model.synthesise(x[b:b + 1, :n].to(dev), x_len[b:b + 1].to(dev), 1, speaker=0, scale_correction=sc[b],
                                length_scale=ls[b], debug=True)
        assert torch.equal(out["phoneme_durations"][b, :n], solo["phoneme_durations"][0])
        assert int(out["mel_lengths"][b]) == int(solo["mel_lengths"][0])


def test_pipeline_signature_and_outputs(cfg4, synthetic, dev, monkeypatch):
    """`pipeline(model, vocoder, text, speaker, voice_mix, n_timesteps, scale_correction, length_scale, debug)` as the
    reference's CLI / server call it (reference inference.py:233-257, cli.py, server.py:116), with the phonemizer (a CPU front
    end outside the path) replaced by a stub: a trimmed 1-D host waveform; with debug=True the (waveform, encoder waveform,
    [(phone, raw duration, duration)]) triple."""
    hp, sd, model, vocoder, voc_sd, inf = cfg4
    ids = synthetic.make_inputs(hp, 1, 20, seed=3)[0][0].tolist()

    def fake_process_text(text, language):
        assert language == "en-us"
        x = torch.tensor(ids, dtype=torch.long, device=dev)[None]
        return {"x_orig": text, "x": x, "x_lengths": torch.tensor([len(ids)], device=dev), "x_phones": "a" * len(ids), "x_phone_ids": ids}

    monkeypatch.setattr(inf, "process_text", fake_process_text)
    model.decoder.solver = inf.DEFAULT_ODE_SOLVER
    wav = inf.pipeline(model, vocoder, "hello", speaker=3, n_timesteps=2, scale_correction=1.03, length_scale=0.9)
    assert wav.device.type == "cpu" and wav.dim() == 1 and wav.numel() > 0 and float(wav.abs().max()) <= 1.0 + 1e-6
    out = model.synthesise(torch.tensor([ids], device=dev), torch.tensor([len(ids)], device=dev), 2, speaker=3,
                           scale_correction=1.03, length_scale=0.9)
    full = inf.to_waveform(out["mel"], vocoder)
    assert wav.numel() <= full.numel() and (full.numel() - wav.numel()) % 240 == 0           # whole 10 ms windows trimmed
    assert torch.allclose(wav, full[: wav.numel()], atol=1e-6)
    w2, enc_wav, pairs = inf.pipeline(model, vocoder, "hello", voice_mix=[(0, 0.5), (1, 0.5)], n_timesteps=2, debug=True)
    assert w2.dim() == 1 and enc_wav.dim() == 1 and len(pairs) == len(ids) and all(len(p) == 3 for p in pairs)


def test_batcher_mixes_plain_voices_and_voice_mixes(cfg4, synthetic, dev):
    """One batch carrying a plain voice, a two-voice mix (reference server.py:96-101, inference.py:57-76) and another plain
    voice, each with its own scale correction and speed: every request gets the mel of its own `synthesise` call."""
    hp, sd, model, vocoder, voc_sd, inf = cfg4
    bt, sv = sub("batcher"), sub("serving")
    reqs = [dict(voice=3, speed=1.0), dict(voice="2(70)+6(30)", speed=1.25), dict(voice=9, speed=0.8)]
    lengths = [30, 22, 41]
    ids = [synthetic.make_inputs(hp, 1, n, seed=70 + i)[0][0].tolist() for i, n in enumerate(lengths)]
    model.decoder.solver = "midpoint"
    with bt.FrameBudgetBatcher(model, max_batch=8, max_tokens=4096, max_wait_ms=50.0) as q:
        futs = []
        for r, tok in zip(reqs, ids):
            p = sv.request_params(**r)
            futs.append((p, q.submit(tok, speaker=p.speaker, voice_mix=p.voice_mix, solver="midpoint", n_timesteps=2,
                                     scale_correction=p.scale_correction, length_scale=p.length_scale)))
        results = [(p, f.result(timeout=120)) for p, f in futs]
        assert q.batches_run == 1
    for (p, res), tok in zip(results, ids):
        x = torch.tensor([tok], device=dev)
        x_len = torch.tensor([len(tok)], device=dev)
        solo = model.synthesise(x, x_len, 2, speaker=p.speaker, voice_mix=p.voice_mix, scale_correction=p.scale_correction,
                                length_scale=p.length_scale)
        assert res["mel_length"] == int(solo["mel_lengths"][0])
        assert maxabs(res["mel"][None], solo["mel"][:, :, :res["mel_length"]]) < 5e-5


def test_config5_closed_loop_short_run(dev):
    """tools/load_sim.py (closed-loop users over FrameBudgetBatcher, Vocos + trim per request) for a few seconds at 1 and 6
    users: every request is answered, latency per audio-second is finite and positive, batches form under concurrency."""
    spec = importlib.util.spec_from_file_location("load_sim", ROOT / "tools" / "load_sim.py")
    ls = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ls)
    assert len(ls.TEXT_SAMPLE_CHARS) == 33 and min(c for _, c in ls.TEXT_SAMPLE_CHARS) == 26 and max(c for _, c in ls.TEXT_SAMPLE_CHARS) == 434
    hp, inference, model, batcher = ls.build(dev, with_vocoder=True, max_batch=8)
    try:
        batcher.submit([1] * 100, solver="midpoint", n_timesteps=4).result()
        one = ls.run_level(batcher, inference, hp, 1, 2.0, 3, time_scale=0.05, min_requests=3, max_seconds=6.0, warm_seconds=1.0)
        six = ls.run_level(batcher, inference, hp, 6, 3.0, 4, time_scale=0.02, min_requests=12, max_seconds=8.0, warm_seconds=1.0)
    finally:
        batcher.close()
    for r in (one, six):
        assert r["requests"] >= r["users"] and r["p50_latency_per_audio_s"] > 0 and r["p95_latency_s"] < 5.0
    assert six["mean_batch"] > 1.0 and 0.0 < six["worker_busy_fraction"] <= 1.0 and six["equivalent_users"] == 300
    assert model.hip.workspace_bytes_held() < 8 << 30          # grow-only scratch: bounded by the largest batch seen
