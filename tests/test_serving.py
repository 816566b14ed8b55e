"""Handler-level adapter: the request -> synthesis-parameter mapping of reference matcha/server.py:64-69,96-115 and the async
service over the batcher (stub model on the CPU; the GPU path is covered by tests/test_hip_serving.py)."""
import asyncio

import pytest

from conftest import sub


def test_request_params_follow_the_reference_handler():
    sv, inf = sub("serving"), sub("inference")
    p = sv.request_params(voice=4, speed=1.0)
    assert (p.speaker, p.voice_mix, p.language) == (4, None, "en-gb")
    assert p.scale_correction == inf.VOICES[4]["scale_correction"] == 1.08 and p.length_scale == 1.0
    assert (p.n_timesteps, p.solver) == (inf.DEFAULT_NUM_STEPS, inf.DEFAULT_ODE_SOLVER)
    p = sv.request_params(voice="2(70)+6(30)", speed=2.0, steps=8, solver="euler")
    assert p.voice_mix == [(2, 0.7), (6, 0.3)] and p.speaker == 0 and p.language == inf.VOICES[2]["lang"]
    assert p.scale_correction == pytest.approx(1.05 * 0.7 + 1.05 * 0.3) and p.length_scale == 0.5 and p.n_timesteps == 8
    assert sv.request_params(speed=0.1).length_scale == 2.0 and sv.request_params(speed=100.0).length_scale == 0.1    # clamps
    assert sv.request_params(voice="7").language == "ro"
    for bad in ("2(70)", "2(70)+x(30)", "1(50)+2(25)+3(25)"):
        with pytest.raises(ValueError):
            sv.parse_voice_mix(bad)


def test_speech_service_submits_what_the_handler_would():
    sv, bt = sub("serving"), sub("batcher")
    seen = []

    def run(batch):
        seen.extend(batch)
        return [{"mel_length": len(r.ids), "audio": f"audio-{r.speaker}-{len(r.ids)}"} for r in batch]

    with bt.FrameBudgetBatcher(model=None, max_batch=4, max_tokens=4096, max_wait_ms=5.0, run_batch=run) as q:
        svc = sv.SpeechService(q, phonemize=lambda text, lang: [1 + (ord(c) % 50) for c in text] + [len(lang)])

        async def main():
            return await asyncio.gather(svc.speak("hello world", voice=3), svc.speak("bonjour", voice="8(60)+9(40)", speed=1.25),
                                        svc.speak("  padded  ", voice=13, steps=6, solver="euler"))

        out = asyncio.run(main())
        with pytest.raises(ValueError):
            svc.submit("x" * 1001)
    assert out[0] == "audio-3-12" and out[1].startswith("audio-0-") and out[2].startswith("audio-13-")
    by_len = {len(r.ids): r for r in seen}
    r = by_len[len("bonjour") + 1]
    assert r.voice_mix == [(8, 0.6), (9, 0.4)] and r.length_scale == pytest.approx(0.8) and r.scale_correction == pytest.approx(1.05 * 0.6 + 1.03 * 0.4)
    r = by_len[len("padded") + 1]                      # the handler strips the text (reference server.py:116)
    assert (r.solver, r.n_timesteps, r.scale_correction) == ("euler", 6, 1.07)
