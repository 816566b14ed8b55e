"""Handler-level adapter for the reference's HTTP server.

The reference handler (reference matcha/server.py:93-127) turns one POST /v1/audio/speech body into
``pipeline(model, vocoder, text, speaker, voice_mix, steps, scale_correction, length_scale)`` and runs it on the event-loop
thread, one request at a time.  ``request_params`` is that mapping (voice / voice-mix parsing, the per-voice duration scale
correction of ``VOICES``, the speed -> length_scale clamp) as a pure function, and ``SpeechService`` is the piece a maintainer
puts behind the same route: it phonemizes, submits to a ``FrameBudgetBatcher`` (so concurrent requests share estimator
launches without changing anybody's audio) and awaits the trimmed waveform.  Transport, response encoding (MP3 / OGG) and the
phonemizer stay the reference's (out of the path's scope).
"""
from __future__ import annotations

import asyncio
import re
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

from .inference import DEFAULT_NUM_STEPS, DEFAULT_ODE_SOLVER, VOICES

LENGTH_SCALE_MIN = 0.1      # fastest (client speed 2.0 clamps here; reference server.py:34-36)
LENGTH_SCALE_MAX = 2.0      # slowest
MAX_TEXT_LENGTH = 1000      # reference server.py:30

_MIX = re.compile(r"^\s*(\d+)\((\d+)\)\s*$")


def parse_voice_mix(voice: str) -> List[Tuple[int, float]]:
    """'2(70)+6(30)' -> [(2, 0.7), (6, 0.3)] (reference server.py:64-69)."""
    parts = voice.split("+")
    if len(parts) != 2:
        raise ValueError("a voice mix names exactly two voices: 'id(weight)+id(weight)'")
    out = []
    for p in parts:
        m = _MIX.match(p)
        if not m:
            raise ValueError(f"malformed voice mix term {p!r}")
        out.append((int(m.group(1)), int(m.group(2)) / 100))
    return out


@dataclass
class SpeechParams:
    speaker: int
    voice_mix: Optional[List[Tuple[int, float]]]
    language: str
    scale_correction: float
    length_scale: float
    n_timesteps: int
    solver: str


def request_params(voice=0, speed: float = 1.0, steps: int = DEFAULT_NUM_STEPS, solver: str = DEFAULT_ODE_SOLVER) -> SpeechParams:
    """The request -> synthesis parameters of reference server.py:96-115 (and the language lookup of inference.py:235-236)."""
    if "+" in str(voice):
        mix = parse_voice_mix(str(voice))
        speaker, primary = 0, mix[0][0]
        scale_correction = sum(VOICES[i]["scale_correction"] * w for i, w in mix)
    else:
        mix, speaker = None, int(voice)
        primary = speaker
        scale_correction = VOICES[speaker]["scale_correction"]
    language = next(v["lang"] for v in VOICES if v["id"] == str(primary))
    length_scale = max(LENGTH_SCALE_MIN, min(LENGTH_SCALE_MAX, 1.0 / speed))
    return SpeechParams(speaker, mix, language, scale_correction, length_scale, int(steps), solver)


class SpeechService:
    """``await service.speak(text, voice, speed, steps, solver)`` -> 1-D waveform tensor on the host.

    ``phonemize(text, language) -> list of phoneme ids`` is the reference's front end (``process_text``); ``batcher`` a
    ``FrameBudgetBatcher`` built with ``vocoder=`` so that results carry ``"audio"``."""

    def __init__(self, batcher, phonemize: Callable[[str, str], Sequence[int]], max_text_length: int = MAX_TEXT_LENGTH):
        self.batcher = batcher
        self.phonemize = phonemize
        self.max_text_length = int(max_text_length)

    def submit(self, text: str, voice=0, speed: float = 1.0, steps: int = DEFAULT_NUM_STEPS, solver: str = DEFAULT_ODE_SOLVER):
        if len(text) > self.max_text_length:
            raise ValueError(f"Text exceeds {self.max_text_length} characters")       # the handler's HTTP 400
        p = request_params(voice, speed, steps, solver)
        ids = self.phonemize(text.strip(), p.language)
        return self.batcher.submit(ids, speaker=p.speaker, voice_mix=p.voice_mix, solver=p.solver, n_timesteps=p.n_timesteps,
                                   scale_correction=p.scale_correction, length_scale=p.length_scale)

    async def speak(self, text: str, voice=0, speed: float = 1.0, steps: int = DEFAULT_NUM_STEPS, solver: str = DEFAULT_ODE_SOLVER):
        res = await asyncio.wrap_future(self.submit(text, voice, speed, steps, solver))
        return res["audio"] if "audio" in res else res["mel"]
