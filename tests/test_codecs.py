"""The post-waveform codec entry points keep the reference's contract (reference matcha/inference.py:290-322): MP3 through the
reference installation's encode_mp3 with vbr_quality=5 / algorithm_quality=5, Ogg/Opus through PyAV with libopus, mono, 48 kbit/s,
compression_level 5.  Neither library exists in the build image, so both are stood in for by recording fakes: what is checked
is the call sequence and the parameters, and that a missing PyAV raises a clear ImportError instead of a silent fallback."""
import sys
import types

import numpy as np
import pytest
import torch

from conftest import sub


def test_convert_to_mp3_passes_the_reference_parameters(monkeypatch):
    inf = sub("inference")
    calls = {}

    def encode_mp3(pcm, sample_rate, vbr_quality, algorithm_quality):
        calls.update(pcm=pcm, sample_rate=sample_rate, vbr_quality=vbr_quality, algorithm_quality=algorithm_quality)
        return b"\xff\xfb" + bytes(10)

    for name in ("matcha", "matcha.utils"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    mod = types.ModuleType("matcha.utils.mp3_converter")
    mod.encode_mp3 = encode_mp3
    monkeypatch.setitem(sys.modules, "matcha.utils.mp3_converter", mod)
    wav = torch.tensor([0.0, 0.5, -1.0, 1.0])
    out = inf.convert_to_mp3(wav)
    assert out.startswith(b"\xff\xfb")
    assert calls["sample_rate"] == inf.SAMPLE_RATE == 24000 and calls["vbr_quality"] == 5 and calls["algorithm_quality"] == 5
    assert calls["pcm"].dtype == np.int16 and calls["pcm"].tolist() == [0, 16383, -32767, 32767]


def test_convert_to_opus_ogg_follows_the_reference_call_sequence(monkeypatch):
    inf = sub("inference")
    log = []

    class Stream:
        def encode(self, frame=None):
            log.append(("encode", None if frame is None else frame.sample_rate))
            return [b"pkt"] if frame is not None else [b"flush"]

    class Container:
        def __init__(self, buf):
            self.buf = buf

        def add_stream(self, codec, rate):
            log.append(("add_stream", codec, rate))
            self.stream = Stream()
            return self.stream

        def mux(self, packet):
            self.buf.write(b"OggS" + packet)

        def close(self):
            log.append(("close", self.stream.layout, self.stream.bit_rate, dict(self.stream.options)))

    class AudioFrame:
        @staticmethod
        def from_ndarray(a, format, layout):
            log.append(("frame", a.shape, str(a.dtype), format, layout))
            return AudioFrame()

    av = types.ModuleType("av")
    av.open = lambda buf, mode, format: (log.append(("open", mode, format)), Container(buf))[1]
    av.AudioFrame = AudioFrame
    monkeypatch.setitem(sys.modules, "av", av)
    out = inf.convert_to_opus_ogg(torch.linspace(-1, 1, 480))
    assert out == b"OggSpktOggSflush"
    assert log[0] == ("open", "w", "ogg") and log[1] == ("add_stream", "libopus", 24000)
    assert log[2] == ("frame", (1, 480), "int16", "s16", "mono")
    assert log[3] == ("encode", 24000) and log[4] == ("encode", None)
    assert log[5] == ("close", "mono", 48000, {"compression_level": "5"})


def test_convert_to_opus_ogg_without_pyav_says_so(monkeypatch):
    inf = sub("inference")
    monkeypatch.setitem(sys.modules, "av", None)           # import av -> ImportError
    with pytest.raises(ImportError, match="PyAV"):
        inf.convert_to_opus_ogg(torch.zeros(10))
