"""Drop-in for the reference's ``matcha.inference`` module (reference matcha/inference.py) on MI355X.

Same public names and call signatures -- ``load_matcha``, ``load_vocoder``, ``pipeline``, ``process_text``,
``to_waveform``, ``MatchaTTSInfer.synthesise``, ``VOICES`` and the constants -- so that the reference's
``matcha/cli.py`` and ``matcha/server.py`` keep working when their import is pointed here.
The arithmetic of ``synthesise`` runs in libmtts_hip.so; there is no CPU path (only the ``debug=True`` extras and
the two-term voice mix use a few PyTorch elementwise ops on device tensors, as the reference does).

Extensions (backwards compatible): ``x`` may hold B > 1 utterances and ``speaker`` may be a LongTensor[B]
(the reference builds a batch-1 speaker embedding and fails for B > 1, inference.py:118-121); ``synthesise``
accepts ``z=`` (explicit noise) for parity checks against the CPU reference stream.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .hparams import N_VOCAB, PathHParams, from_reference_kwargs
from .modules import Runtime, build_trees

# Voice table of the shipped model: id, language, per-speaker duration scale correction (reference inference.py:16-32).
VOICES = [
    {"id": str(i), "lang": lang, "gender": gender, "name": name, "scale_correction": sc}
    for i, (lang, gender, name, sc) in enumerate([
        ("en-us", "male", "Kai", 1.08), ("en-us", "female", "Jane", 1.05), ("en-us", "female", "Aria", 1.05),
        ("en-us", "female", "Bella", 1.03), ("en-gb", "male", "Brian", 1.08), ("en-gb", "male", "Arthur", 1.08),
        ("en-us", "female", "Nicole", 1.05), ("ro", "male", "Emil", 1.04), ("fr-fr", "female", "Denise", 1.05),
        ("fr-fr", "male", "Henri", 1.03), ("en-us", "male", "Matthew", 1.06), ("en-us", "male", "Lewis", 1.08),
        ("en-us", "male", "Michael", 1.03), ("it", "female", "Isabella", 1.07), ("it", "male", "Marcello", 1.07),
    ])
]

SAMPLE_RATE = 24000
STD_RES_HOP_LENGTH = 256
HIGH_RES_HOP_LENGTH = 128
DEFAULT_ODE_SOLVER = "midpoint"
DEFAULT_NUM_STEPS = 4
DEVICE = torch.device("cuda")


def fix_len_compatibility(length: int, num_downsamplings_in_unet: int = 1) -> int:
    """ceil(length / 2^n) * 2^n (reference utils/model.py:15-21)."""
    f = 2 ** num_downsamplings_in_unet
    return int(math.ceil(int(length) / f) * f)


class MatchaTTSInfer(nn.Module):
    """Inference model: speaker tables + text encoder + CFM decoder (reference inference.py:44-183)."""

    def __init__(self, n_spks, n_feats, encoder, decoder, cfm, data_statistics, spk_emb_dim, **_):
        super().__init__()
        hp = from_reference_kwargs(n_spks, n_feats, encoder, decoder, cfm, data_statistics, spk_emb_dim)
        self._init_from_hparams(hp)

    @classmethod
    def from_hparams(cls, hp: PathHParams) -> "MatchaTTSInfer":
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self._init_from_hparams(hp)
        return self

    def _init_from_hparams(self, hp: PathHParams) -> None:
        object.__setattr__(self, "hp", hp)
        object.__setattr__(self, "_rt", Runtime(hp, self))
        build_trees(hp, self, self._rt)
        with torch.no_grad():
            self.mel_mean.fill_(hp.mel_mean)
            self.mel_std.fill_(hp.mel_std)

    # ---- parameter changes invalidate the packed device image
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {k.replace("_orig_mod.", ""): v for k, v in state_dict.items() if "rope." not in k}
        out = super().load_state_dict(sd, strict=strict, assign=assign)
        self._rt.dirty = True
        return out

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._rt.dirty = True
        return r

    @property
    def hip(self):
        return self._rt.ready()

    # ---- reference inference.py:57-76
    def mix_speakers(self, speaker_mix):
        dev = next(self.parameters()).device
        mixed_enc = mixed_dur = None
        hip = self._rt.ready()
        for spk_id, weight in speaker_mix:
            ids = torch.tensor([spk_id], device=dev, dtype=torch.long)
            e_enc, e_dur = hip.speaker_embedding(0, ids), hip.speaker_embedding(1, ids)
            mixed_enc = weight * e_enc if mixed_enc is None else mixed_enc + weight * e_enc
            mixed_dur = weight * e_dur if mixed_dur is None else mixed_dur + weight * e_dur
        return mixed_enc, mixed_dur

    #: what to do when the default arithmetic (fp16 two-term split) met an operand beyond +-65504 (include/mtts.h "range
    #: guard"): "rerun" the call on the full-range arithmetic (three bf16 terms), "raise", or "ignore" (no flag read, no sync)
    range_policy = "rerun"

    def synthesise(self, x, x_lengths, n_timesteps, speaker=0, voice_mix=None, scale_correction=1.0, length_scale=1.0,
                   debug=False, z=None, sync_max=None, per_request_padding=False, speaker_embeddings=None):
        """``_synthesise`` + the range guard: one read of the sticky device flag per call (a stream synchronisation)."""
        args = (x, x_lengths, n_timesteps, speaker, voice_mix, scale_correction, length_scale, debug, z, sync_max, per_request_padding,
                speaker_embeddings)
        rt = self._rt
        if rt.use_wide:                       # a previous call or the weights already needed the wide arithmetic
            return self._synthesise(*args)
        hip = rt.ready()
        if hip.gemm_terms() not in (1, 2, 16, 17) or self.range_policy == "ignore":     # only the fp16-based arithmetics saturate (17: the text encoder's)
            return self._synthesise(*args)
        saturated = hip.weights_saturate()
        out = None
        if not saturated:
            out = self._synthesise(*args)
            flags = torch.cat([hip.range_flags(), hip.pair_timeouts()]).tolist()       # one read, one synchronisation
            if flags[2]:
                raise RuntimeError("matcha-tts-24k_amd: a pair-form chain launch timed out waiting for its partner workgroup (another "
                                   "kernel held CUs during the launch?); set MTTS_CHAIN_PAIR=0")
            saturated = bool(flags[0] or flags[1])
        if not saturated:
            return out
        if self.range_policy == "raise" or sync_max is not None:     # (a rank-local rerun would repeat sync_max's collective)
            raise FloatingPointError("matcha-tts-24k_amd: an operand left the fp16 range (|x| > 65504) in the default split "
                                     "arithmetic; set model.range_policy = 'rerun' or MTTS_GEMM_TERMS=6")
        if not getattr(self, "_range_warned", False):
            print("[matcha-tts-24k_amd] an operand left the fp16 range of the default arithmetic: this model now runs on "
                  "three-term bf16 products (full fp32 range)")
            object.__setattr__(self, "_range_warned", True)
        rt.use_wide = True                     # sticky: a checkpoint that overflows once will do so again
        return self._synthesise(*args)

    @torch.inference_mode()
    def speaker_rows(self, voices):
        """One (e_enc, e_dur) row per entry of ``voices``: an int speaker id, or a voice mix ``[(id, weight), ...]`` combined as
        ``mix_speakers`` does (reference inference.py:57-76).  Returns two [len(voices), spk_emb_dim] device tensors."""
        dev = next(self.parameters()).device
        hip = self._rt.ready()
        enc, dur = [], []
        for v in voices:
            if isinstance(v, (list, tuple)):
                e, d = self.mix_speakers(v)
            else:
                ids = torch.tensor([int(v)], device=dev, dtype=torch.long)
                e, d = hip.speaker_embedding(0, ids), hip.speaker_embedding(1, ids)
            enc.append(e)
            dur.append(d)
        return torch.cat(enc, 0), torch.cat(dur, 0)

    @torch.inference_mode()
    def _synthesise(self, x, x_lengths, n_timesteps, speaker=0, voice_mix=None, scale_correction=1.0, length_scale=1.0,
                    debug=False, z=None, sync_max=None, per_request_padding=False, speaker_embeddings=None):
        """Text ids -> mel (reference inference.py:78-183).  Returns ``{"mel": [B, n_feats, T_valid_max]}`` (+ the
        reference's debug tensors when ``debug``).

        ``z``: explicit noise [B, n_feats, T_pad], or a callable ``T_pad -> noise``; default = the device seed-42
        generator like the reference.  ``sync_max``: callable mapping this process's maximum fine length to the
        batch-wide one (data-parallel shards must pad like the whole batch, see dp.py).
        ``speaker_embeddings``: per-utterance ``(e_enc, e_dur)`` rows, e.g. from ``speaker_rows`` (overrides speaker / voice_mix).
        ``per_request_padding``: the reference derives the padded length, and with it the GroupNorm statistics, the
        attention key set and the noise shape, from the longest utterance of the call, so a request's mel depends on what
        it is batched with.  With this flag every utterance is padded (logically) to its OWN length: each row of a ragged
        batch equals the batch-of-one result for that request to rounding (what a dynamic batcher in front of the reference's
        one-request-at-a-time server needs); one extra host read of the B fine lengths."""
        hip = self._rt.ready()
        dev = x.device
        B = x.shape[0]
        if speaker_embeddings is not None:      # (e_enc, e_dur) [B, spk_emb_dim] each: a batch that mixes plain voices and voice mixes
            e_enc, e_dur = speaker_embeddings
        elif voice_mix is not None:
            e_enc, e_dur = self.mix_speakers(voice_mix)
        else:
            ids = torch.as_tensor(speaker, dtype=torch.long, device=dev).reshape(-1)
            e_enc, e_dur = hip.speaker_embedding(0, ids), hip.speaker_embedding(1, ids)
        if e_enc.shape[0] not in (1, B):
            raise ValueError("speaker must be an int or a LongTensor with one id per utterance")

        mu_x, logw, x_mask = self.encoder(x, x_lengths, e_enc, e_dur)
        durations, cum, y_fine_lengths = hip.durations(logw, x_mask, scale_correction, length_scale)
        # the one host sync of the path, as in the reference (utils/model.py:19: .item())
        max_fine = int(y_fine_lengths.max().item())
        if sync_max is not None:
            max_fine = int(sync_max(max_fine))
        t_pad = fix_len_compatibility(max_fine)
        if callable(z):
            z = z(t_pad)
        mu_y, y_mask, y_lengths = hip.align_pool(mu_x, cum, y_fine_lengths, t_pad)
        y_max_length = max((max_fine + 1) // 2, 1)

        t_len = None
        if per_request_padding:
            if sync_max is not None:
                raise ValueError("per_request_padding needs no batch-wide length: do not combine it with sync_max")
            t_len = [fix_len_compatibility(max(int(v), 1)) for v in y_fine_lengths.tolist()]
        mel = self.decoder(mu_y, y_mask, n_timesteps, z=z, t_out=y_max_length, out_scale=self._rt.mel_std,
                           out_shift=self._rt.mel_mean, t_len=t_len, y_lengths=y_lengths, y_max=y_max_length)
        if not debug:
            return {"mel": mel, "mel_lengths": y_lengths}
        encoder_mel = mu_y[:, :, :y_max_length] * self._rt.mel_std + self._rt.mel_mean
        raw = ((torch.exp(logw) - 2) * x_mask).squeeze(1)
        return {"mel": mel, "encoder_mel": encoder_mel, "phoneme_durations": durations, "raw_phoneme_durations": raw,
                "mel_lengths": y_lengths, "mu_y": mu_y, "y_mask": y_mask, "logw": logw, "mu_x": mu_x}


def _plain(obj):
    """OmegaConf containers -> plain python, when omegaconf is importable (it is not on the GPU box)."""
    try:
        from omegaconf import OmegaConf  # type: ignore
        if OmegaConf.is_config(obj):
            return OmegaConf.to_container(obj, resolve=True)
    except Exception:
        pass
    return obj


def load_matcha(model_name, checkpoint_path):
    """reference inference.py:186-197: Lightning checkpoint with ``hyper_parameters`` + ``state_dict``; also accepts a
    directory written by ``checkpoint.convert_lightning_checkpoint`` (flat safetensors + JSON, no lightning / omegaconf)."""
    print(f"[!] Loading {model_name}!")
    from . import checkpoint as ck
    if ck.is_converted(checkpoint_path):
        hp, sd = ck.load_converted(checkpoint_path)
        model = MatchaTTSInfer(**hp.as_reference_kwargs())
        model.load_state_dict(sd, strict=True)
        model._rt.cache_dir = checkpoint_path        # the packed weight image is cached beside the converted tensors
    else:
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        hparams = dict(_plain(ckpt["hyper_parameters"]))
        hparams.pop("optimizer", None)
        hparams.pop("scheduler", None)
        model = MatchaTTSInfer(**hparams)
        model.load_state_dict(ckpt["state_dict"], strict=False)
    model = model.to(DEVICE).eval()
    print(f"[+] {model_name} loaded!")
    return model


def process_text(text: str, language: str):
    """reference inference.py:212-220.  The phonemizer (eSpeak + NeMo) is a CPU front end outside this package; it is
    taken from the reference's ``matcha.text`` when that package is installed."""
    try:
        from matcha.text.phonemizers import multilingual_phonemizer  # type: ignore
    except Exception as e:  # pragma: no cover - depends on the host installation
        raise RuntimeError("process_text needs the reference's matcha.text phonemizer (eSpeak/NeMo); "
                           "feed phoneme ids to synthesise() directly otherwise") from e
    import re
    emphasized = re.sub(r"(?<![?!])\?(?![?!])", "??", text)
    separated, ids = multilingual_phonemizer(emphasized, language)
    x = torch.tensor(ids, dtype=torch.long, device=DEVICE)[None]
    x_lengths = torch.tensor([x.shape[-1]], dtype=torch.long, device=DEVICE)
    return {"x_orig": text, "x": x, "x_lengths": x_lengths, "x_phones": "".join(separated), "x_phone_ids": ids}


def load_vocoder(vocoder_name, checkpoint=None, state_dict=None):
    """reference inference.py:223-231.  The Vocos-24k head runs on the HIP library (vocoder.py); weights come from a local
    file (``VOCOS_CHECKPOINT``) because the reference's ``from_pretrained`` hub fetch needs a network."""
    print(f"[!] Loading {vocoder_name}!")
    if vocoder_name != "vocos":
        raise NotImplementedError(f"Vocoder {vocoder_name} not implemented!")
    from .vocoder import load_model
    vocoder = load_model(DEVICE, checkpoint=checkpoint, state_dict=state_dict)
    print(f"[+] {vocoder_name} loaded!")
    return vocoder


def _waveform_on_device(mel, vocoder):
    """Vocoder + peak normalisation of reference inference.py:260-264, left on the device."""
    audio = vocoder(mel)
    max_abs = audio.abs().max()
    if max_abs > 1.0:
        audio = audio / max_abs * 0.95
    return audio


def to_waveform(mel, vocoder):
    """reference inference.py:260-265."""
    return _waveform_on_device(mel, vocoder).cpu().squeeze()


def trim_trailing_silence(audio, silence_threshold_db=-60.0):
    """reference inference.py:268-287, window for window: 10 ms windows anchored at sample 0 (the ``len % window`` remainder is
    never examined), RMS per window, count the run of trailing windows with ``rms < threshold`` (strict; a NaN window stops the
    run as in the reference's loop), drop ``count * window`` samples from the END of the signal.  All full windows may go.
    Works on a 1-D tensor on any device: the window RMS and the run length are computed where the audio lives (one scalar
    comes back to the host), so ``pipeline`` trims before the device-to-host copy."""
    win = int(0.01 * SAMPLE_RATE)
    thr = 10 ** (silence_threshold_db / 20.0)
    n_full = len(audio) // win
    if n_full == 0:
        return audio
    rms = audio[: n_full * win].reshape(n_full, win).pow(2).mean(dim=1).sqrt()
    loud = torch.logical_not(rms < thr)
    idx = torch.arange(1, n_full + 1, device=audio.device)
    last_loud = int((loud * idx).max())            # 1-based index of the last window that is not silent; 0 = none
    trim = (n_full - last_loud) * win
    if trim == 0:
        return audio
    return audio[:-trim]


@torch.inference_mode()
def pipeline(model, vocoder, text, speaker=0, voice_mix=None, n_timesteps=DEFAULT_NUM_STEPS, scale_correction=1.0,
             length_scale=1.0, debug=False):
    """reference inference.py:233-257.  The reference wraps ``synthesise`` in ``torch.autocast`` (fp16 on its CUDA device);
    here the estimator's arithmetic is chosen when the model is created (``MTTS_GEMM_TERMS``: default fp32-equivalent; 1 = fp16
    operands / fp32 accumulate, the autocast arithmetic), not per call.  The trailing-silence trim runs on the device."""
    primary = voice_mix[0][0] if voice_mix is not None else speaker
    language = next(v["lang"] for v in VOICES if v["id"] == str(primary))
    tp = process_text(text, language)
    out = model.synthesise(tp["x"], tp["x_lengths"], n_timesteps=n_timesteps, speaker=speaker, voice_mix=voice_mix,
                           scale_correction=scale_correction, length_scale=length_scale, debug=debug)
    waveform = trim_trailing_silence(_waveform_on_device(out["mel"], vocoder).squeeze()).cpu()
    if not debug:
        return waveform
    durs = out["phoneme_durations"].squeeze(0).tolist()
    raws = out["raw_phoneme_durations"].squeeze(0).tolist()
    pairs = list(zip(tp["x_phones"], raws, durs))
    return waveform, to_waveform(out["encoder_mel"], vocoder), pairs


def convert_to_mp3(waveform):
    """reference inference.py:290-298: int16 PCM -> MP3 through the reference installation's LAME binding
    (matcha.utils.mp3_converter.encode_mp3, vbr_quality=5, algorithm_quality=5).  The codec is a post-waveform CPU step outside
    the synthesis path: it is delegated, not re-implemented, and raises ImportError where the reference
    package (and its lameenc dependency) is not installed."""
    import time
    import numpy as np
    from matcha.utils.mp3_converter import encode_mp3  # type: ignore
    start = time.perf_counter()
    audio_np = (waveform.detach().cpu().numpy() * 32767).astype(np.int16)
    wav_size = audio_np.size * 2
    mp3_data = encode_mp3(audio_np, sample_rate=SAMPLE_RATE, vbr_quality=5, algorithm_quality=5)
    pct = (len(mp3_data) / wav_size * 100) if wav_size > 0 else 0
    print(f"MP3 conversion: {(time.perf_counter() - start) * 1000:.1f}ms | {pct:.0f}% size")
    return mp3_data


def convert_to_opus_ogg(waveform):
    """reference inference.py:301-322: int16 mono PCM -> Ogg/Opus with PyAV (libopus, 48 kbit/s, compression_level 5), the
    reference's own settings and call sequence.  PyAV is imported here, as the reference imports it at module load; where it is
    not installed the ImportError says so (no silent fallback)."""
    import io
    import time
    import numpy as np
    try:
        import av  # type: ignore
    except ImportError as e:
        raise ImportError("convert_to_opus_ogg needs PyAV (`av`), as the reference's matcha/inference.py does") from e
    start = time.perf_counter()
    audio_np = (waveform.detach().cpu().numpy() * 32767).astype(np.int16).reshape(1, -1)
    wav_size = audio_np.size * 2
    buffer = io.BytesIO()
    container = av.open(buffer, mode="w", format="ogg")
    stream = container.add_stream("libopus", rate=SAMPLE_RATE)
    stream.layout = "mono"
    stream.bit_rate = 48000
    stream.options = {"compression_level": "5"}
    frame = av.AudioFrame.from_ndarray(audio_np, format="s16", layout="mono")
    frame.sample_rate = SAMPLE_RATE
    for packet in stream.encode(frame):
        container.mux(packet)
    for packet in stream.encode():
        container.mux(packet)
    container.close()
    ogg_data = buffer.getvalue()
    pct = (len(ogg_data) / wav_size * 100) if wav_size > 0 else 0
    print(f"OGG conversion: {(time.perf_counter() - start) * 1000:.1f}ms | {pct:.0f}% size")
    return bytes(ogg_data)
