"""Vocos-24k head on MI355X: drop-in for the reference's ``matcha.vocos24k.vocos_wrapper`` (VocosWrapper / load_model,
reference matcha/vocos24k/vocos_wrapper.py:3-16) and for ``vocos.Vocos.decode``.

The reference fetches pretrained weights by name from the HF hub (``charactr/vocos-mel-24khz``); there is no network
here, so ``load_model`` reads a local state dict (``VOCOS_CHECKPOINT`` or an explicit path) in the vocos package's
key layout.  Arithmetic runs in libmtts_hip.so (``mtts_vocos_decode``); there is no CPU path.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _hip
from .modules import ParamTree
from .synthetic import vocos_spec

# reference matcha/vocos24k/config.yaml:10-24
DEFAULT_CFG = dict(n_mels=100, dim=512, inter=1536, layers=8, n_fft=1024, hop=256)


class Vocos(ParamTree):
    """Parameter holder under the vocos state-dict names + ``decode(mel)``."""

    def __init__(self, **cfg):
        super().__init__()
        self.cfg = {**DEFAULT_CFG, **cfg}
        c = self.cfg
        for key, shape, kind in vocos_spec(n_mels=c["n_mels"], dim=c["dim"], inter=c["inter"], layers=c["layers"], n_fft=c["n_fft"]):
            self.attach(key, shape, kind)
        object.__setattr__(self, "_ctx", None)
        object.__setattr__(self, "_weights", None)
        object.__setattr__(self, "_ws", {})
        object.__setattr__(self, "_dirty", True)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        # the published checkpoint also carries feature-extractor buffers and the iSTFT window; only the decoder is used
        sd = {k: v for k, v in state_dict.items() if k.startswith(("backbone.", "head.out."))}
        out = super().load_state_dict(sd, strict=strict, assign=assign)
        object.__setattr__(self, "_dirty", True)
        return out

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        object.__setattr__(self, "_dirty", True)
        return r

    def _ready(self):
        lib = _hip.load()
        c = self.cfg
        if self._ctx is None:
            ctx = lib.mtts_vocos_create(c["n_mels"], c["dim"], c["inter"], c["layers"], c["n_fft"], c["hop"])
            if not ctx:
                raise RuntimeError("mtts_vocos_create: " + lib.mtts_last_error().decode())
            object.__setattr__(self, "_ctx", ctx)
        if self._dirty:
            p = next(self.parameters())
            if not p.is_cuda:
                raise RuntimeError("matcha-tts-24k_amd: the vocoder must be on a HIP device; there is no CPU path")
            tensors = dict(self.state_dict())
            tensors["aux.window"] = torch.hann_window(c["n_fft"], dtype=torch.float32)   # as torch.istft's caller passes it
            for k, v in tensors.items():
                a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
                _hip.check(lib.mtts_vocos_set_tensor(self._ctx, k.encode(), a.ctypes.data, a.size))
            n = lib.mtts_vocos_weights_bytes(self._ctx)
            if n < 0:
                _hip.check(-1)
            w = torch.empty(n, dtype=torch.uint8, device=p.device)
            _hip.check(lib.mtts_vocos_upload_weights(self._ctx, w.data_ptr(), n))
            object.__setattr__(self, "_weights", w)
            self._ws.clear()
            object.__setattr__(self, "_dirty", False)
        return lib

    @torch.inference_mode()
    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, n_mels, T] (or [n_mels, T]) -> audio [B, hop*(T-1)]."""
        lib = self._ready()
        if mel.dim() == 2:
            mel = mel[None]
        if not mel.is_cuda:
            raise RuntimeError("matcha-tts-24k_amd: mel is not on a HIP device; there is no CPU path")
        mel = mel.detach().to(torch.float32).contiguous()
        B, _, T = mel.shape
        need = lib.mtts_vocos_workspace_bytes(self._ctx, B, T)
        key = _hip.stream_ptr()                     # one grow-only scratch buffer per stream (see HipModel._workspace)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = None
            self._ws.pop(key, None)
            ws = torch.empty(need, dtype=torch.uint8, device=mel.device)
            self._ws[key] = ws
        audio = torch.empty(B, self.cfg["hop"] * (T - 1), dtype=torch.float32, device=mel.device)
        _hip.check(lib.mtts_vocos_decode(self._ctx, _hip.ptr(mel), B, T, _hip.ptr(audio), ws.data_ptr(), ws.numel(), _hip.stream_ptr()))
        return audio

    def __del__(self):
        try:
            if self._ctx:
                _hip.load().mtts_vocos_destroy(self._ctx)
        except Exception:
            pass


class VocosWrapper(nn.Module):
    """reference matcha/vocos24k/vocos_wrapper.py:3-9"""

    def __init__(self, model: Vocos):
        super().__init__()
        self.model = model

    def forward(self, mel):
        return self.model.decode(mel)


def load_model(device="cuda", checkpoint: Optional[str] = None, state_dict: Optional[Dict[str, torch.Tensor]] = None):
    """reference matcha/vocos24k/vocos_wrapper.py:11-16, from a local file instead of the HF hub."""
    model = Vocos()
    if state_dict is None:
        path = checkpoint or os.environ.get("VOCOS_CHECKPOINT")
        if not path:
            raise RuntimeError("no network: point VOCOS_CHECKPOINT at a local charactr/vocos-mel-24khz state dict "
                               "(pytorch_model.bin) or pass state_dict=")
        state_dict = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(state_dict, strict=True)
    return VocosWrapper(model.to(device).eval())
