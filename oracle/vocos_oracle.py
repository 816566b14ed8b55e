"""CPU ORACLE for the Vocos-24k head -- TEST INFRASTRUCTURE ONLY (see matcha_oracle.py for the rules).

PARITY UNPINNED: the reference only wraps the third-party ``vocos`` package (reference matcha/vocos24k/vocos_wrapper.py:3-16,
``Vocos.from_pretrained("charactr/vocos-mel-24khz")``); neither its source nor its weights are in the container.  Only the
layer sizes are in-tree (reference matcha/vocos24k/config.yaml:10-24: input_channels 100, dim 512, intermediate_dim 1536,
num_layers 8, n_fft 1024, hop_length 256, padding center).  The block internals below restate the published Vocos
architecture (VocosBackbone / ConvNeXtBlock / ISTFTHead of vocos 0.1.0): k7 convs, LayerNorm eps 1e-6, exact GELU,
layer-scale, exp + clip 1e2, cos/sin phase, torch.istft(center=True, periodic hann window).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def convnext_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ConvNeXtBlock.forward on [B, C, T]."""
    c = x.shape[1]
    r = x
    x = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=c)
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
    x = F.linear(x, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
    x = F.gelu(x)
    x = F.linear(x, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    x = sd[p + "gamma"] * x
    return r + x.transpose(1, 2)


def backbone(sd: SD, mel: torch.Tensor, num_layers: int) -> torch.Tensor:
    """VocosBackbone.forward: [B, n_mels, T] -> [B, T, dim]."""
    x = F.conv1d(mel, sd["backbone.embed.weight"], sd["backbone.embed.bias"], padding=3)
    c = x.shape[1]
    x = F.layer_norm(x.transpose(1, 2), (c,), sd["backbone.norm.weight"], sd["backbone.norm.bias"], eps=1e-6).transpose(1, 2)
    for i in range(num_layers):
        x = convnext_block(sd, f"backbone.convnext.{i}.", x)
    return F.layer_norm(x.transpose(1, 2), (c,), sd["backbone.final_layer_norm.weight"], sd["backbone.final_layer_norm.bias"], eps=1e-6)


def istft_head(sd: SD, x: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """ISTFTHead.forward: [B, T, dim] -> audio [B, hop*(T-1)]."""
    x = F.linear(x, sd["head.out.weight"], sd["head.out.bias"]).transpose(1, 2)
    mag, p = x.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    spec = mag * (torch.cos(p) + 1j * torch.sin(p))
    window = torch.hann_window(n_fft, dtype=x.dtype)
    return torch.istft(spec, n_fft, hop, n_fft, window, center=True)


def decode(sd: SD, mel: torch.Tensor, num_layers: int = 8, n_fft: int = 1024, hop: int = 256) -> torch.Tensor:
    """Vocos.decode(mel) (reference vocos_wrapper.py:8-9)."""
    return istft_head(sd, backbone(sd, mel, num_layers), n_fft, hop)


def to_waveform_scale(audio: torch.Tensor) -> torch.Tensor:
    """reference inference.py:260-264 (without the .cpu().squeeze())."""
    m = audio.abs().max()
    return audio / m * 0.95 if m > 1.0 else audio
