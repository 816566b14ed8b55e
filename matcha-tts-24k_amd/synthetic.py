"""Deterministic synthetic weights and inputs (no checkpoints or datasets exist offline).

Everything is drawn from a counter-based integer hash (splitmix64) so that the reference
model in the survey container, the CPU oracle and the HIP path on the GPU box all
regenerate bit-identical tensors from (seed, tensor name) without shipping weights.
Shapes and key names are those of the reference state dict.

The "duration recipe"  zeroes the duration predictor's last
projection and sets its bias to ln 7, so every token lasts round(e^{ln 7} - 2) = 5 fine
frames: T_pad = 5*Tx, T_valid = ceil(5*Tx/2), far from any rounding boundary.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

from .hparams import PathHParams

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        return z ^ (z >> np.uint64(31))


def _raw(seed: int, stream: int, n: int, lane: int) -> np.ndarray:
    base = _splitmix64(np.array([(seed * 0x100000001B3 + stream * 0x9E3779B1 + lane) & 0xFFFFFFFFFFFFFFFF],
                                dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + base) & _MASK64
    return _splitmix64(ctr)


def portable_uniform(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """float64 uniforms in (0, 1) with 53 random bits."""
    r = _raw(seed, stream, n, lane)
    return ((r >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def portable_normal(seed: int, stream: int, n: int) -> np.ndarray:
    """float32 standard normals (Box-Muller on two hashed uniform streams)."""
    u1 = portable_uniform(seed, stream, n, 0)
    u2 = portable_uniform(seed, stream, n, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return z.astype(np.float32)


def portable_randint(seed: int, stream: int, n: int, high: int) -> np.ndarray:
    return (_raw(seed, stream, n, 2) % np.uint64(high)).astype(np.int64)


def _stream_of(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF


# --------------------------------------------------------------------------------------
# state-dict inventory (names/shapes as produced by the reference with TORCHDYNAMO_DISABLE=1)
# --------------------------------------------------------------------------------------
def state_dict_spec(hp: PathHParams) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every tensor on the path. kind drives the init scale."""
    e, d = hp.encoder, hp.decoder
    S = hp.spk_emb_dim
    H = e.n_channels + S
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def add(key, shape, kind):
        out.append((key, tuple(int(s) for s in shape), kind))

    def conv(prefix, cout, cin, k):
        add(prefix + ".weight", (cout, cin, k), "w")
        add(prefix + ".bias", (cout,), "b")

    def lin(prefix, cout, cin, bias=True):
        add(prefix + ".weight", (cout, cin), "w")
        if bias:
            add(prefix + ".bias", (cout,), "b")

    def chan_ln(prefix, c):
        add(prefix + ".gamma", (c,), "g")
        add(prefix + ".beta", (c,), "nb")

    def torch_norm(prefix, c):
        add(prefix + ".weight", (c,), "g")
        add(prefix + ".bias", (c,), "nb")

    add("speaker_embeddings_enc.weight", (hp.n_spks, S), "spk")
    add("speaker_embeddings_dur.weight", (hp.n_spks, S), "spk")
    add("mel_mean", (), "mel_mean")
    add("mel_std", (), "mel_std")
    # ---- text encoder ----
    add("encoder.emb.weight", (hp.n_vocab, e.n_channels), "emb")
    for i in range(e.prenet_layers):
        conv(f"encoder.prenet.conv_layers.{i}", e.n_channels, e.n_channels, e.prenet_kernel_size)
        chan_ln(f"encoder.prenet.norm_layers.{i}", e.n_channels)
    conv("encoder.prenet.proj", e.n_channels, e.n_channels, 1)
    for i in range(e.n_layers):
        for nm in ("q", "k", "v", "o"):
            conv(f"encoder.encoder.attn_layers.{i}.conv_{nm}", H, H, 1)
        chan_ln(f"encoder.encoder.norm_layers_1.{i}", H)
        conv(f"encoder.encoder.ffn_layers.{i}.conv_1", e.filter_channels, H, e.kernel_size)
        conv(f"encoder.encoder.ffn_layers.{i}.conv_2", H, e.filter_channels, e.kernel_size)
        chan_ln(f"encoder.encoder.norm_layers_2.{i}", H)
    conv("encoder.proj_m.0", e.n_channels, H, 1)
    conv("encoder.proj_m.2", e.n_feats, e.n_channels, 1)
    F = e.dp_filter_channels
    add("encoder.proj_w.spk_proj.weight", (2 * F, S), "film_w")
    add("encoder.proj_w.spk_proj.bias", (2 * F,), "film_b")
    for i in range(e.dp_n_layers):
        conv(f"encoder.proj_w.conv_layers.{i}", F, H if i == 0 else F, e.dp_kernel_size)
        chan_ln(f"encoder.proj_w.norm_layers.{i}", F)
    add("encoder.proj_w.proj.weight", (1, F, 1), "dp_w")
    add("encoder.proj_w.proj.bias", (1,), "dp_b")
    # ---- decoder (U-Net velocity estimator) ----
    P = "decoder.estimator."
    cin0 = 2 * hp.n_feats
    ch = tuple(d.channels)
    temb = ch[0] * 4
    lin(P + "time_mlp.linear_1", temb, cin0)
    lin(P + "time_mlp.linear_2", temb, temb)

    def resnet(prefix, ci, co):
        conv(prefix + ".block1.block.0", co, ci, 3)
        torch_norm(prefix + ".block1.block.1", co)
        lin(prefix + ".mlp.1", co, temb)
        conv(prefix + ".block2.block.0", co, co, 3)
        torch_norm(prefix + ".block2.block.1", co)
        conv(prefix + ".res_conv", co, ci, 1)

    def tblock(prefix, c):
        inner = d.num_heads * d.attention_head_dim
        torch_norm(prefix + ".norm1", c)
        for nm in ("to_q", "to_k", "to_v"):
            lin(prefix + ".attn1." + nm, inner, c, bias=False)
        lin(prefix + ".attn1.to_out.0", c, inner)
        torch_norm(prefix + ".norm3", c)
        lin(prefix + ".ff.net.0.proj", 4 * c, c)
        add(prefix + ".ff.net.0.alpha", (4 * c,), "snake")
        add(prefix + ".ff.net.0.beta", (4 * c,), "snake")
        lin(prefix + ".ff.net.2", c, 4 * c)

    co = cin0
    for i, c in enumerate(ch):
        ci, co = co, c
        resnet(P + f"down_blocks.{i}.0", ci, co)
        for j in range(d.n_blocks):
            tblock(P + f"down_blocks.{i}.1.{j}", co)
        if i < len(ch) - 1:
            conv(P + f"down_blocks.{i}.2.conv", co, co, 3)
        else:
            conv(P + f"down_blocks.{i}.2", co, co, 3)
    for i in range(d.num_mid_blocks):
        resnet(P + f"mid_blocks.{i}.0", ch[-1], ch[-1])
        for j in range(d.n_blocks):
            tblock(P + f"mid_blocks.{i}.1.{j}", ch[-1])
    up = ch[::-1] + (ch[0],)
    for i in range(len(up) - 1):
        ci, co = up[i], up[i + 1]
        resnet(P + f"up_blocks.{i}.0", 2 * ci, co)
        for j in range(d.n_blocks):
            tblock(P + f"up_blocks.{i}.1.{j}", co)
        if i < len(up) - 2:
            add(P + f"up_blocks.{i}.2.conv.weight", (co, co, 4), "wT")   # ConvTranspose1d: [in, out, k]
            add(P + f"up_blocks.{i}.2.conv.bias", (co,), "b")
        else:
            conv(P + f"up_blocks.{i}.2", co, co, 3)
    conv(P + "final_block.block.0", up[-1], up[-1], 3)
    torch_norm(P + "final_block.block.1", up[-1])
    conv(P + "final_proj", hp.n_feats, up[-1], 1)
    return out


def make_state_dict(hp: PathHParams, seed: int = 7, duration_recipe: bool = True,
                    frames_per_token: int = 5) -> Dict[str, torch.Tensor]:
    """Random-init weights of the path's architecture, identical on every machine."""
    sd: Dict[str, torch.Tensor] = {}
    F = hp.encoder.dp_filter_channels
    for key, shape, kind in state_dict_spec(hp):
        n = int(np.prod(shape)) if len(shape) else 1
        z = portable_normal(seed, _stream_of(key), n)
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            v = z / math.sqrt(fan_in)
        elif kind == "wT":
            v = z / math.sqrt(shape[0] * shape[2] / 2.0)
        elif kind == "b":
            v = 0.05 * z
        elif kind == "g":
            v = 1.0 + 0.1 * z
        elif kind == "nb":
            v = 0.1 * z
        elif kind == "snake":
            v = 0.2 * z
        elif kind == "emb":
            v = z * hp.encoder.n_channels ** -0.5
        elif kind == "spk":
            v = 0.5 * z
        elif kind == "film_w":
            v = 0.3 * z / math.sqrt(shape[1])
        elif kind == "film_b":
            v = 0.1 * z
            v[:F] += 1.0
        elif kind == "dp_w":
            v = np.zeros(n, np.float32) if duration_recipe else z / math.sqrt(shape[1])
        elif kind == "dp_b":
            v = (np.full(n, math.log(frames_per_token + 2.0), np.float32) if duration_recipe
                 else np.full(n, math.log(6.0), np.float32))
        elif kind == "mel_mean":
            v = np.array([hp.mel_mean], np.float32)
        elif kind == "mel_std":
            v = np.array([hp.mel_std], np.float32)
        else:
            raise KeyError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shape).copy())
    return sd


def make_inputs(hp: PathHParams, batch: int, n_tokens: int, seed: int = 1234, lengths=None):
    """Phoneme ids ~ U{0..n_vocab-1} [B, Tx] int64, lengths [B] int64, speakers b mod n_spks."""
    ids = portable_randint(seed, _stream_of("ids"), batch * n_tokens, hp.n_vocab).reshape(batch, n_tokens)
    x = torch.from_numpy(ids.copy())
    if lengths is None:
        x_lengths = torch.full((batch,), n_tokens, dtype=torch.long)
    else:
        x_lengths = torch.as_tensor(lengths, dtype=torch.long)
    speakers = torch.arange(batch, dtype=torch.long) % hp.n_spks
    return x, x_lengths, speakers


def cpu_noise(shape, seed: int = 42) -> torch.Tensor:
    """The reference's seed-42 draw on the CPU generator (flow_matching.py:43-55); identical to
    ``torch.randn_like(mu, generator=g)`` for a CPU ``mu``."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


# --------------------------------------------------------------------------------------
# Vocos-24k head (shapes from reference matcha/vocos24k/config.yaml:10-24; key names of the vocos package)
# --------------------------------------------------------------------------------------
def vocos_spec(n_mels: int = 100, dim: int = 512, inter: int = 1536, layers: int = 8, n_fft: int = 1024):
    out = [("backbone.embed.weight", (dim, n_mels, 7), "w"), ("backbone.embed.bias", (dim,), "b"),
           ("backbone.norm.weight", (dim,), "g"), ("backbone.norm.bias", (dim,), "nb")]
    for i in range(layers):
        p = f"backbone.convnext.{i}."
        out += [(p + "dwconv.weight", (dim, 1, 7), "w"), (p + "dwconv.bias", (dim,), "b"),
                (p + "norm.weight", (dim,), "g"), (p + "norm.bias", (dim,), "nb"),
                (p + "pwconv1.weight", (inter, dim), "w"), (p + "pwconv1.bias", (inter,), "b"),
                (p + "pwconv2.weight", (dim, inter), "w"), (p + "pwconv2.bias", (dim,), "b"),
                (p + "gamma", (dim,), "ls")]
    out += [("backbone.final_layer_norm.weight", (dim,), "g"), ("backbone.final_layer_norm.bias", (dim,), "nb"),
            ("head.out.weight", (n_fft + 2, dim), "head_w"), ("head.out.bias", (n_fft + 2,), "head_b")]
    return out


def make_vocos_state_dict(seed: int = 11, **kw) -> Dict[str, torch.Tensor]:
    """Random-init Vocos weights (the pretrained charactr/vocos-mel-24khz cannot be fetched offline)."""
    layers = kw.get("layers", 8)
    sd: Dict[str, torch.Tensor] = {}
    for key, shape, kind in vocos_spec(**kw):
        n = int(np.prod(shape))
        z = portable_normal(seed, _stream_of("vocos." + key), n)
        if kind == "w":
            v = z / math.sqrt(int(np.prod(shape[1:])))
        elif kind == "b":
            v = 0.05 * z
        elif kind == "g":
            v = 1.0 + 0.1 * z
        elif kind == "nb":
            v = 0.1 * z
        elif kind == "ls":                      # layer scale, init 1/num_layers
            v = (1.0 / layers) * (1.0 + 0.2 * z)
        elif kind == "head_w":                  # keep log-magnitudes moderate so exp() stays inside the 1e2 clip mostly
            v = 0.5 * z / math.sqrt(shape[1])
        elif kind == "head_b":
            v = 0.1 * z
        else:
            raise KeyError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shape).copy())
    return sd
