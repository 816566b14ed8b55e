"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's mel-synthesis path,
written functionally over a flat state dict.  It exists to *check* the HIP path; it is
never imported by the product package (``matcha-tts-24k_amd``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pinning: every function below is checked against the reference's own code (imported
unmodified from /root/reference in the build container) by ``tests/golden/make_golden.py``;
the resulting vectors are committed under ``tests/golden/*.npz`` and re-checked on every
CPU test run (tests/test_oracle_golden.py).  Three third-party boundaries have no source in
the container (diffusers ``Attention``, torchdiffeq ``odeint``, vocos): their semantics are
restated from their documented behaviour => "parity unpinned" there.

Reference files restated (all under /root/reference/matcha/):
  utils/model.py:7-68                      sequence_mask, fix_len_compatibility, generate_path, downsample, denormalize
  models/components/text_encoder.py:10-406 LayerNorm, ConvSiluNorm, DurationPredictor, RoPE, MultiHeadAttention, FFN, Encoder, TextEncoder
  models/components/decoder.py:14-426      SinusoidalPosEmb, Block1D, ResnetBlock1D, Downsample1D, TimestepEmbedding, Upsample1D, Decoder
  models/components/transformer.py:14-303  SnakeBeta, FeedForward, BasicTransformerBlock (+ diffusers Attention semantics)
  models/components/flow_matching.py:25-63 BASECFM.forward / solve (+ torchdiffeq fixed-grid solvers)
  inference.py:57-183                      mix_speakers, synthesise
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- utils/model.py
def sequence_mask(length: torch.Tensor, max_length: int) -> torch.Tensor:
    """utils/model.py:7-9"""
    r = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return r.unsqueeze(0) < length.unsqueeze(1)


def fix_len_compatibility(length: int, num_downsamplings: int = 1) -> int:
    """utils/model.py:15-21 -- ceil(length / 2^n) * 2^n as a Python int."""
    f = 2 ** num_downsamplings
    return int(math.ceil(int(length) / f) * f)


def generate_path(duration: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """utils/model.py:24-40 -- one-hot monotonic alignment [B, Tx, Ty] from integer durations."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration.long(), 1)
    path = sequence_mask(cum.view(b * t_x), t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, [0, 0, 1, 0, 0, 0])[:, :-1]
    return path * mask


def downsample(mu_y_fine: torch.Tensor) -> torch.Tensor:
    """utils/model.py:57-68 -- avg_pool1d(k=3, s=2, p=1), padding counted in the divisor."""
    return F.avg_pool1d(mu_y_fine, kernel_size=3, stride=2, padding=1)


def denormalize(x, mean, std):
    """utils/model.py:52-54"""
    return x * std + mean


# ----------------------------------------------------------------------------- text encoder
def channel_layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """text_encoder.py:19-27 -- LayerNorm over the channel axis of [B, C, T], biased variance."""
    mean = torch.mean(x, 1, keepdim=True)
    var = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(var + eps)
    return x * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def prenet_forward(sd: SD, p: str, x: torch.Tensor, x_mask: torch.Tensor, n_layers: int, k: int) -> torch.Tensor:
    """ConvSiluNorm.forward, text_encoder.py:55-62."""
    x_org = x
    for i in range(n_layers):
        x = F.conv1d(x * x_mask, sd[f"{p}conv_layers.{i}.weight"], sd[f"{p}conv_layers.{i}.bias"], padding=k // 2)
        x = channel_layer_norm(x, sd[f"{p}norm_layers.{i}.gamma"], sd[f"{p}norm_layers.{i}.beta"])
        x = F.silu(x)
    x = x_org + F.conv1d(x, sd[f"{p}proj.weight"], sd[f"{p}proj.bias"])
    return x * x_mask


def rope_tables(d: int, n: int, dtype=torch.float32, base: int = 10000):
    """text_encoder.py:140-146 -- cos/sin caches [n, d]; computed in fp32 like the reference, then cast."""
    theta = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    idx = torch.einsum("n,d->nd", torch.arange(n).float(), theta)
    idx2 = torch.cat([idx, idx], dim=1)
    return idx2.cos().to(dtype), idx2.sin().to(dtype)


def apply_rope(x: torch.Tensor, d: int) -> torch.Tensor:
    """text_encoder.py:151-173 -- x [B, H, T, Dh]; rotate the first d dims, half-rotation pairs (i, i+d/2)."""
    t = x.shape[2]
    cos, sin = rope_tables(d, t, x.dtype)
    cos = cos.to(x.device)[None, None]
    sin = sin.to(x.device)[None, None]
    xr, xp = x[..., :d], x[..., d:]
    neg_half = torch.cat([-xr[..., d // 2:], xr[..., : d // 2]], dim=-1)
    xr = xr * cos + neg_half * sin
    return torch.cat([xr, xp], dim=-1)


def sdpa_reference(q, k, v, attn_mask=None, scale=None):
    """torch.nn.functional.scaled_dot_product_attention, math form.  bool mask: False => -inf;
    float mask: added to the logits.  Rows with no allowed key return 0 (the CPU fused kernel's behaviour
    observed in the survey; such rows are zeroed by ``* x_mask`` downstream anyway)."""
    scale = (1.0 / math.sqrt(q.shape[-1])) if scale is None else scale
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            s = s.masked_fill(~attn_mask, float("-inf"))
        else:
            s = s + attn_mask
    m = s.amax(dim=-1, keepdim=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    p = torch.where(l > 0, p / l.clamp_min(1e-38), torch.zeros_like(p))
    return torch.matmul(p, v)


def mha_forward(sd: SD, p: str, x: torch.Tensor, attn_mask: torch.Tensor, n_heads: int, use_torch_sdpa: bool) -> torch.Tensor:
    """MultiHeadAttention.forward/attention, text_encoder.py:210-237 (self-attention: c == x)."""
    q = F.conv1d(x, sd[p + "conv_q.weight"], sd[p + "conv_q.bias"])
    k = F.conv1d(x, sd[p + "conv_k.weight"], sd[p + "conv_k.bias"])
    v = F.conv1d(x, sd[p + "conv_v.weight"], sd[p + "conv_v.bias"])
    b, c, t = q.shape
    dh = c // n_heads

    def split(z):  # "b (h c) t -> b h t c"
        return z.view(b, n_heads, dh, t).transpose(2, 3)

    q, k, v = split(q), split(k), split(v)
    d_rope = int(dh * 0.5)
    q, k = apply_rope(q, d_rope), apply_rope(k, d_rope)
    if use_torch_sdpa:
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask.bool())
    else:
        o = sdpa_reference(q, k, v, attn_mask.bool())
    o = o.transpose(2, 3).reshape(b, c, t)  # "b h t c -> b (h c) t"
    return F.conv1d(o, sd[p + "conv_o.weight"], sd[p + "conv_o.bias"])


def encoder_stack_forward(sd: SD, p: str, x, x_mask, n_layers, n_heads, k, use_torch_sdpa=True):
    """Encoder.forward, text_encoder.py:299-316 (post-LN transformer, conv FFN)."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    for i in range(n_layers):
        x = x * x_mask
        y = mha_forward(sd, f"{p}attn_layers.{i}.", x, attn_mask, n_heads, use_torch_sdpa)
        x = channel_layer_norm(x + y, sd[f"{p}norm_layers_1.{i}.gamma"], sd[f"{p}norm_layers_1.{i}.beta"])
        # FFN.forward, text_encoder.py:253-258
        y = F.conv1d(x * x_mask, sd[f"{p}ffn_layers.{i}.conv_1.weight"], sd[f"{p}ffn_layers.{i}.conv_1.bias"], padding=k // 2)
        y = torch.relu(y)
        y = F.conv1d(y * x_mask, sd[f"{p}ffn_layers.{i}.conv_2.weight"], sd[f"{p}ffn_layers.{i}.conv_2.bias"], padding=k // 2)
        y = y * x_mask
        x = channel_layer_norm(x + y, sd[f"{p}norm_layers_2.{i}.gamma"], sd[f"{p}norm_layers_2.{i}.beta"])
    return x * x_mask


def duration_predictor_forward(sd: SD, p: str, x, x_mask, spk_emb, n_layers, filt, k):
    """DurationPredictor.forward, text_encoder.py:101-112 (conv -> ReLU -> LN -> FiLM)."""
    film = F.linear(spk_emb, sd[p + "spk_proj.weight"], sd[p + "spk_proj.bias"]).unsqueeze(-1)
    gamma, beta = torch.split(film, filt, dim=1)
    for i in range(n_layers):
        x = F.conv1d(x * x_mask, sd[f"{p}conv_layers.{i}.weight"], sd[f"{p}conv_layers.{i}.bias"], padding=k // 2)
        x = torch.relu(x)
        x = channel_layer_norm(x, sd[f"{p}norm_layers.{i}.gamma"], sd[f"{p}norm_layers.{i}.beta"])
        x = x * gamma + beta
    x = F.conv1d(x * x_mask, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return x * x_mask


def text_encoder_forward(sd: SD, hp, x: torch.Tensor, x_lengths: torch.Tensor, e_enc: torch.Tensor,
                         e_dur: torch.Tensor, prefix: str = "encoder.", use_torch_sdpa: bool = True):
    """TextEncoder.forward, text_encoder.py:375-406.  Returns mu_x [B,nf,Tx], logw [B,1,Tx], x_mask [B,1,Tx]."""
    e = hp.encoder
    dt = sd[prefix + "emb.weight"].dtype
    h = F.embedding(x, sd[prefix + "emb.weight"]) * math.sqrt(e.n_channels)
    h = h.transpose(1, -1)
    x_mask = sequence_mask(x_lengths, h.shape[2]).unsqueeze(1).to(dt)
    h = prenet_forward(sd, prefix + "prenet.", h, x_mask, e.prenet_layers, e.prenet_kernel_size)
    h = torch.cat([h, e_enc.unsqueeze(-1).expand(-1, -1, h.shape[-1])], dim=1)
    h = encoder_stack_forward(sd, prefix + "encoder.", h, x_mask, e.n_layers, e.n_heads, e.kernel_size, use_torch_sdpa)
    mu = F.conv1d(h, sd[prefix + "proj_m.0.weight"], sd[prefix + "proj_m.0.bias"])
    mu = F.conv1d(F.silu(mu), sd[prefix + "proj_m.2.weight"], sd[prefix + "proj_m.2.bias"]) * x_mask
    logw = duration_predictor_forward(sd, prefix + "proj_w.", h, x_mask, e_dur, e.dp_n_layers, e.dp_filter_channels,
                                      e.dp_kernel_size)
    return mu, logw, x_mask


# ----------------------------------------------------------------------------- decoder (velocity estimator)
def sinusoidal_pos_emb(t: torch.Tensor, dim: int, scale: float = 1000.0) -> torch.Tensor:
    """SinusoidalPosEmb.forward, decoder.py:20-29.  The frequency table is built in fp32 as the reference does."""
    if t.ndim < 1:
        t = t.unsqueeze(0)
    half = dim // 2
    c = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half).float() * -c).to(t.dtype)
    emb = scale * t.unsqueeze(1) * freq.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def block1d(sd: SD, p: str, x, mask, groups: int = 8):
    """Block1D.forward, decoder.py:43-45: Mish(GroupNorm8(conv_k3(x*m))) * m."""
    y = F.conv1d(x * mask, sd[p + "block.0.weight"], sd[p + "block.0.bias"], padding=1)
    y = F.group_norm(y, groups, sd[p + "block.1.weight"], sd[p + "block.1.bias"], eps=1e-5)
    return F.mish(y) * mask


def resnet_block1d(sd: SD, p: str, x, mask, temb):
    """ResnetBlock1D.forward, decoder.py:58-63."""
    h = block1d(sd, p + "block1.", x, mask)
    h = h + F.linear(F.mish(temb), sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"]).unsqueeze(-1)
    h = block1d(sd, p + "block2.", h, mask)
    return h + F.conv1d(x * mask, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])


def snake_beta(sd: SD, p: str, x):
    """SnakeBeta.forward, transformer.py:61-77 (log-scale alpha/beta)."""
    x = F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])
    alpha = torch.exp(sd[p + "alpha"])
    beta = torch.exp(sd[p + "beta"])
    return x + (1.0 / (beta + 0.000000001)) * torch.pow(torch.sin(x * alpha), 2)


def transformer_block(sd: SD, p: str, x, mask_bt, heads: int, dim_head: int, use_torch_sdpa: bool = True):
    """BasicTransformerBlock.forward, transformer.py:230-303 with attn2=None (self-attention only),
    + diffusers Attention/AttnProcessor2_0 semantics: q/k/v without bias, float mask
    [B,T] broadcast to [B,heads,1,T] and passed *as is* to SDPA => ADDITIVE bias (+1 valid, +0 padded)."""
    b, t, c = x.shape
    h = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-5)
    q = F.linear(h, sd[p + "attn1.to_q.weight"])
    k = F.linear(h, sd[p + "attn1.to_k.weight"])
    v = F.linear(h, sd[p + "attn1.to_v.weight"])

    def split(z):
        return z.view(b, t, heads, dim_head).transpose(1, 2)

    bias = mask_bt.to(x.dtype).view(b, 1, 1, t).expand(b, heads, 1, t)
    if use_torch_sdpa:
        o = F.scaled_dot_product_attention(split(q), split(k), split(v), attn_mask=bias, dropout_p=0.0, is_causal=False)
    else:
        o = sdpa_reference(split(q), split(k), split(v), bias)
    o = o.transpose(1, 2).reshape(b, t, heads * dim_head)
    x = F.linear(o, sd[p + "attn1.to_out.0.weight"], sd[p + "attn1.to_out.0.bias"]) + x
    h = F.layer_norm(x, (c,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], eps=1e-5)
    h = snake_beta(sd, p + "ff.net.0.", h)
    h = F.linear(h, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    return h + x


def decoder_forward(sd: SD, hp, x, mask, mu, t, prefix: str = "decoder.estimator.", use_torch_sdpa: bool = True):
    """Decoder.forward, decoder.py:359-426.  x, mu [B,nf,T]; mask [B,1,T]; t 0-dim or [1]."""
    d = hp.decoder
    P = prefix
    cin = 2 * hp.n_feats
    temb = sinusoidal_pos_emb(t.to(x.dtype), cin)
    temb = F.linear(temb, sd[P + "time_mlp.linear_1.weight"], sd[P + "time_mlp.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd[P + "time_mlp.linear_2.weight"], sd[P + "time_mlp.linear_2.bias"])
    x = torch.cat([x, mu], dim=1)

    def tblocks(pp, x, m):
        x = x.transpose(1, 2)
        m_bt = m[:, 0, :]
        for j in range(d.n_blocks):
            x = transformer_block(sd, f"{pp}{j}.", x, m_bt, d.num_heads, d.attention_head_dim, use_torch_sdpa)
        return x.transpose(1, 2)

    hiddens, masks = [], [mask]
    n_lvl = len(d.channels)
    for i in range(n_lvl):
        m = masks[-1]
        x = resnet_block1d(sd, f"{P}down_blocks.{i}.0.", x, m, temb)
        x = tblocks(f"{P}down_blocks.{i}.1.", x, m)
        hiddens.append(x)
        if i < n_lvl - 1:
            x = F.conv1d(x * m, sd[f"{P}down_blocks.{i}.2.conv.weight"], sd[f"{P}down_blocks.{i}.2.conv.bias"], stride=2, padding=1)
        else:
            x = F.conv1d(x * m, sd[f"{P}down_blocks.{i}.2.weight"], sd[f"{P}down_blocks.{i}.2.bias"], padding=1)
        masks.append(m[:, :, ::2])
    masks = masks[:-1]
    m_mid = masks[-1]
    for i in range(d.num_mid_blocks):
        x = resnet_block1d(sd, f"{P}mid_blocks.{i}.0.", x, m_mid, temb)
        x = tblocks(f"{P}mid_blocks.{i}.1.", x, m_mid)
    m_up = None
    for i in range(n_lvl):
        m_up = masks.pop()
        x = resnet_block1d(sd, f"{P}up_blocks.{i}.0.", torch.cat([x, hiddens.pop()], dim=1), m_up, temb)
        x = tblocks(f"{P}up_blocks.{i}.1.", x, m_up)
        if i < n_lvl - 1:
            x = F.conv_transpose1d(x * m_up, sd[f"{P}up_blocks.{i}.2.conv.weight"], sd[f"{P}up_blocks.{i}.2.conv.bias"], stride=2, padding=1)
        else:
            x = F.conv1d(x * m_up, sd[f"{P}up_blocks.{i}.2.weight"], sd[f"{P}up_blocks.{i}.2.bias"], padding=1)
    x = block1d(sd, P + "final_block.", x, m_up)
    out = F.conv1d(x * m_up, sd[P + "final_proj.weight"], sd[P + "final_proj.bias"])
    return out * mask


# ----------------------------------------------------------------------------- CFM / ODE
def odeint_fixed(f: Callable, y0: torch.Tensor, t_span: torch.Tensor, method: str) -> torch.Tensor:
    """torchdiffeq.odeint(..., method=) for the fixed-grid solvers the reference recommends
    (flow_matching.py:60-63; configs/model/cfm/default.yaml:3-4): euler, midpoint, rk4 (torchdiffeq's rk4 is the
    3/8 rule).  Grid = t_span; returns the final state.  f(t, y) receives t as a 0-dim tensor."""
    y = y0
    for i in range(len(t_span) - 1):
        t0, t1 = t_span[i], t_span[i + 1]
        dt = t1 - t0
        if method == "euler":
            y = y + dt * f(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            y = y + dt * f(t0 + half, y + half * f(t0, y))
        elif method == "rk4":
            k1 = f(t0, y)
            k2 = f(t0 + dt / 3, y + dt * k1 / 3)
            k3 = f(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
            k4 = f(t1, y + dt * (k1 - k2 + k3))
            y = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        else:
            raise ValueError(f"unsupported solver {method!r}")
    return y


def cfm_forward(sd: SD, hp, mu, mask, n_timesteps: int, solver: Optional[str] = None, z: Optional[torch.Tensor] = None,
                use_torch_sdpa: bool = True):
    """BASECFM.forward + solve, flow_matching.py:25-63.  ``z`` overrides the seed-42 draw (noise only, mu is added
    here when use_mu_prior)."""
    if z is None:
        g = torch.Generator(device="cpu")
        g.manual_seed(42)
        z = torch.randn(mu.shape, generator=g, dtype=torch.float32).to(mu.dtype)
    x0 = mu + z if hp.use_mu_prior else z
    t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32).to(mu.dtype)
    f = lambda t, y: decoder_forward(sd, hp, y, mask, mu, t, use_torch_sdpa=use_torch_sdpa)
    return odeint_fixed(f, x0, t_span, solver or hp.solver)


# ----------------------------------------------------------------------------- synthesise
def speaker_embeddings(sd: SD, speaker, voice_mix=None):
    """inference.py:57-76,115-121.  ``speaker`` int or LongTensor[B] (batched extension)."""
    if voice_mix is not None:
        e_enc = sum(w * sd["speaker_embeddings_enc.weight"][i][None] for i, w in voice_mix)
        e_dur = sum(w * sd["speaker_embeddings_dur.weight"][i][None] for i, w in voice_mix)
        return e_enc, e_dur
    idx = torch.as_tensor(speaker, dtype=torch.long).reshape(-1)
    return sd["speaker_embeddings_enc.weight"][idx], sd["speaker_embeddings_dur.weight"][idx]


def durations_from_logw(logw, x_mask, scale_correction=1.0, length_scale=1.0):
    """inference.py:127-143."""
    d = ((torch.exp(logw) - 2) * x_mask).squeeze(1)
    d = d * scale_correction * length_scale
    return d.round().clamp(min=1) * x_mask.squeeze(1)


def align_and_pool(mu_x, durations, x_mask):
    """inference.py:146-167.  Returns mu_y [B,nf,T_pad], y_mask [B,1,T_pad], y_lengths, y_max_length, T_pad."""
    y_fine_lengths = torch.clamp_min(durations.sum(dim=1).long(), 1)
    t_fine = fix_len_compatibility(int(y_fine_lengths.max())) * 2
    y_fine_mask = sequence_mask(y_fine_lengths, t_fine).unsqueeze(1).to(x_mask.dtype)
    attn_mask_fine = x_mask.unsqueeze(-1) * y_fine_mask.unsqueeze(2)
    attn_fine = generate_path(durations, attn_mask_fine.squeeze(1))
    mu_y_fine = torch.matmul(mu_x, attn_fine.to(mu_x.dtype))
    mu_y = downsample(mu_y_fine)
    t_pad = t_fine // 2
    y_lengths = torch.clamp_min((y_fine_lengths + 1) // 2, 1)
    y_mask = sequence_mask(y_lengths, t_pad).unsqueeze(1).to(x_mask.dtype)
    return mu_y, y_mask, y_lengths, int(y_lengths.max()), t_pad


def synthesise(sd: SD, hp, x, x_lengths, n_timesteps, speaker=0, voice_mix=None, scale_correction=1.0,
               length_scale=1.0, solver: Optional[str] = None, z: Optional[torch.Tensor] = None,
               use_torch_sdpa: bool = True) -> dict:
    """MatchaTTSInfer.synthesise, inference.py:78-183 (batched speakers allowed)."""
    e_enc, e_dur = speaker_embeddings(sd, speaker, voice_mix)
    if e_enc.shape[0] == 1 and x.shape[0] > 1:
        e_enc, e_dur = e_enc.expand(x.shape[0], -1), e_dur.expand(x.shape[0], -1)
    mu_x, logw, x_mask = text_encoder_forward(sd, hp, x, x_lengths, e_enc, e_dur, use_torch_sdpa=use_torch_sdpa)
    durations = durations_from_logw(logw, x_mask, scale_correction, length_scale)
    mu_y, y_mask, y_lengths, y_max, t_pad = align_and_pool(mu_x, durations, x_mask)
    dec = cfm_forward(sd, hp, mu_y, y_mask, n_timesteps, solver, z, use_torch_sdpa)[:, :, :y_max]
    mel = denormalize(dec, sd["mel_mean"], sd["mel_std"])
    return {"mel": mel, "mel_lengths": y_lengths, "mu_x": mu_x, "logw": logw, "durations": durations, "mu_y": mu_y,
            "y_mask": y_mask, "decoder_outputs": dec, "t_pad": t_pad}
