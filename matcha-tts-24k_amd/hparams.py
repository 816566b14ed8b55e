"""Hyper-parameters of the mel-synthesis path, normalised into plain dataclasses.

The reference passes Hydra/OmegaConf ``DictConfig`` objects straight from the
checkpoint's ``hyper_parameters`` into ``MatchaTTSInfer(**hparams)``
(reference matcha/inference.py:44-55,186-193).  Here the same objects (DictConfig,
dict or SimpleNamespace, attribute- or item-addressed) are flattened once into
``PathHParams`` so that neither omegaconf nor hydra is needed on the GPU box.

Defaults follow the reference constructors:
  Decoder(...)            reference matcha/models/components/decoder.py:203-216
  configs/model/decoder/default.yaml (n_blocks=2, num_mid_blocks=2)
  TextEncoder / DurationPredictor   text_encoder.py:64-99,319-373
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Any, Tuple

N_VOCAB = 600  # reference matcha/text/symbols.py:29-39 (only this constant is consumed by the path)


def cfg_get(obj: Any, name: str, default: Any = None) -> Any:
    """Read ``name`` from a DictConfig / dict / namespace, attribute or item style."""
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get(name, default)
    if hasattr(obj, name):
        return getattr(obj, name)
    try:
        return obj[name]
    except Exception:
        return default


@dataclass
class EncoderHParams:
    n_feats: int = 100
    n_channels: int = 192
    filter_channels: int = 1152
    n_heads: int = 6
    n_layers: int = 4
    kernel_size: int = 5
    prenet: bool = True
    prenet_kernel_size: int = 3
    prenet_layers: int = 6          # hard-wired in the reference (text_encoder.py:343)
    # duration predictor
    dp_filter_channels: int = 96
    dp_kernel_size: int = 5
    dp_n_layers: int = 4


@dataclass
class DecoderHParams:
    channels: Tuple[int, ...] = (384, 384)
    attention_head_dim: int = 64
    n_blocks: int = 2
    num_mid_blocks: int = 2
    num_heads: int = 6


@dataclass
class PathHParams:
    n_spks: int = 1
    n_feats: int = 100
    spk_emb_dim: int = 96
    n_vocab: int = N_VOCAB
    mel_mean: float = 0.0
    mel_std: float = 1.0
    solver: str = "midpoint"
    use_mu_prior: bool = False
    sigma_min: float = 1e-4
    encoder: EncoderHParams = field(default_factory=EncoderHParams)
    decoder: DecoderHParams = field(default_factory=DecoderHParams)

    def to_dict(self) -> dict:
        d = asdict(self)
        d["decoder"]["channels"] = list(self.decoder.channels)
        return d

    # -- the nested objects the reference constructor expects ----------------
    def as_reference_kwargs(self) -> dict:
        """kwargs for ``MatchaTTSInfer(**kw)`` in the reference's own format."""
        from types import SimpleNamespace as NS
        e = self.encoder
        return dict(
            n_spks=self.n_spks,
            n_feats=self.n_feats,
            spk_emb_dim=self.spk_emb_dim,
            encoder=NS(
                encoder_params=NS(n_feats=e.n_feats, n_channels=e.n_channels, filter_channels=e.filter_channels,
                                  n_heads=e.n_heads, n_layers=e.n_layers, kernel_size=e.kernel_size,
                                  p_dropout=0.05, prenet=e.prenet, prenet_kernel_size=e.prenet_kernel_size),
                duration_predictor_params=NS(filter_channels_dp=e.dp_filter_channels, kernel_size=e.dp_kernel_size,
                                             p_dropout=0.05, n_layers=e.dp_n_layers),
            ),
            decoder=dict(channels=list(self.decoder.channels), dropout=0.05,
                         attention_head_dim=self.decoder.attention_head_dim, n_blocks=self.decoder.n_blocks,
                         num_mid_blocks=self.decoder.num_mid_blocks, num_heads=self.decoder.num_heads),
            cfm=NS(name="CFM", solver=self.solver, sigma_min=self.sigma_min, use_mu_prior=self.use_mu_prior),
            data_statistics=dict(mel_mean=self.mel_mean, mel_std=self.mel_std),
        )


def from_reference_kwargs(n_spks, n_feats, encoder, decoder, cfm, data_statistics, spk_emb_dim, **_) -> PathHParams:
    """Flatten the constructor arguments of reference ``MatchaTTSInfer.__init__`` (inference.py:45)."""
    ep = cfg_get(encoder, "encoder_params")
    dp = cfg_get(encoder, "duration_predictor_params")
    if not cfg_get(ep, "prenet", True):
        raise NotImplementedError("prenet=false is not used by any reference experiment config")
    for key in ("down_block_type", "mid_block_type", "up_block_type"):
        if cfg_get(decoder, key, "transformer") != "transformer":
            raise NotImplementedError("only transformer decoder blocks are on the path (SURVEY section 2, row 4)")
    stats = data_statistics or {}
    enc = EncoderHParams(
        n_feats=int(cfg_get(ep, "n_feats", n_feats)),
        n_channels=int(cfg_get(ep, "n_channels")),
        filter_channels=int(cfg_get(ep, "filter_channels")),
        n_heads=int(cfg_get(ep, "n_heads")),
        n_layers=int(cfg_get(ep, "n_layers")),
        kernel_size=int(cfg_get(ep, "kernel_size")),
        prenet=True,
        prenet_kernel_size=int(cfg_get(ep, "prenet_kernel_size")),
        dp_filter_channels=int(cfg_get(dp, "filter_channels_dp")),
        dp_kernel_size=int(cfg_get(dp, "kernel_size")),
        dp_n_layers=int(cfg_get(dp, "n_layers", 2)),
    )
    dec = DecoderHParams(
        channels=tuple(int(c) for c in cfg_get(decoder, "channels", (256, 256))),
        attention_head_dim=int(cfg_get(decoder, "attention_head_dim", 64)),
        n_blocks=int(cfg_get(decoder, "n_blocks", 1)),
        num_mid_blocks=int(cfg_get(decoder, "num_mid_blocks", 2)),
        num_heads=int(cfg_get(decoder, "num_heads", 4)),
    )
    return PathHParams(
        n_spks=int(n_spks), n_feats=int(n_feats), spk_emb_dim=int(spk_emb_dim),
        mel_mean=float(cfg_get(stats, "mel_mean", 0.0)), mel_std=float(cfg_get(stats, "mel_std", 1.0)),
        solver=str(cfg_get(cfm, "solver", "midpoint")),
        use_mu_prior=bool(cfg_get(cfm, "use_mu_prior", False)),
        sigma_min=float(cfg_get(cfm, "sigma_min", 1e-4)),
        encoder=enc, decoder=dec,
    )


def prod_v20(n_spks: int = 1) -> PathHParams:
    """The shipped architecture (reference configs/experiment/v20.yaml:17-63 over configs/model/*/default.yaml;
    data statistics configs/data/corpus-24k.yaml:28-30)."""
    return PathHParams(
        n_spks=n_spks, n_feats=100, spk_emb_dim=96,
        mel_mean=-4.684777, mel_std=6.512275, solver="euler", use_mu_prior=True,
        encoder=EncoderHParams(),
        decoder=DecoderHParams(),
    )


def tiny(n_spks: int = 2) -> PathHParams:
    """A small architecture for fast unit tests: same topology, narrow layers."""
    return PathHParams(
        n_spks=n_spks, n_feats=20, spk_emb_dim=16,
        mel_mean=-4.0, mel_std=2.0, solver="euler", use_mu_prior=True,
        encoder=EncoderHParams(n_feats=20, n_channels=32, filter_channels=96, n_heads=2, n_layers=2, kernel_size=5,
                               prenet_kernel_size=3, dp_filter_channels=32, dp_kernel_size=5, dp_n_layers=2),
        decoder=DecoderHParams(channels=(64, 64), attention_head_dim=32, n_blocks=1, num_mid_blocks=1, num_heads=2),
    )
