"""Host-side mirror of the reference's model objects for the synthesis path.

The reference classes (TextEncoder text_encoder.py:319, Decoder decoder.py:202, CFM flow_matching.py:110)
are ``nn.Module`` trees whose ``forward`` runs PyTorch ops.  Here the trees only *hold* the parameters, under
exactly the reference's state-dict names (generated from ``synthetic.state_dict_spec``), and ``forward`` calls
the HIP library through one shared ``HipModel``.  There is no PyTorch arithmetic and no CPU fallback.
"""
from __future__ import annotations

import collections
import os
from typing import Optional

import torch
import torch.nn as nn

from ._hip import HipModel
from .hparams import PathHParams
from .synthetic import state_dict_spec

_BUFFER_KINDS = ("mel_mean", "mel_std")


class ParamTree(nn.Module):
    """A bare container node; children and parameters are attached by dotted state-dict key."""

    def node(self, name: str) -> "ParamTree":
        if name not in self._modules:
            self.add_module(name, ParamTree())
        return self._modules[name]

    def attach(self, key: str, shape, kind: str) -> None:
        parts = key.split(".")
        n = self
        for p in parts[:-1]:
            n = n.node(p)
        t = torch.zeros(shape, dtype=torch.float32)
        if kind in _BUFFER_KINDS:
            n.register_buffer(parts[-1], t)
        else:
            n.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


class Runtime:
    """Shared by every module of one model: the HIP context and the lazily refreshed packed weights."""

    def __init__(self, hp: PathHParams, owner: nn.Module):
        self.hp = hp
        self.owner = owner          # root module whose state_dict is the source of truth
        self.hip: Optional[HipModel] = None
        self.wide: Optional[HipModel] = None      # the same weights on the full-range arithmetic (range guard fallback)
        self.use_wide = False
        self.dirty = True
        self.mel_mean, self.mel_std = hp.mel_mean, hp.mel_std
        self.cache_dir = None       # a converted checkpoint's directory: the packed weight image is cached there (checkpoint.packed_cache)

    def _load(self, hip: HipModel) -> None:
        p = next(self.owner.parameters())
        if not p.is_cuda:
            raise RuntimeError("matcha-tts-24k_amd: the model must be on a HIP device (model.to('cuda')); "
                               "there is no CPU path")
        sd = self.owner.state_dict()
        hip.load_state_dict(sd, p.device, cache_dir=self.cache_dir)
        # denormalisation constants come from the checkpoint's buffers (reference inference.py:54-55,172)
        self.mel_mean, self.mel_std = float(sd["mel_mean"]), float(sd["mel_std"])

    def ready(self) -> HipModel:
        if self.hip is None:
            self.hip = HipModel(self.hp)
        if self.dirty:
            self._load(self.hip)
            self.wide, self.use_wide = None, False       # new weights: back to the default arithmetic until it overflows
            self.dirty = False
        if self.use_wide:
            if self.wide is None:
                # three exact bf16 terms per operand: the fp32 exponent range, twice the MFMA products of the default
                self.wide = HipModel(self.hp, terms=6)
                self._load(self.wide)
            return self.wide
        return self.hip


class _PathModule(ParamTree):
    """Base of the callable modules: keeps a (non-registered) handle on the shared runtime."""

    def _bind(self, rt: Runtime) -> None:
        object.__setattr__(self, "_rt", rt)


class TextEncoder(_PathModule):
    """``encoder(x, x_lengths, e_enc, e_dur) -> (mu_x, logw, x_mask)`` -- reference text_encoder.py:375-406."""

    def forward(self, x, x_lengths, speaker_embedding_enc, speaker_embedding_dur):
        if x.shape[1] > 4000:   # the reference's RoPE cache limit (text_encoder.py:138,164)
            raise AssertionError("Phonetic representation too long, exceeds RoPE cache size 4000")
        return self._rt.ready().text_encoder(x, x_lengths, speaker_embedding_enc, speaker_embedding_dur)


class Estimator(_PathModule):
    """``estimator(x, mask, mu, t) -> velocity`` -- reference decoder.py:359-426."""

    def forward(self, x, mask, mu, t):
        return self._rt.ready().decoder_forward(x, mask, mu, float(t))


class CFM(nn.Module):
    """``decoder(mu, mask, n_timesteps)`` / ``.solve(x, t_span, mu, mask)`` -- reference flow_matching.py:25-63,110-117.

    ``solver`` is a plain attribute that callers overwrite per request (reference cli.py:94, server.py:43,109).
    ``estimator`` may be re-assigned with a ``torch.compile`` wrapper (reference server.py:47); the wrapper is
    unwrapped, because the estimator already is a fixed sequence of HIP launches.
    """

    def __init__(self, hp: PathHParams, rt: Runtime):
        super().__init__()
        self.n_feats = 2 * hp.n_feats
        self.solver = hp.solver
        self.sigma_min = hp.sigma_min
        self.use_mu_prior = hp.use_mu_prior
        self.fold_padding = os.environ.get("MTTS_FOLD", "1") != "0"
        self.fold_align = int(os.environ.get("MTTS_FOLD_ALIGN", "1"))
        # HIP graphs for launch-bound sizes (SURVEY section 7 "launch-bound small batches"): "auto" = when the estimator holds
        # at most graph_max_rows rows (B * rows per utterance), "1" always, "0" never
        self.graph_mode = os.environ.get("MTTS_GRAPH", "auto")
        self.graph_max_rows = int(os.environ.get("MTTS_GRAPH_MAX_ROWS", "6144"))
        self.graph_bucket = 64
        object.__setattr__(self, "_graphs", collections.OrderedDict())
        self.graph_cache_entries = int(os.environ.get("MTTS_GRAPH_CACHE", "24"))
        self.graph_cache_bytes = int(float(os.environ.get("MTTS_GRAPH_CACHE_GB", "4")) * (1 << 30))   # static buffers held
        self.graph_replays = 0
        object.__setattr__(self, "_rt", rt)
        self.estimator = Estimator()
        self.estimator._bind(rt)

    def __setattr__(self, name, value):
        if name == "estimator" and hasattr(value, "_orig_mod"):
            value = value._orig_mod
        super().__setattr__(name, value)

    def noise(self, like: torch.Tensor) -> torch.Tensor:
        """The seed-42 draw of flow_matching.py:43-55 on ``like``'s device generator."""
        g = torch.Generator(device=like.device)
        g.manual_seed(42)
        return torch.randn(like.shape, generator=g, dtype=like.dtype, device=like.device)

    def noise_per_request(self, like: torch.Tensor, t_len) -> torch.Tensor:
        """Per-request padding: utterance b gets the seed-42 draw of shape [1, n_feats, t_len[b]] a batch-of-one call would
        make (the draw depends on the shape), zero beyond it."""
        z = torch.zeros_like(like)
        for b, t in enumerate(t_len):
            z[b:b + 1, :, :t] = self.noise(like[b:b + 1, :, :t])
        return z

    def fold_plan(self, T: int, y_max: Optional[int]) -> Optional[int]:
        """Rows per utterance for the folded estimator (include/mtts.h mtts_cfm_solve_folded), or None to run all T frames.
        ``fold_padding`` (attribute; env MTTS_FOLD=0 turns the default off) and ``fold_align`` (MTTS_FOLD_ALIGN, rows at the coarsest
        level; default 1 = the minimum, 322 / 161 rows for 320 valid frames: no kernel needs whole tiles per utterance, and
        every extra row is pure cost -- 29.1 / 29.4 / 29.8 / 33.5 ms per step at align 2 / 4 / 8 / 32 on one box) are plain attributes like ``solver``."""
        if not self.fold_padding or y_max is None:
            return None
        t_fold = self._rt.ready().fold_rows(y_max, self.fold_align)
        return t_fold if t_fold < T else None

    @torch.inference_mode()
    def forward(self, mu, mask, n_timesteps, z: Optional[torch.Tensor] = None, t_out: Optional[int] = None,
                out_scale: float = 1.0, out_shift: float = 0.0, t_len=None, y_lengths=None, y_max: Optional[int] = None):
        """``z``: optional explicit noise (e.g. the CPU-generator stream for parity with the CPU reference).
        ``t_len``: per-utterance padded lengths (list of even ints) for per-request padding (mtts_set_frame_limits).
        ``y_lengths`` (LongTensor[B] on the device) + ``y_max`` (their maximum, on the host): ``mask`` is the prefix mask of
        these lengths, which lets the estimator fold the padded frames (``fold_plan``); results equal the unfolded call."""
        hip = self._rt.ready()
        if z is None:
            z = self.noise(mu) if t_len is None else self.noise_per_request(mu, t_len)
        t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
        kw = dict(add_mu=self.use_mu_prior, t_out=t_out, out_scale=out_scale, out_shift=out_shift)
        if y_lengths is not None and y_max is not None and self._graph_wanted(mu.shape[0], y_max):
            return self._solve_on_graph(hip, mu, z, n_timesteps, t_out, out_scale, out_shift, t_len, y_lengths, y_max)
        t_fold = self.fold_plan(mu.shape[-1], y_max) if y_lengths is not None else None
        if t_fold is not None and (t_out is None or t_out <= t_fold):
            kw.update(y_lengths=y_lengths, y_max=y_max, t_fold=t_fold)
        if t_len is None:
            return hip.cfm_solve(z, mu, mask, t_span, self.solver, **kw)
        hip.set_frame_limits(torch.tensor(list(t_len), dtype=torch.int32, device=mu.device))
        try:
            return hip.cfm_solve(z, mu, mask, t_span, self.solver, **kw)
        finally:
            hip.set_frame_limits(None)

    # ------------------------------------------------------------------ HIP graphs
    def _graph_rows(self, y_max: int) -> int:
        """Rows per utterance of the graph that serves valid lengths up to y_max: the folded row count, rounded up to a bucket
        so that requests of similar length replay the same graph.  Filler rows change nothing (include/mtts.h
        mtts_cfm_solve_folded): the reference's padded length reaches the kernels as DATA (mtts_set_frame_limits), not as a shape."""
        rows = self._rt.ready().fold_rows(y_max, self.fold_align)
        return (rows + self.graph_bucket - 1) // self.graph_bucket * self.graph_bucket

    def _graph_wanted(self, B: int, y_max: int) -> bool:
        if self.graph_mode == "0" or not self.fold_padding:
            return False
        # auto: few utterances (the host's launch cost is what a graph saves; the kernels take the same time) and few enough
        # (B, bucket) combinations that captures (~15 ms each, first use only) stay rare.  The row bucket pads each utterance
        # (322 -> 384 rows): free at B <= 4 (14.7 ms per step either way at B = 1), 6 % of GPU time at B = 8 (16.6 vs 15.7 ms,
        # profiles/r02d_small_batch.log) -- beyond four utterances the direct launches are the faster path
        return self.graph_mode == "1" or (B <= 4 and B * self._graph_rows(y_max) <= self.graph_max_rows)

    def _solve_on_graph(self, hip, mu, z, n_timesteps, t_out, out_scale, out_shift, t_len, y_lengths, y_max):
        """One ODE solve as ONE graph launch: the ~100 kernels per evaluation are captured once per (batch, row bucket, solver,
        steps) with static input / scratch / output buffers and replayed; per call only the inputs are copied in."""
        B, nf, T = mu.shape
        rows = self._graph_rows(y_max)
        f = 2 ** (len(self._rt.hp.decoder.channels) - 1)
        y_cap = (rows // f - 1) * f                       # the longest valid length these rows can hold
        t_src = 2 * rows                                  # source row stride: >= any T_pad whose valid length fits
        # A captured graph bakes device pointers into its kernel nodes: the context's packed weights above all.  The key names
        # the context AND the generation of its weights (HipModel.load_state_dict bumps it), and graphs of an older generation
        # are dropped before the lookup: after model.load_state_dict / .to() they would read freed memory.
        gen = (hip.uid, hip.generation)
        for k in [k for k in self._graphs if k[0] == gen[0] and k[1] != gen[1]]:
            del self._graphs[k]
        key = gen + (B, rows, self.solver, int(n_timesteps), float(out_scale), float(out_shift), bool(self.use_mu_prior), hip.gemm_terms())
        e = self._graphs.get(key)
        if e is None:
            dev = mu.device
            e = {"z": torch.zeros(B, nf, t_src, device=dev), "mu": torch.zeros(B, nf, t_src, device=dev),
                 "ylen": torch.ones(B, dtype=torch.int64, device=dev), "tlen": torch.full((B,), 2, dtype=torch.int32, device=dev),
                 "ws": torch.empty(hip.decoder_workspace_bytes(B, rows), dtype=torch.uint8, device=dev)}
            t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
            kw = dict(add_mu=self.use_mu_prior, t_out=rows, out_scale=out_scale, out_shift=out_shift, y_lengths=e["ylen"],
                      y_max=y_cap, t_fold=rows, ws=e["ws"])
            hip.set_frame_limits(e["tlen"])
            try:
                hip.cfm_solve(e["z"], e["mu"], None, t_span, self.solver, **kw)      # warm-up: lazy kernel attributes are set here
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    e["out"] = hip.cfm_solve(e["z"], e["mu"], None, t_span, self.solver, **kw)
                e["graph"] = g
            finally:
                hip.set_frame_limits(None)
            e["bytes"] = sum(int(e[k].numel()) * e[k].element_size() for k in ("z", "mu", "ws", "out"))
            self._graphs[key] = e
            while len(self._graphs) > 1 and (len(self._graphs) > self.graph_cache_entries or
                                             sum(v["bytes"] for v in self._graphs.values()) > self.graph_cache_bytes):
                self._graphs.popitem(last=False)          # least recently used graph and its buffers
        else:
            self._graphs.move_to_end(key)
        w = min(T, rows)
        e["z"][:, :, :w].copy_(z[:, :, :w])
        e["mu"][:, :, :w].copy_(mu[:, :, :w])
        e["ylen"].copy_(y_lengths)
        if t_len is None:
            e["tlen"].fill_(T)                              # the reference's batch-wide padded length
        else:
            e["tlen"].copy_(torch.as_tensor(list(t_len), dtype=torch.int32))
        e["graph"].replay()
        hip.note_workspace("dec", e["ws"])
        self.graph_replays += 1
        n_out = rows if t_out is None else int(t_out)
        return e["out"][:, :, :n_out].clone()

    def solve(self, x, t_span, mu, mask):
        return self._rt.ready().cfm_solve(x, mu, mask, t_span, self.solver)

    def solve_euler(self, x, t_span, mu, mask):
        """Upstream Matcha-TTS name for the same loop with the euler solver."""
        return self._rt.ready().cfm_solve(x, mu, mask, t_span, "euler")


def build_trees(hp: PathHParams, root: nn.Module, rt: Runtime) -> None:
    """Attach every parameter of the path to ``root`` under the reference's names."""
    enc, cfm, est = TextEncoder(), CFM(hp, rt), None
    enc._bind(rt)
    est = cfm.estimator
    root.encoder = enc
    root.decoder = cfm
    root.speaker_embeddings_enc = ParamTree()
    root.speaker_embeddings_dur = ParamTree()
    for key, shape, kind in state_dict_spec(hp):
        if key.startswith("encoder."):
            enc.attach(key[len("encoder."):], shape, kind)
        elif key.startswith("decoder.estimator."):
            est.attach(key[len("decoder.estimator."):], shape, kind)
        elif key.startswith("speaker_embeddings_enc."):
            root.speaker_embeddings_enc.attach(key.split(".", 1)[1], shape, kind)
        elif key.startswith("speaker_embeddings_dur."):
            root.speaker_embeddings_dur.attach(key.split(".", 1)[1], shape, kind)
        elif kind in _BUFFER_KINDS:
            root.register_buffer(key, torch.zeros(shape, dtype=torch.float32))
        else:
            raise KeyError(key)
