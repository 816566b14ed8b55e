"""Utterance-level data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-GPU path; utterances are independent (per-sample norms and
attention), so the path shards with NO collective on the compute path.  Two exchanges exist:
  * one 8-byte all_reduce(MAX) of the batch-wide fine length, because the reference pads the decoder to the
    batch maximum (inference.py:146-148) and GroupNorm / attention / the noise draw depend on that length --
    without it a shard would not reproduce the single-GPU result;
  * one all_gather of the finished mels (+ lengths).
The same code runs on CPU tensors with the gloo backend (tests/test_dp_gloo.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_slice(n: int, world: int, rank: int) -> slice:
    """Contiguous, balanced shard of n utterances (the first n % world ranks get one more)."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return slice(start, start + q + (1 if rank < r else 0))


def all_reduce_max_int(v: int, device) -> int:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v
    t = torch.tensor([v], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def all_gather_mels(mel: torch.Tensor, world: int) -> torch.Tensor:
    """Equal-shaped shards [b, n_feats, T] -> [b*world, n_feats, T] in rank order (one all_gather_into_tensor)."""
    if world == 1:
        return mel
    mel = mel.contiguous()
    out = torch.empty((mel.shape[0] * world,) + tuple(mel.shape[1:]), dtype=mel.dtype, device=mel.device)
    dist.all_gather_into_tensor(out, mel)
    return out


def all_gather_ragged(mel: torch.Tensor, lengths: torch.Tensor, n_total: int, world: int, rank: int
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Shards of possibly different batch size / frame count -> ([n_total, n_feats, T_max], lengths[n_total])."""
    if world == 1:
        return mel, lengths
    per = max(shard_slice(n_total, world, r).stop - shard_slice(n_total, world, r).start for r in range(world))
    t_max = all_reduce_max_int(int(mel.shape[-1]), mel.device)
    pad = torch.zeros(per, mel.shape[1], t_max, dtype=mel.dtype, device=mel.device)
    pad[: mel.shape[0], :, : mel.shape[-1]] = mel
    plen = torch.zeros(per, dtype=torch.int64, device=mel.device)
    plen[: lengths.shape[0]] = lengths
    g_mel = torch.empty(per * world, mel.shape[1], t_max, dtype=mel.dtype, device=mel.device)
    g_len = torch.empty(per * world, dtype=torch.int64, device=mel.device)
    dist.all_gather_into_tensor(g_mel, pad)
    dist.all_gather_into_tensor(g_len, plen)
    keep = []
    for r in range(world):
        s = shard_slice(n_total, world, r)
        keep.extend(range(r * per, r * per + (s.stop - s.start)))
    idx = torch.tensor(keep, dtype=torch.long, device=mel.device)
    return g_mel.index_select(0, idx), g_len.index_select(0, idx)


def synthesise_dp(synth_fn: Callable, x: torch.Tensor, x_lengths: torch.Tensor, speakers: torch.Tensor,
                  noise_fn: Optional[Callable] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run ``synth_fn`` on this rank's shard of (x, x_lengths, speakers) and gather every rank's mel.

    ``synth_fn(x, x_lengths, speakers, sync_max, z_fn) -> (mel [b, n_feats, T], mel_lengths [b])`` must call
    ``sync_max(local_max_fine_length)`` to obtain the batch-wide maximum before choosing T_pad, and, when ``z_fn``
    is given, ``z_fn(T_pad)`` for its rows of the batch-wide noise tensor (drawn identically on every rank and sliced,
    so that the sharded result equals the single-process batch).
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = x.shape[0]
    sl = shard_slice(n, world, rank)
    dev = x.device
    z_fn = None
    if noise_fn is not None:
        z_fn = lambda t_pad: noise_fn(n, t_pad)[sl]
    mel, lens = synth_fn(x[sl], x_lengths[sl], speakers[sl], lambda m: all_reduce_max_int(int(m), dev), z_fn)
    return all_gather_ragged(mel, lens, n, world, rank)
