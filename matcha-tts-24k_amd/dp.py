"""Utterance-level data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-GPU path; utterances are independent (per-sample norms and
attention), so the path shards with NO collective on the compute path.  Two exchanges exist:
  * one 8-byte all_reduce(MAX) of the batch-wide fine length, because the reference pads the decoder to the
    batch maximum (inference.py:146-148) and GroupNorm / attention / the noise draw depend on that length --
    without it a shard would not reproduce the single-GPU result;
  * one all_gather of the finished mels (+ lengths).
The same code runs on CPU tensors with the gloo backend (tests/test_dp_gloo.py), and on device tensors with the gloo
backend (collectives staged through host memory: a rehearsal of N ranks on ONE card, tests/test_hip_dp.py).
A rank whose shard is empty (fewer utterances than ranks) still takes part in every collective.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# At world size 1 every exchange is the identity and is skipped -- unless this is set: tests/test_hip_rccl.py runs the collectives
# themselves (RCCL on device tensors) with the one GPU a test box has.
FORCE_COLLECTIVES = False


def _single(world: int) -> bool:
    return world == 1 and not FORCE_COLLECTIVES


def shard_slice(n: int, world: int, rank: int) -> slice:
    """Contiguous, balanced shard of n utterances (the first n % world ranks get one more)."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return slice(start, start + q + (1 if rank < r else 0))


def _host_staged() -> bool:
    """gloo moves host memory: device tensors are staged through the CPU (rehearsals only; RCCL takes device pointers)."""
    return dist.get_backend() == "gloo"


def all_reduce_max_ints(vals: Sequence[int], device) -> List[int]:
    if not (dist.is_available() and dist.is_initialized()) or _single(dist.get_world_size()):
        return [int(v) for v in vals]
    t = torch.tensor([int(v) for v in vals], dtype=torch.int64, device="cpu" if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.tolist()]


def all_reduce_max_int(v: int, device) -> int:
    return all_reduce_max_ints([v], device)[0]


def _all_gather_rows(t: torch.Tensor, world: int) -> torch.Tensor:
    """[b, ...] on every rank -> [b*world, ...] in rank order."""
    t = t.contiguous()
    if _host_staged() and t.is_cuda:
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return torch.cat(parts, 0).to(t.device)
    out = torch.empty((t.shape[0] * world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if _host_staged():
        parts = list(out.chunk(world, 0))
        dist.all_gather(parts, t)
        return out
    dist.all_gather_into_tensor(out, t)
    return out


def all_gather_mels(mel: torch.Tensor, world: int) -> torch.Tensor:
    """Equal-shaped shards [b, n_feats, T] -> [b*world, n_feats, T] in rank order (one all_gather_into_tensor)."""
    if _single(world):
        return mel
    return _all_gather_rows(mel, world)


def all_gather_ragged(mel: torch.Tensor, lengths: torch.Tensor, n_total: int, world: int, rank: int
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Shards of possibly different batch size / frame count -> ([n_total, n_feats, T_max], lengths[n_total])."""
    if _single(world):
        return mel, lengths
    per = max(shard_slice(n_total, world, r).stop - shard_slice(n_total, world, r).start for r in range(world))
    # an empty shard knows neither the frame count nor n_feats: both come from the ranks that have rows
    t_max, n_feats = all_reduce_max_ints([mel.shape[-1] if mel.shape[0] else 0, mel.shape[1] if mel.shape[0] else 0], mel.device)
    pad = torch.zeros(per, n_feats, t_max, dtype=mel.dtype, device=mel.device)
    if mel.shape[0]:
        pad[: mel.shape[0], :, : mel.shape[-1]] = mel
    plen = torch.zeros(per, dtype=torch.int64, device=mel.device)
    plen[: lengths.shape[0]] = lengths
    g_mel = _all_gather_rows(pad, world)
    g_len = _all_gather_rows(plen, world)
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
    sync_max = lambda m: all_reduce_max_int(int(m), dev)
    if sl.stop > sl.start:
        mel, lens = synth_fn(x[sl], x_lengths[sl], speakers[sl], sync_max, z_fn)
    else:           # fewer utterances than ranks: nothing to synthesise here, but the peers are waiting in the collectives
        sync_max(0)
        mel = torch.zeros(0, 0, 0, dtype=torch.float32, device=dev)
        lens = torch.zeros(0, dtype=torch.int64, device=dev)
    return all_gather_ragged(mel, lens, n, world, rank)
