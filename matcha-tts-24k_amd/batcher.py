"""Dynamic batching in front of ``MatchaTTSInfer.synthesise``.

The reference server handles one request at a time on the event-loop thread (server.py:93-127).  On an MI355X one request
uses a few percent of the chip (batch sweep), so a serving process wants to run whatever is waiting as
ONE ragged batch -- without changing any request's audio.  ``per_request_padding`` (inference.py) makes that exact: each row
of the batch equals the batch-of-one result (to rounding: tile shapes, hence summation order, depend on the grid).

``FrameBudgetBatcher`` is the queue + grouping policy:
  * requests are grouped by what must be uniform inside one call (solver, n_timesteps); voices, their duration scale
    corrections and client speeds are per-utterance inputs (mtts_durations_per_utterance);
  * a batch takes the oldest waiting request and then the waiting requests of the same group that are closest to it in token
    count (padding wastes MFMA work: the estimator's cost is ~linear in padded frames), up to ``max_batch`` utterances and
    ``max_tokens`` padded tokens (B * longest), the frame budget idea of the reference's training sampler
    (text_mel_datamodule.py:111-154) applied to inference;
  * one worker thread drives the model (the HIP context is not re-entrant: model.decoder.solver is per-call state).

The class is transport-agnostic: an HTTP handler submits and awaits the future (``submit(...).result()``), see INTEGRATION.md.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import torch


@dataclass
class Request:
    ids: Sequence[int]                      # phoneme ids of one utterance
    speaker: int = 0
    voice_mix: Optional[Sequence[Tuple[int, float]]] = None    # [(id, weight), ...] as reference server.py:96-101; overrides speaker
    solver: str = "midpoint"
    n_timesteps: int = 4
    scale_correction: float = 1.0
    length_scale: float = 1.0
    future: Future = field(default_factory=Future, repr=False)
    t_submit: float = field(default_factory=time.monotonic, repr=False)

    @property
    def group(self) -> Tuple[Any, ...]:
        return (self.solver, int(self.n_timesteps))


def plan_batch(waiting: List[Request], max_batch: int, max_tokens: int) -> List[int]:
    """Indices (into ``waiting``, which is in arrival order) of the next batch.  Pure function: unit-tested on the CPU."""
    if not waiting:
        return []
    head = waiting[0]
    n0 = len(head.ids)
    same = [i for i, r in enumerate(waiting) if i > 0 and r.group == head.group]
    same.sort(key=lambda i: (abs(len(waiting[i].ids) - n0), i))     # nearest in length first, then oldest
    chosen, longest = [0], n0
    for i in same:
        if len(chosen) >= max_batch:
            break
        cand = max(longest, len(waiting[i].ids))
        if cand * (len(chosen) + 1) > max_tokens:
            continue
        chosen.append(i)
        longest = cand
    return sorted(chosen)


class FrameBudgetBatcher:
    """``submit()`` from any thread; results arrive on the request's future as ``{"mel": [n_feats, T_b], "mel_length": T_b}``."""

    def __init__(self, model, max_batch: int = 32, max_tokens: int = 8192, max_wait_ms: float = 2.0,
                 run_batch: Optional[Callable[[List[Request]], List[Dict[str, Any]]]] = None, vocoder=None):
        """``vocoder``: a ``load_vocoder("vocos")`` object; results then also carry ``"audio"`` = the reference handler's
        ``trim_trailing_silence(to_waveform(mel, vocoder))`` (reference inference.py:246, server.py:116), computed per request on
        its exact-length mel (the vocoder's k7 convolutions would otherwise see a neighbour-dependent padded tail)."""
        self.model = model
        self.vocoder = vocoder
        self.max_batch = int(max_batch)
        self.max_tokens = int(max_tokens)
        self.max_wait = float(max_wait_ms) / 1e3
        self._run = run_batch or self._run_on_model
        self._waiting: List[Request] = []
        self._cv = threading.Condition()
        self._stop = False
        self.batches_run = 0
        self.busy_s = 0.0                   # wall time the worker spent inside batches (device work + its host side): a load gauge
        self._thread = threading.Thread(target=self._loop, name="mtts-batcher", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ producer side
    def submit(self, ids: Sequence[int], **kw) -> Future:
        if len(ids) == 0:
            raise ValueError("empty utterance")
        if len(ids) > self.max_tokens:
            raise ValueError(f"utterance of {len(ids)} tokens exceeds the batch budget of {self.max_tokens}")
        r = Request(ids=list(ids), **kw)
        with self._cv:
            if self._stop:
                raise RuntimeError("batcher is closed")
            self._waiting.append(r)
            self._cv.notify()
        return r.future

    def close(self) -> None:
        with self._cv:
            self._stop = True
            self._cv.notify()
        self._thread.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ worker
    def _loop(self) -> None:
        while True:
            with self._cv:
                while not self._waiting and not self._stop:
                    self._cv.wait()
                if self._stop and not self._waiting:
                    return
                # give concurrent submitters a moment to arrive, bounded by the oldest request's age
                deadline = self._waiting[0].t_submit + self.max_wait
                while len(self._waiting) < self.max_batch and not self._stop:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        break
                    self._cv.wait(left)
                take = plan_batch(self._waiting, self.max_batch, self.max_tokens)
                batch = [self._waiting[i] for i in take]
                for i in reversed(take):
                    del self._waiting[i]
            t_run = time.monotonic()
            try:
                results = self._run(batch)
                self.busy_s += time.monotonic() - t_run
                for r, res in zip(batch, results):
                    r.future.set_result(res)
            except BaseException as e:  # noqa: BLE001 - every waiter must be released
                for r in batch:
                    if not r.future.done():
                        r.future.set_exception(e)
            self.batches_run += 1

    def _run_on_model(self, batch: List[Request]) -> List[Dict[str, Any]]:
        dev = next(iter(self.model.state_dict().values())).device       # where load_matcha / .to() put the model
        B, n_max = len(batch), max(len(r.ids) for r in batch)
        x = torch.zeros(B, n_max, dtype=torch.long)
        for b, r in enumerate(batch):
            x[b, :len(r.ids)] = torch.as_tensor(r.ids, dtype=torch.long)
        x_len = torch.tensor([len(r.ids) for r in batch], dtype=torch.long)
        head = batch[0]
        self.model.decoder.solver = head.solver
        emb = self.model.speaker_rows([list(r.voice_mix) if r.voice_mix is not None else r.speaker for r in batch])
        out = self.model.synthesise(x.to(dev), x_len.to(dev), head.n_timesteps, speaker_embeddings=emb,
                                    scale_correction=[r.scale_correction for r in batch],
                                    length_scale=[r.length_scale for r in batch], per_request_padding=True)
        lens = out["mel_lengths"].tolist()
        res = [{"mel": out["mel"][b, :, :int(lens[b])], "mel_length": int(lens[b])} for b in range(B)]
        if self.vocoder is not None:
            from .inference import _waveform_on_device, trim_trailing_silence
            for r in res:
                r["audio"] = trim_trailing_silence(_waveform_on_device(r["mel"][None], self.vocoder).squeeze()).cpu()
        return res
