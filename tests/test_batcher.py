"""Dynamic batcher: grouping policy on the CPU, exactness of batched results on the GPU."""
import threading

import pytest
import torch

from conftest import sub


def _req(bt, n, **kw):
    return bt.Request(ids=list(range(1, n + 1)), **kw)


def test_plan_batch_groups_by_call_parameters_and_length():
    bt = sub("batcher")
    w = [_req(bt, 100), _req(bt, 400), _req(bt, 90, solver="euler"), _req(bt, 110), _req(bt, 30), _req(bt, 95, n_timesteps=8)]
    # oldest first, then same-group requests nearest in length; other solvers / step counts wait for their own batch
    assert bt.plan_batch(w, max_batch=3, max_tokens=10_000) == [0, 3, 4]
    assert bt.plan_batch(w, max_batch=8, max_tokens=10_000) == [0, 1, 3, 4]
    # token budget = B * longest: adding the 400-token request to three others would cost 1600
    assert bt.plan_batch(w, max_batch=8, max_tokens=500) == [0, 3, 4]
    assert bt.plan_batch(w[1:2], max_batch=8, max_tokens=500) == [0]
    assert bt.plan_batch([], 8, 500) == []


def test_batcher_queue_with_a_stub_model():
    """Threads submit concurrently; every future resolves with its own result; batches respect the limits."""
    bt = sub("batcher")
    seen = []

    def run(batch):
        seen.append([len(r.ids) for r in batch])
        assert len({r.group for r in batch}) == 1
        return [{"mel_length": 2 * len(r.ids), "speaker": r.speaker} for r in batch]

    with bt.FrameBudgetBatcher(model=None, max_batch=4, max_tokens=64, max_wait_ms=20.0, run_batch=run) as q:
        futs = {}

        def client(i):
            futs[i] = q.submit(list(range(1, 4 + i)), speaker=i, solver="euler" if i % 2 else "midpoint")

        ts = [threading.Thread(target=client, args=(i,)) for i in range(10)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i, f in futs.items():
            r = f.result(timeout=10)
            assert r == {"mel_length": 2 * (3 + i), "speaker": i}
        with pytest.raises(ValueError):
            q.submit([])
        with pytest.raises(ValueError):
            q.submit(list(range(100)))
    assert sum(len(b) for b in seen) == 10
    assert all(len(b) <= 4 and len(b) * max(b) <= 64 for b in seen)
    assert len(seen) < 10                      # something was actually batched


def test_batcher_propagates_errors_to_every_waiter():
    bt = sub("batcher")

    def run(batch):
        raise RuntimeError("device lost")

    with bt.FrameBudgetBatcher(model=None, max_batch=4, max_wait_ms=5.0, run_batch=run) as q:
        fs = [q.submit([1, 2, 3]) for _ in range(3)]
        for f in fs:
            with pytest.raises(RuntimeError, match="device lost"):
                f.result(timeout=10)


@pytest.mark.gpu
def test_batched_requests_equal_individual_calls():
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    hparams, synthetic, inf, bt = sub("hparams"), sub("synthetic"), sub("inference"), sub("batcher")
    dev = torch.device("cuda")
    hp = hparams.tiny(n_spks=3)
    model = inf.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(synthetic.make_state_dict(hp, seed=7), strict=True)
    model = model.to(dev).eval()
    lengths = [12, 7, 3, 10, 5, 1]
    x, _, spk = synthetic.make_inputs(hp, len(lengths), max(lengths), seed=5, lengths=lengths)
    with bt.FrameBudgetBatcher(model, max_batch=4, max_tokens=64, max_wait_ms=50.0) as q:
        futs = [q.submit(x[b, :n].tolist(), speaker=int(spk[b]), solver="midpoint", n_timesteps=2) for b, n in enumerate(lengths)]
        res = [f.result(timeout=120) for f in futs]
        assert q.batches_run < len(lengths)
    model.decoder.solver = "midpoint"
    for b, n in enumerate(lengths):
        alone = model.synthesise(x[b:b + 1, :n].to(dev), torch.tensor([n], device=dev), 2, speaker=spk[b:b + 1].to(dev))
        t = int(alone["mel_lengths"][0])
        assert res[b]["mel_length"] == t
        assert (res[b]["mel"] - alone["mel"][0, :, :t]).abs().max().item() < 5e-5      # equal up to tile-shape summation order
