"""Data-parallel synthesis with the HIP path as the compute function: two ranks, one process each, rehearsed on ONE card
(both ranks use cuda:0; the gloo backend stages the two collectives through host memory, see dp.py).  The sharded result
must equal the single-process batch: same padded length (all_reduce MAX of the fine length), same slice of the batch-wide
noise, rank-ordered gather.  RCCL itself needs one device per rank and is exercised by `bench.py --gpus N` on a node."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
LENGTHS = [12, 5, 9, 12, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    import importlib
    hparams = importlib.import_module("matcha-tts-24k_amd.hparams")
    synthetic = importlib.import_module("matcha-tts-24k_amd.synthetic")
    inference = importlib.import_module("matcha-tts-24k_amd.inference")
    hp = hparams.tiny(n_spks=2)
    sd = synthetic.make_state_dict(hp, seed=7, duration_recipe=False)      # ragged durations: the MAX exchange matters
    m = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    m.decoder.solver = "midpoint"
    return hp, synthetic, m


def _worker(rank, world, port, n_utts, out_dir):
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHDYNAMO_DISABLE="1",
                      MTTS_CHAIN_PAIR="0")          # two ranks share the card here: the pair form needs the chip to itself
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    hp, synthetic, model = _model(dev)
    dp = importlib.import_module("matcha-tts-24k_amd.dp")
    x, x_len, spk = synthetic.make_inputs(hp, n_utts, 12, seed=77, lengths=LENGTHS[:n_utts])

    def synth(xs, ls, ss, sync_max, z_fn):
        out = model.synthesise(xs, ls, 2, speaker=ss, sync_max=sync_max, z=z_fn)
        return out["mel"], out["mel_lengths"]

    noise = lambda n, t: synthetic.cpu_noise((n, hp.n_feats, t)).to(dev)
    mel, lens = dp.synthesise_dp(synth, x.to(dev), x_len.to(dev), spk.to(dev), noise_fn=noise)
    if rank == 0:
        torch.save({"mel": mel.cpu(), "lens": lens.cpu()}, os.path.join(out_dir, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_utts", [5, 1])
def test_hip_dp_two_ranks_equal_single_process(tmp_path, n_utts):
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    mp.spawn(_worker, args=(2, _free_port(), n_utts, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "dp.pt")
    dev = torch.device("cuda", 0)
    hp, synthetic, model = _model(dev)
    x, x_len, spk = synthetic.make_inputs(hp, n_utts, 12, seed=77, lengths=LENGTHS[:n_utts])
    ref = model.synthesise(x.to(dev), x_len.to(dev), 2, speaker=spk.to(dev),
                           z=lambda t: synthetic.cpu_noise((n_utts, hp.n_feats, t)).to(dev))
    assert torch.equal(got["lens"], ref["mel_lengths"].cpu())
    t = ref["mel"].shape[-1]
    assert got["mel"].shape[0] == n_utts and got["mel"].shape[-1] >= t
    # same kernels on the same rows; a shard's smaller batch may pick another tile shape => rounding-level differences only
    assert (got["mel"][:, :, :t] - ref["mel"].cpu()).abs().max() < 5e-5
