"""RCCL on the hardware a test box has: ONE rank, backend "nccl" (= RCCL on ROCm), device tensors.  world_size 1 makes every
collective the identity, so `dp.FORCE_COLLECTIVES` keeps dp.py from skipping them: init_process_group with a device id, the
int64 all_reduce(MAX) of the fine length, all_gather_into_tensor of the mels and the full synthesise_dp flow run through the
RCCL code path in a fresh child process and tear down cleanly.  Multi-rank RCCL over xGMI needs one device per rank: the
driver's `bench.py --gpus N` run."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, out_dir):
    import importlib
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHDYNAMO_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    dp = importlib.import_module("matcha-tts-24k_amd.dp")
    dp.FORCE_COLLECTIVES = True
    res = {}
    # the two primitives, on device tensors
    res["max"] = dp.all_reduce_max_ints([7, 640, 3], dev)
    mel = torch.arange(2 * 5 * 11, dtype=torch.float32, device=dev).reshape(2, 5, 11)
    res["gather_equal"] = bool(torch.equal(dp.all_gather_mels(mel, 1), mel))
    g_mel, g_len = dp.all_gather_ragged(mel, torch.tensor([11, 4], device=dev), 2, 1, 0)
    res["ragged_equal"] = bool(torch.equal(g_mel, mel)) and g_len.tolist() == [11, 4]
    # the whole data-parallel flow with the HIP path as the compute function
    hparams = importlib.import_module("matcha-tts-24k_amd.hparams")
    synthetic = importlib.import_module("matcha-tts-24k_amd.synthetic")
    inference = importlib.import_module("matcha-tts-24k_amd.inference")
    hp = hparams.tiny(n_spks=2)
    sd = synthetic.make_state_dict(hp, seed=7, duration_recipe=False)
    model = inference.MatchaTTSInfer(**hp.as_reference_kwargs())
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    model.decoder.solver = "midpoint"
    lengths = [12, 5, 9]
    x, x_len, spk = synthetic.make_inputs(hp, 3, 12, seed=77, lengths=lengths)
    noise = lambda n, t: synthetic.cpu_noise((n, hp.n_feats, t)).to(dev)

    def synth(xs, ls, ss, sync_max, z_fn):
        out = model.synthesise(xs, ls, 2, speaker=ss, sync_max=sync_max, z=z_fn)
        return out["mel"], out["mel_lengths"]

    mel_dp, lens_dp = dp.synthesise_dp(synth, x.to(dev), x_len.to(dev), spk.to(dev), noise_fn=noise)
    dp.FORCE_COLLECTIVES = False
    direct = model.synthesise(x.to(dev), x_len.to(dev), 2, speaker=spk.to(dev),
                              z=lambda t: noise(3, t))
    res["dp_equal"] = bool(torch.equal(mel_dp, direct["mel"])) and bool(torch.equal(lens_dp, direct["mel_lengths"]))
    res["finite"] = bool(torch.isfinite(mel_dp).all())
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    torch.save(res, os.path.join(out_dir, "rccl.pt"))


def test_rccl_world1_collectives_and_dp_flow(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / "rccl.pt")
    assert res["max"] == [7, 640, 3]
    assert res["gather_equal"] and res["ragged_equal"]
    assert res["dp_equal"] and res["finite"]
