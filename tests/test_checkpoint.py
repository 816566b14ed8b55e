"""Checkpoint converter: Lightning-style pickle -> flat safetensors + JSON and back."""
import json

import pytest
import torch

from conftest import sub


def _fake_lightning_ckpt(tmp_path, hp, sd):
    """What the reference's trainer writes, minus lightning: hyper_parameters in the constructor's nested format, state-dict
    keys with the torch.compile infix on the compiled sub-modules, optimizer / scheduler entries that must be dropped, and
    training-only tensors the path does not read."""
    kw = hp.as_reference_kwargs()
    kw["optimizer"] = {"lr": 1e-4}
    kw["scheduler"] = None
    noisy = {}
    for k, v in sd.items():
        if k.startswith("decoder.estimator."):
            k = k.replace("decoder.estimator.", "decoder.estimator._orig_mod.", 1)
        noisy[k] = v
    noisy["some.training_only.buffer"] = torch.zeros(3)
    path = tmp_path / "last.ckpt"
    torch.save({"hyper_parameters": kw, "state_dict": noisy, "epoch": 7}, path)
    return path


def test_convert_roundtrip(tmp_path):
    hparams, synthetic, ck = sub("hparams"), sub("synthetic"), sub("checkpoint")
    hp = hparams.tiny(n_spks=3)
    sd = synthetic.make_state_dict(hp, seed=7)
    src = _fake_lightning_ckpt(tmp_path, hp, sd)
    out = ck.convert_lightning_checkpoint(src, tmp_path / "conv")
    assert ck.is_converted(out) and not ck.is_converted(src)
    meta = json.loads((out / "hparams.json").read_text())
    assert meta["format_version"] == 1 and meta["path_hparams"]["decoder"]["channels"] == list(hp.decoder.channels)
    hp2, sd2 = ck.load_converted(out)
    assert hp2 == hp
    assert set(sd2) == {k for k, _, _ in synthetic.state_dict_spec(hp)}
    for k, v in sd2.items():
        assert v.dtype == torch.float32 and torch.equal(v, sd[k].float()), k


def test_convert_rejects_missing_and_misshaped(tmp_path):
    hparams, synthetic, ck = sub("hparams"), sub("synthetic"), sub("checkpoint")
    hp = hparams.tiny(n_spks=2)
    sd = synthetic.make_state_dict(hp, seed=7)
    key = next(k for k in sd if k.endswith("weight") and sd[k].dim() == 2)
    bad = dict(sd)
    del bad[key]
    with pytest.raises(KeyError):
        ck.select_path_tensors(hp, bad)
    bad = dict(sd)
    bad[key] = sd[key][:, :-1]
    with pytest.raises(ValueError):
        ck.select_path_tensors(hp, bad)


@pytest.mark.gpu
def test_load_matcha_from_converted_dir(tmp_path):
    """load_matcha on the converted directory gives the same mel as loading the state dict directly."""
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    hparams, synthetic, ck, inf = sub("hparams"), sub("synthetic"), sub("checkpoint"), sub("inference")
    hp = hparams.tiny(n_spks=2)
    sd = synthetic.make_state_dict(hp, seed=7)
    out = ck.convert_lightning_checkpoint(_fake_lightning_ckpt(tmp_path, hp, sd), tmp_path / "conv")
    a = inf.load_matcha("converted", str(out))
    b = inf.load_matcha("pickle", str(tmp_path / "last.ckpt"))
    x, x_len, spk = synthetic.make_inputs(hp, 2, 12, seed=3)
    dev = torch.device("cuda")
    ma = a.synthesise(x.to(dev), x_len.to(dev), 2, speaker=spk.to(dev))["mel"]
    mb = b.synthesise(x.to(dev), x_len.to(dev), 2, speaker=spk.to(dev))["mel"]
    assert torch.equal(ma, mb)
