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
    # the first load wrote the packed weight image beside the converted tensors; a second load adopts it instead of packing
    assert not a.hip.cache_hit and len(list(out.glob("packed-*.bin"))) == 1
    c = inf.load_matcha("converted again", str(out))
    mc = c.synthesise(x.to(dev), x_len.to(dev), 2, speaker=spk.to(dev))["mel"]
    assert c.hip.cache_hit and torch.equal(mc, ma)
    assert len(list(out.glob("packed-*.bin"))) == 1


def _register(hip, h, hp, sd):
    """What HipModel.load_state_dict registers before packing (no device needed)."""
    for k, v in sd.items():
        if k in ("mel_mean", "mel_std"):
            continue
        h._set(k, v)
        if k.endswith("ff.net.0.alpha"):
            h._set(k + "_exp", torch.exp(v))
        elif k.endswith("ff.net.0.beta"):
            h._set(k[:-4] + "inv_beta", 1.0 / (torch.exp(v) + 1e-9))
    e = hp.encoder
    cos, sin = hip.rope_tables(int(((e.n_channels + hp.spk_emb_dim) // e.n_heads) * 0.5))
    h._set("aux.rope_cos", cos)
    h._set("aux.rope_sin", sin)
    h._set("aux.time_freqs", hip.time_freqs(2 * hp.n_feats))


def test_packed_image_export_import_roundtrip_without_gpu(tmp_path):
    """The packed-image cache (include/mtts.h mtts_export_weights / mtts_import_weights, checkpoint.packed_cache): an exported
    image adopted by a second context of the same signature reproduces itself byte for byte through the layout-only pass;
    another architecture's image is refused; the cache key follows the signature and the tensors."""
    import ctypes as C
    import dataclasses
    import numpy as np
    hparams, synthetic, ck, hip = sub("hparams"), sub("synthetic"), sub("checkpoint"), sub("_hip")
    hip.build()
    hp = dataclasses.replace(hparams.tiny(n_spks=2), decoder=dataclasses.replace(hparams.tiny().decoder, channels=(128, 128), attention_head_dim=64,
                                                                                 n_blocks=2, num_heads=2))       # chain streams included
    sd = synthetic.make_state_dict(hp, seed=7)
    a, b = hip.HipModel(hp), hip.HipModel(hp)
    _register(hip, a, hp, sd)
    _register(hip, b, hp, sd)
    assert a.weights_signature() == b.weights_signature() and a.weights_signature().startswith("mtts-2-")
    n = a.lib.mtts_weights_bytes(a.ctx)
    img, sat = np.empty(n, dtype=np.uint8), C.c_int(-1)
    assert a.lib.mtts_export_weights(a.ctx, img.ctypes.data, n, C.byref(sat)) == 0 and sat.value == 0
    assert b.lib.mtts_import_weights(b.ctx, img.ctypes.data, n, 0) == 0
    assert b.lib.mtts_weights_bytes(b.ctx) == n
    back = np.empty(n, dtype=np.uint8)
    assert b.lib.mtts_export_weights(b.ctx, back.ctypes.data, n, None) == 0
    assert np.array_equal(back, img)
    other = hip.HipModel(hparams.tiny(n_spks=2))
    _register(hip, other, hparams.tiny(n_spks=2), synthetic.make_state_dict(hparams.tiny(n_spks=2), seed=7))
    assert other.weights_signature() != a.weights_signature()
    assert other.lib.mtts_import_weights(other.ctx, img.ctypes.data, n, 0) == -1
    assert b"size" in other.lib.mtts_last_error()
    # the cache file: keyed by signature and tensors, damaged files read as absent
    c1 = ck.packed_cache(tmp_path, a.weights_signature(), sd)
    assert c1.read() is None
    c1.write(img, False)
    got = c1.read()
    assert got is not None and np.array_equal(got["data"], img) and got["saturates"] is False
    sd2 = dict(sd)
    k0 = next(iter(sd2))
    sd2[k0] = sd2[k0] + 1.0
    assert ck.packed_cache(tmp_path, a.weights_signature(), sd2).path != c1.path
    assert ck.packed_cache(tmp_path, other.weights_signature(), sd).path != c1.path
    c1.path.write_bytes(c1.path.read_bytes()[:100])
    assert c1.read() is None
