"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mtts.h declares; the Python mirror exposes the reference's names; host logic that needs no GPU."""
import ctypes
import re

import pytest
import torch

from conftest import ROOT, sub


@pytest.fixture(scope="module")
def lib():
    hip = sub("_hip")
    hip.build()
    return hip.load()


def test_library_exports_every_declared_symbol(lib):
    header = (ROOT / "include" / "mtts.h").read_text()
    names = sorted(set(re.findall(r"\b(mtts_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mtts.h but not exported"
    assert lib.mtts_abi_version() == 2


def test_context_rejects_bad_configs_and_missing_tensors(lib, hparams):
    hip = sub("_hip")
    h = hip.HipModel(hparams.tiny())
    assert lib.mtts_weights_bytes(h.ctx) == -1                      # nothing registered yet
    assert b"missing tensor" in lib.mtts_last_error()
    bad = hparams.tiny()
    bad.decoder.channels = (48, 48)                                 # not a multiple of 32
    with pytest.raises(RuntimeError):
        hip.HipModel(bad)


def test_packing_sizes_without_gpu(lib, hparams, synthetic):
    """Weight packing is host-side: every tensor of the spec is consumed and the image size is reported."""
    hip = sub("_hip")
    hp = hparams.tiny(n_spks=2)
    h = hip.HipModel(hp)
    with pytest.raises(RuntimeError):
        h.load_state_dict(synthetic.make_state_dict(hp), "cpu")    # no CPU path
    for k, v in synthetic.make_state_dict(hp).items():
        if k in ("mel_mean", "mel_std"):
            continue
        h._set(k, v)
        if k.endswith("ff.net.0.alpha"):
            h._set(k + "_exp", torch.exp(v))
        elif k.endswith("ff.net.0.beta"):
            h._set(k[:-4] + "inv_beta", 1.0 / (torch.exp(v) + 1e-9))
    cos, sin = hip.rope_tables(12)
    h._set("aux.rope_cos", cos)
    h._set("aux.rope_sin", sin)
    h._set("aux.time_freqs", hip.time_freqs(2 * hp.n_feats))
    n = lib.mtts_weights_bytes(h.ctx)
    raw = sum(v.numel() for v in synthetic.make_state_dict(hp).values()) * 4
    assert n > raw                                                  # padded panels are larger than the raw tensors
    assert lib.mtts_decoder_workspace_bytes(h.ctx, 2, 24) > 0
    assert lib.mtts_decoder_workspace_bytes(h.ctx, 2, 25) == -1     # odd T cannot pass the U-Net (fix_len_compatibility)
    assert lib.mtts_encoder_workspace_bytes(h.ctx, 2, 12) > 0


def test_module_tree_matches_reference_state_dict_names(hparams, synthetic):
    inf = sub("inference")
    hp = hparams.prod_v20(n_spks=10)
    m = inf.MatchaTTSInfer(**hp.as_reference_kwargs())
    sd = synthetic.make_state_dict(hp)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert sum(p.numel() for p in m.parameters()) == 52_839_657     # total with 10 speakers
    # checkpoints trained under torch.compile carry _orig_mod infixes
    renamed = {k.replace("encoder.encoder.", "encoder.encoder._orig_mod.").replace(".ff.net", ".ff._orig_mod.net"): v
               for k, v in sd.items()}
    assert m.load_state_dict(renamed, strict=True).missing_keys == []
    with pytest.raises(RuntimeError):
        m.synthesise(torch.zeros(1, 4, dtype=torch.long), torch.tensor([4]), 2)   # model on CPU: fails loudly


def test_hparams_roundtrip_from_reference_kwargs(hparams):
    hp = hparams.prod_v20(n_spks=3)
    again = hparams.from_reference_kwargs(**hp.as_reference_kwargs())
    assert again.to_dict() == hp.to_dict()
    with pytest.raises(NotImplementedError):
        kw = hp.as_reference_kwargs()
        kw["decoder"]["down_block_type"] = "conformer"
        hparams.from_reference_kwargs(**kw)
