"""GPU parity of the Vocos-24k head (mtts_vocos_decode through the Python mirror) against the CPU oracle
(oracle/vocos_oracle.py; "parity unpinned" vs the upstream vocos package, see its header)."""
import sys

import pytest
import torch

from conftest import ROOT, sub

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    import vocos_oracle
    syn = sub("synthetic")
    voc = sub("vocoder")
    sd = syn.make_vocos_state_dict(seed=11)
    wrapper = voc.load_model("cuda", state_dict=sd)
    return vocos_oracle, sd, wrapper


@pytest.mark.parametrize("B,T", [(1, 320), (3, 77), (2, 5)])
def test_decode_matches_oracle(env, B, T):
    V, sd, wrapper = env
    g = torch.Generator().manual_seed(B * 1000 + T)
    mel = torch.randn(B, 100, T, generator=g) * 2.0 - 4.0
    with torch.inference_mode():
        ref = V.decode(sd, mel.double() if False else mel)
    out = wrapper(mel.cuda())
    assert out.shape == ref.shape == (B, 256 * (T - 1))
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


def test_backbone_against_fp64(env):
    """fp64 oracle as the arbiter: the HIP result must be as close to it as the fp32 CPU oracle is (same order)."""
    V, sd, wrapper = env
    mel = torch.randn(2, 100, 64, generator=torch.Generator().manual_seed(1)) * 2.0 - 4.0
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.inference_mode():
        ref64 = V.decode(sd64, mel.double())
        ref32 = V.decode(sd, mel)
    out = wrapper(mel.cuda()).cpu().double()
    e_hip, e_cpu = (out - ref64).abs().max().item(), (ref32.double() - ref64).abs().max().item()
    assert e_hip < 10 * e_cpu + 1e-6, (e_hip, e_cpu)


def test_pipeline_tail_to_waveform(env):
    """mel -> to_waveform (reference inference.py:260-265): peak normalisation only when |a| > 1."""
    V, sd, wrapper = env
    inf = sub("inference")
    mel = torch.randn(1, 100, 40, generator=torch.Generator().manual_seed(2)) * 2.0 - 4.0
    wav = inf.to_waveform(mel.cuda(), wrapper)
    with torch.inference_mode():
        ref = V.to_waveform_scale(V.decode(sd, mel)).squeeze()
    assert wav.device.type == "cpu" and wav.shape == ref.shape
    assert (wav - ref).abs().max() < 2e-4
    with pytest.raises(RuntimeError):
        wrapper(mel)          # CPU mel: no fallback
