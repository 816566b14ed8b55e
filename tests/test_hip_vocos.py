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


def test_istft_stage_against_torch_istft():
    """Independent of the restated oracle: with `head.out.weight` = 0 the head's spectrum is its bias in every frame, whatever the
    backbone does, so the audio is torch.istft of a stationary spectrogram -- an analytic check of spec_polar (exp / clip / cos /
    sin), the inverse-DFT-times-window GEMM, the overlap-add with the squared-window envelope and the centre trim against
    PyTorch's own istft (n_fft 1024, hop 256, periodic hann, center=True: reference vocos24k/config.yaml:18-24)."""
    if not torch.cuda.is_available():
        pytest.fail("a HIP device is required for -m gpu tests (no CPU fallback exists)")
    syn, voc = sub("synthetic"), sub("vocoder")
    sd = syn.make_vocos_state_dict(seed=11)
    n_fft, hop, nb, T = 1024, 256, 513, 40
    g = torch.Generator().manual_seed(7)
    logmag = torch.rand(nb, generator=g) * 4.0 - 3.0
    logmag[5] = 6.0                                            # exp(6) = 403 > the head's clip at 1e2
    phase = (torch.rand(nb, generator=g) * 2 - 1) * 3.0
    sd["head.out.weight"] = torch.zeros_like(sd["head.out.weight"])
    sd["head.out.bias"] = torch.cat([logmag, phase])
    wrapper = voc.load_model("cuda", state_dict=sd)
    mel = torch.randn(2, 100, T, generator=g)
    out = wrapper(mel.cuda()).cpu()
    mag = torch.clip(torch.exp(logmag), max=1e2)
    spec = (mag * torch.cos(phase) + 1j * mag * torch.sin(phase)).to(torch.complex64)
    ref = torch.istft(spec[None, :, None].expand(1, nb, T).contiguous(), n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.hann_window(n_fft), center=True)
    assert out.shape == (2, hop * (T - 1)) and ref.shape == (1, hop * (T - 1))
    scale = float(ref.abs().max())
    assert (out[0] - ref[0]).abs().max() < 2e-5 * scale and torch.equal(out[0], out[1])
