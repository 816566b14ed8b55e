"""The CPU oracle against the golden vectors recorded from the reference's own code
(tests/golden/make_golden.py; reference files cited there).  Bit-exactness was observed in the
build container; the tolerance here (1e-5) allows for a different host CPU / BLAS blocking."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

TOL = 1e-5


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def tiny(hparams, synthetic):
    hp = hparams.tiny(n_spks=2)
    return hp, synthetic.make_state_dict(hp, seed=7)


def test_randn_stream_matches_recorded(synthetic):
    g = np.load(GOLDEN / "randn42.npz")
    z = synthetic.cpu_noise((1, 100, 640)).flatten()
    assert np.array_equal(z[:16].numpy(), g["head"]) and np.array_equal(z[-16:].numpy(), g["tail"])


def test_tiny_encoder(tiny, oracle):
    hp, sd = tiny
    g = np.load(GOLDEN / "tiny_encoder.npz")
    spk = _t(g["speakers"])
    with torch.inference_mode():
        mu, logw, mask = oracle.text_encoder_forward(sd, hp, _t(g["x"]), _t(g["x_lengths"]),
                                                     sd["speaker_embeddings_enc.weight"][spk],
                                                     sd["speaker_embeddings_dur.weight"][spk])
    assert (mu - _t(g["mu_x"])).abs().max() < TOL
    assert (logw - _t(g["logw"])).abs().max() < TOL
    assert torch.equal(mask, _t(g["x_mask"]))


@pytest.mark.parametrize("sdpa", [True, False])
def test_tiny_decoder(tiny, oracle, synthetic, sdpa):
    hp, sd = tiny
    g = np.load(GOLDEN / "tiny_decoder.npz")
    T, nf = int(g["T"]), hp.n_feats
    x = _t(synthetic.portable_normal(11, 1, 2 * nf * T).reshape(2, nf, T))
    mu = _t(synthetic.portable_normal(11, 2, 2 * nf * T).reshape(2, nf, T))
    mask = oracle.sequence_mask(_t(g["lengths"]), T).unsqueeze(1).float()
    for tv in (0.0, 0.37):
        with torch.inference_mode():
            v = oracle.decoder_forward(sd, hp, x, mask, mu, torch.tensor(tv), use_torch_sdpa=sdpa)
        assert (v - _t(g[f"v_t{tv}"])).abs().max() < (TOL if sdpa else 1e-4)


@pytest.mark.parametrize("solver,steps", [("euler", 2), ("midpoint", 2), ("rk4", 1)])
def test_tiny_synthesise(tiny, oracle, synthetic, solver, steps):
    hp, sd = tiny
    g = np.load(GOLDEN / "tiny_synth.npz")
    x, x_len, spk = synthetic.make_inputs(hp, 2, 12, seed=1234, lengths=[12, 9])
    for b in range(2):
        with torch.inference_mode():
            out = oracle.synthesise(sd, hp, x[b:b + 1, : int(x_len[b])], x_len[b:b + 1], steps, speaker=int(spk[b]),
                                    scale_correction=1.03, length_scale=0.9, solver=solver)
        ref = _t(g[f"mel_{solver}{steps}_b{b}"])
        assert out["mel"].shape == ref.shape
        assert (out["mel"] - ref).abs().max() < TOL * 10
        assert torch.equal(out["durations"], _t(g[f"dur_b{b}"]))


def test_tiny_voice_mix(tiny, oracle, synthetic):
    hp, sd = tiny
    g = np.load(GOLDEN / "tiny_synth.npz")
    x, x_len, _ = synthetic.make_inputs(hp, 2, 12, seed=1234, lengths=[12, 9])
    with torch.inference_mode():
        out = oracle.synthesise(sd, hp, x[:1], x_len[:1], 2, voice_mix=[(0, 0.7), (1, 0.3)], solver="euler")
    assert (out["mel"] - _t(g["mel_mix"])).abs().max() < TOL * 10


def test_prod_single_utterance(hparams, synthetic, oracle):
    hp = hparams.prod_v20(n_spks=1)
    sd = synthetic.make_state_dict(hp, seed=7)
    g = np.load(GOLDEN / "prod_synth.npz")
    x, x_len, _ = synthetic.make_inputs(hp, 1, 128, seed=1234)
    assert np.array_equal(x.numpy(), g["x"])
    with torch.inference_mode():
        out = oracle.synthesise(sd, hp, x, x_len, 2, speaker=0, solver="euler")
    assert out["t_pad"] == 640 and out["mel"].shape == (1, 100, 320)
    assert (out["logw"] - _t(g["logw"])).abs().max() < TOL
    assert (out["mu_y"] - _t(g["mu_y"])).abs().max() < TOL
    assert (out["mel"] - _t(g["mel_euler2"])).abs().max() < 2e-4


def test_prod_ragged_batch(hparams, synthetic, oracle):
    hp = hparams.prod_v20(n_spks=3)
    sd = synthetic.make_state_dict(hp, seed=7)
    g = np.load(GOLDEN / "prod_batch.npz")
    x, x_len, spk = synthetic.make_inputs(hp, 3, 128, seed=1234, lengths=[128, 100, 77])
    with torch.inference_mode():
        out = oracle.synthesise(sd, hp, x, x_len, 2, speaker=spk, solver="euler")
    assert torch.equal(out["mel_lengths"], _t(g["y_lengths"]))
    assert (out["mel"] - _t(g["mel"])).abs().max() < 2e-4


@pytest.mark.parametrize("tag,which", [("dp_tiny", "tiny"), ("dp_prod", "prod")])
def test_duration_predictor_live(tag, which, hparams, synthetic, oracle):
    """duration_recipe=False: the DurationPredictor's conv/ReLU/LN/FiLM stack decides logw (reference text_encoder.py:64-112),
    hence the durations, T_pad and everything downstream (inference.py:127-146).  Fixtures recorded from the reference."""
    g = np.load(GOLDEN / f"{tag}.npz")
    hp = hparams.tiny(n_spks=2) if which == "tiny" else hparams.prod_v20(n_spks=3)
    sd = synthetic.make_state_dict(hp, seed=int(g["seed_w"]), duration_recipe=False)
    lengths = [int(v) for v in g["x_lengths"]]
    x, x_len, spk = synthetic.make_inputs(hp, len(lengths), max(lengths), seed=int(g["seed_x"]), lengths=lengths)
    assert np.array_equal(x.numpy(), g["x"]) and np.array_equal(spk.numpy(), g["speakers"])
    sc, ls, steps = float(g["sc"]), float(g["ls"]), int(g["steps"])
    logw_g = _t(g["logw"])
    valid = oracle.sequence_mask(x_len, max(lengths))
    assert float(logw_g[:, 0][valid].std()) > 0.3            # the fixture is NOT the constant-duration recipe
    with torch.inference_mode():
        out = oracle.synthesise(sd, hp, x, x_len, steps, speaker=spk, scale_correction=sc, length_scale=ls, solver="euler")
    assert (out["logw"] - logw_g).abs().max() < TOL
    assert (out["mu_x"] - _t(g["mu_x"])).abs().max() < TOL
    assert torch.equal(out["durations"], _t(g["durations"]))
    assert torch.equal(out["mel_lengths"], _t(g["y_lengths"]))
    assert (out["mu_y"] - _t(g["mu_y"])).abs().max() < TOL
    assert (out["mel"] - _t(g["mel"])).abs().max() < 2e-4
    if which == "tiny":     # the reference's own batch-1 synthesise (its phoneme_durations)
        n0 = lengths[0]
        with torch.inference_mode():
            solo = oracle.synthesise(sd, hp, x[:1, :n0], x_len[:1], steps, speaker=int(spk[0]), scale_correction=sc,
                                     length_scale=ls, solver="euler")
        assert torch.equal(solo["durations"], _t(g["solo_dur"]))
        assert (solo["mel"] - _t(g["solo_mel"])).abs().max() < 1e-4


def test_autocast_anchor_is_self_consistent():
    """prod_autocast.npz (the reference's synthesise under torch.autocast on the prod_synth inputs, noise pinned to the fp32
    draw): the recorded error figures are those of the recorded mels against the fp32 golden, 16-bit operands cost the
    reference itself 1e-2..3e-1 on |mel| ~ 45, and bf16 (8 significand bits) costs several times fp16 (11)."""
    g, a = np.load(GOLDEN / "prod_synth.npz"), np.load(GOLDEN / "prod_autocast.npz")
    gold = _t(g["mel_euler10"])
    for name in ("bf16", "fp16"):
        mel = _t(a[f"mel_{name}"])
        assert mel.shape == gold.shape
        d = (mel - gold).abs()
        assert abs(float(d.max()) - float(a[f"err_{name}"][0])) < 1e-6 and abs(float(d.mean()) - float(a[f"err_{name}"][1])) < 1e-6
    assert 1e-2 < a["err_fp16"][0] < 1e-1 < a["err_bf16"][0] < 1.0
    assert abs(float(gold.abs().max()) - float(a["mel_abs_max"])) < 1e-6
