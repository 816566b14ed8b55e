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


@pytest.mark.parametrize("C,inner,ch,n_qkv", [(384, 384, 128, 1152), (384, 384, 256, 1152), (256, 192, 128, 576), (128, 128, 128, 0)])
def test_chain_fragment_stream_layout(lib, C, inner, ch, n_qkv):
    """Host packing of the transformer-block chain's weight stream (csrc/tblock_chain.hip), checked WITHOUT a GPU by walking
    the stream exactly as the kernel does -- per wave, per phase, per k-step, per tile: head fragment then residual fragment,
    lane (r, q) = panel row n0 + r, columns k0 + 8 q .. + 7 -- and rebuilding every panel from it (h + l / 2^11 == w to 22 bits)."""
    import numpy as np
    rng = np.random.default_rng(5)
    w_out = rng.standard_normal((C, inner)).astype(np.float32)
    w1 = rng.standard_normal((4 * C, C)).astype(np.float32)
    w2 = rng.standard_normal((C, 4 * C)).astype(np.float32)
    w_qkv = rng.standard_normal((n_qkv, C)).astype(np.float32) if n_qkv else None
    frags = lib.mtts_chain_stream_frags(C, inner, ch, n_qkv)
    assert frags > 0
    dst = np.full(frags * 8 * 512 + 64, 0x7E00, dtype=np.uint16)            # NaN canary behind the buffer
    assert lib.mtts_chain_stream_pack(C, inner, ch, n_qkv, w_out.ctypes.data, w1.ctypes.data, w2.ctypes.data,
                                      w_qkv.ctypes.data if n_qkv else None, dst.ctypes.data) == 0
    assert (dst[-64:] == 0x7E00).all()                                       # nothing written past the end
    stream = dst[:-64].view(np.float16).reshape(8, frags, 64, 8).astype(np.float64)
    NT, NT1, KG, KG2, R = C // 128, ch // 128, C // 32, ch // 32, (24 if C == 384 else 8)
    lane = np.arange(64)
    r, q = lane & 15, lane >> 4
    got = {"out": np.zeros_like(w_out, dtype=np.float64), "w1": np.zeros_like(w1, dtype=np.float64),
           "w2": np.zeros_like(w2, dtype=np.float64), "qkv": np.zeros((n_qkv, C))}

    def take(panel, wave_frags, pos, n0, k0, n_valid):
        h, l = wave_frags[pos], wave_frags[pos + 1]
        val = h + l / 2048.0
        for j in range(8):
            rows = n0 + r
            ok = rows < n_valid
            panel[rows[ok], (k0 + 8 * q + j)[ok]] = val[ok, j]
        if (n0 + 15) >= n_valid:                                            # padding rows of a partly empty tile are zero
            assert (val[(n0 + r) >= n_valid] == 0).all()
        return pos + 2

    passes = -(-(n_qkv // 16) // (8 * NT)) if n_qkv else 0
    for w in range(8):
        f, pos = stream[w], 0
        for s in range(inner // 32):
            for t in range(NT):
                pos = take(got["out"], f, pos, 16 * (w * NT + t), 32 * s, C)
        assert pos % R == 0                                                   # the FeedForward starts on ring slot 0
        for j in range(4 * C // ch):
            for s in range(KG):
                for t in range(NT1):
                    pos = take(got["w1"], f, pos, j * ch + 16 * (w * NT1 + t), 32 * s, 4 * C)
            for s in range(KG2):
                for t in range(NT):
                    pos = take(got["w2"], f, pos, 16 * (w * NT + t), j * ch + 32 * s, C)
            assert pos % R == 0
        for ps in range(passes):
            for s in range(KG):
                for t in range(NT):
                    pos = take(got["qkv"], f, pos, 16 * (ps * 8 * NT + w * NT + t), 32 * s, n_qkv)
        assert pos + R == frags and (f[pos:] == 0).all()                      # the ring's run-out reads zeros
    for name, ref in (("out", w_out), ("w1", w1), ("w2", w2), ("qkv", w_qkv)):
        if ref is not None and ref.size:
            assert np.abs(got[name] - ref).max() <= 2.0 ** -21 * np.abs(ref).max(), name


def test_context_refuses_concurrent_use(lib, hparams):
    """A context is single-threaded (per-call state: range-flag pointer, frame limits, profiler records).  While one thread is
    inside an entry point -- stood in for by the mtts_debug_hold test hook, no GPU needed -- another thread's call returns an
    error naming the cause instead of interleaving its launches; afterwards the context works again."""
    import threading
    import time
    hip = sub("_hip")
    h = hip.HipModel(hparams.tiny())
    t = threading.Thread(target=lambda: lib.mtts_debug_hold(h.ctx, 400))
    t.start()
    time.sleep(0.1)
    rc = lib.mtts_text_encoder_forward(h.ctx, None, None, None, None, 1, 4, None, None, None, None, 0, None)
    msg = lib.mtts_last_error()
    t.join()
    assert rc == -1 and b"in use by another thread" in msg
    assert lib.mtts_debug_hold(h.ctx, 1) == 0                       # released
    rc = lib.mtts_text_encoder_forward(h.ctx, None, None, None, None, 1, 4, None, None, None, None, 0, None)
    assert rc == -1 and b"weights not uploaded" in lib.mtts_last_error()      # the ordinary check, not the guard


def test_chain_launch_plan_keeps_grids_to_one_round(lib):
    """mtts_chain_plan (host arithmetic of the model's chain launches, default MTTS_CHAIN_PF = 16): 32-row workgroups while they and
    the prefetch workgroups are one round of the 256 CUs, 48-row ones beyond (B = 25: 8050 rows -> 168 + 8, not 252 + 8), and no
    prefetchers when they alone would push a one-round grid into a second round (B = 37: 249 workgroups)."""
    import ctypes, os
    if "MTTS_CHAIN_PF" in os.environ:
        pytest.skip("MTTS_CHAIN_PF overrides the default this test states")
    def plan(M, ch=256):
        qb, pf = ctypes.c_int(0), ctypes.c_int(0)
        assert lib.mtts_chain_plan(M, ch, ctypes.byref(qb), ctypes.byref(pf)) == 0
        return qb.value, pf.value
    assert plan(32 * 161) == (32, 16)             # the half-length level of B = 32 (when the thresholds let it through)
    assert plan(7680) == (32, 16)                 # 240 + 16 = 256
    assert plan(7681) == (48, 16)
    assert plan(25 * 322) == (48, 16)
    assert plan(32 * 322) == (48, 16)             # 215 + 16
    assert plan(36 * 322) == (48, 8)              # 242 workgroups: only eight prefetchers fit the round
    assert plan(37 * 322) == (48, 0)              # 249 workgroups: none do
    assert plan(64 * 322) == (48, 16)             # several rounds anyway
    assert plan(5152, ch=128) == (32, 16) and plan(10304, ch=128) == (64, 16)
    assert lib.mtts_chain_plan(0, 256, None, None) != 0


@pytest.mark.parametrize("C,inner,ch,n_qkv", [(384, 384, 256, 1152), (384, 384, 256, 0), (256, 192, 128, 576), (128, 128, 128, 384)])
def test_chain_pair_stream_layout(lib, C, inner, ch, n_qkv):
    """The pair form's streams (csrc/tblock_chain.hip chain_stream_pack_pair), walked as the kernel walks them: both halves carry the
    whole out-projection, half h the hidden chunks [h NCH/2, (h+1) NCH/2) and the q|k|v passes [0, ceil(P/2)) / [ceil(P/2), P); every
    (half, wave) stream has the same length, ends in zeros, and together the two halves rebuild every panel."""
    import numpy as np
    rng = np.random.default_rng(9)
    w_out = rng.standard_normal((C, inner)).astype(np.float32)
    w1 = rng.standard_normal((4 * C, C)).astype(np.float32)
    w2 = rng.standard_normal((C, 4 * C)).astype(np.float32)
    w_qkv = rng.standard_normal((n_qkv, C)).astype(np.float32) if n_qkv else None
    frags = lib.mtts_chain_stream_frags_pair(C, inner, ch, n_qkv)
    assert frags > 0
    dst = np.full(2 * frags * 8 * 512 + 64, 0x7E00, dtype=np.uint16)
    assert lib.mtts_chain_stream_pack_pair(C, inner, ch, n_qkv, w_out.ctypes.data, w1.ctypes.data, w2.ctypes.data,
                                           w_qkv.ctypes.data if n_qkv else None, dst.ctypes.data) == 0
    assert (dst[-64:] == 0x7E00).all()
    stream = dst[:-64].view(np.float16).reshape(2, 8, frags, 64, 8).astype(np.float64)
    NT, NT1, KG, KG2, R, NCH = C // 128, ch // 128, C // 32, ch // 32, (24 if C == 384 else 8), 4 * C // ch
    lane = np.arange(64)
    r, q = lane & 15, lane >> 4
    got = {"w1": np.zeros_like(w1, dtype=np.float64), "w2": np.zeros_like(w2, dtype=np.float64), "qkv": np.zeros((n_qkv, C))}

    def take(panel, wave_frags, pos, n0, k0, n_valid):
        val = wave_frags[pos] + wave_frags[pos + 1] / 2048.0
        for j in range(8):
            rows = n0 + r
            ok = rows < n_valid
            panel[rows[ok], (k0 + 8 * q + j)[ok]] = val[ok, j]
        return pos + 2

    passes = -(-(n_qkv // 16) // (8 * NT)) if n_qkv else 0
    p_half = (passes + 1) // 2
    for h in range(2):
        out = np.zeros_like(w_out, dtype=np.float64)
        for w in range(8):
            f, pos = stream[h, w], 0
            for s in range(inner // 32):
                for t in range(NT):
                    pos = take(out, f, pos, 16 * (w * NT + t), 32 * s, C)
            assert pos % (8 if C != 384 else 12) == 0
            for j in range(h * NCH // 2, (h + 1) * NCH // 2):
                for s in range(KG):
                    for t in range(NT1):
                        pos = take(got["w1"], f, pos, j * ch + 16 * (w * NT1 + t), 32 * s, 4 * C)
                for s in range(KG2):
                    for t in range(NT):
                        pos = take(got["w2"], f, pos, 16 * (w * NT + t), j * ch + 32 * s, C)
            for ps in range(p_half if h else 0, passes if h else p_half):
                for s in range(KG):
                    for t in range(NT):
                        pos = take(got["qkv"], f, pos, 16 * (ps * 8 * NT + w * NT + t), 32 * s, n_qkv)
            assert pos + R <= frags and (f[pos:] == 0).all()
        assert np.abs(out - w_out).max() <= 2.0 ** -21 * np.abs(w_out).max()
    for name, ref in (("w1", w1), ("w2", w2), ("qkv", w_qkv)):
        if ref is not None and ref.size:
            assert np.abs(got[name] - ref).max() <= 2.0 ** -21 * np.abs(ref).max(), name
