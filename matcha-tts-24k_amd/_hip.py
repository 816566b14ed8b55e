"""ctypes binding of libmtts_hip.so (C ABI in include/mtts.h) + the in-tree hipcc build.

There is no CPU fallback: every entry point raises if the library is missing or a tensor is not
on a HIP device.  PyTorch is used only for device memory (workspaces, outputs) and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

from .hparams import PathHParams

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = Path(os.environ["MTTS_HIP_LIB"]) if os.environ.get("MTTS_HIP_LIB") else HERE / "libmtts_hip.so"   # override: A/B of two builds
SOURCES = ["gemm_f32.hip", "attention_f32.hip", "gemm_p16.hip", "tblock_chain.hip", "norm_glue.hip", "vocos.hip", "model.hip"]
HEADERS = [CSRC / "kernels.h", CSRC / "device_utils.h", CSRC / "model.h", HERE.parent / "include" / "mtts.h"]
SOLVERS = {"euler": 0, "midpoint": 1, "rk4": 2}


class MttsConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_feats", "n_spks", "spk_emb_dim", "n_vocab", "enc_channels", "enc_filter", "enc_heads", "enc_layers",
        "enc_kernel", "prenet_layers", "prenet_kernel", "dp_filter", "dp_kernel", "dp_layers", "dec_levels")] + [
        ("dec_channels", C.c_int32 * 4)] + [(n, C.c_int32) for n in (
            "dec_head_dim", "dec_heads", "dec_n_blocks", "dec_mid_blocks")]


def _deps(src: Path, seen=None):
    """The source and the local headers it includes, transitively (quoted #include lines)."""
    import re
    seen = set() if seen is None else seen
    if src in seen or not src.exists():
        return seen
    seen.add(src)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', src.read_text(), flags=re.M):
        _deps((src.parent / inc).resolve(), seen)
    return seen


BUILD_MODE = "not built in this process"     # what the last build() call did (printed by __graft_entry__.build)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP sources for gfx950 into libmtts_hip.so next to this file (cross-compiles without a GPU).
    Incremental: a translation unit is recompiled when it or a header it includes is newer than its object; the link runs
    when any object is newer than the library.  ``BUILD_MODE`` records what happened (compiled / reused)."""
    global BUILD_MODE
    srcs = [CSRC / s for s in SOURCES]
    if LIB.exists() and not force:
        deps = set(HEADERS)
        for src in srcs:
            deps |= _deps(src.resolve())
        newest = max(p.stat().st_mtime for p in deps)
        if LIB.stat().st_mtime >= newest:
            BUILD_MODE = f"reused {LIB.name} (newer than every source and header)"
            return LIB
    # -ffp-contract=off: keep the reference's rounding points (no silent FMA fusion in the element-wise math).
    # -fno-slp-vectorize: hipcc (ROCm 7.2) otherwise packs adjacent fp32 ops into v_pk_mul_f32 / v_pk_add_f32.  With SLP on,
    #   gemm_f32_kernel<*, *, NORM=true, TERMS=6|2> returns grossly wrong rows (error 0.6 at magnitude 7) when its LayerNorm
    #   statistics come from partial moments (a_part): DETERMINISTIC repro = build with MTTS_SLP=1 and run
    #   tests/test_hip_kernels.py::test_layernorm_stats_travel_through_epilogue[bf16x6|f16x3s] (2 of 75 kernel tests fail, every
    #   run; all other translation units pass with SLP on).  The packed ops sit in the normalise -> split -> LDS-store chain
    #   ((x - mean) * rstd as v_pk_mul_f32 op_sel:[0,1], the split's subtractions as v_pk_add_f32 neg_lo/neg_hi); a scan for
    #   VALU -> DPP wait-state violations found none, the root cause inside the compiler's output is not isolated.  Packed fp32
    #   VALU is an anti-lever beside MFMAs anyway (MI355X_MICROARCH.md cycle table; 29.98 vs 29.8 ms/step), so the flag stays
    #   on for every translation unit.
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    slp = [] if os.environ.get("MTTS_SLP") == "1" else ["-fno-slp-vectorize"]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *slp, "-fPIC",
             *os.environ.get("MTTS_HIPCC_EXTRA", "").split()]
    # one translation unit per process (the GEMM files instantiate dozens of kernels each), objects in build/, then one link
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)

    compiled = []

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if obj.exists() and not force and obj.stat().st_mtime >= max(p.stat().st_mtime for p in _deps(src.resolve())):
            return obj
        compiled.append(src.name)
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}")
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    BUILD_MODE = f"compiled {', '.join(sorted(compiled)) or 'nothing'} with hipcc --offload-arch=gfx950 and linked {LIB.name}"
    return LIB


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the library and declare every signature of include/mtts.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists():
        raise RuntimeError(f"{LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP path has no CPU fallback)")
    lib = C.CDLL(str(LIB))
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    sig = {
        "mtts_abi_version": (i32, []),
        "mtts_last_error": (C.c_char_p, []),
        "mtts_create": (vp, [C.POINTER(MttsConfig)]),
        "mtts_destroy": (None, [vp]),
        "mtts_set_tensor": (i32, [vp, C.c_char_p, vp, i64]),
        "mtts_weights_bytes": (i64, [vp]),
        "mtts_upload_weights": (i32, [vp, vp, i64]),
        "mtts_debug_hold": (i32, [vp, i32]),
        "mtts_weights_signature": (i32, [vp, C.c_char_p, i64]),
        "mtts_export_weights": (i32, [vp, vp, i64, C.POINTER(i32)]),
        "mtts_import_weights": (i32, [vp, vp, i64, i32]),
        "mtts_encoder_workspace_bytes": (i64, [vp, i32, i32]),
        "mtts_text_encoder_forward": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp]),
        "mtts_speaker_embedding": (i32, [vp, i32, vp, i32, vp, vp]),
        "mtts_durations": (i32, [vp, vp, f32, f32, i32, i32, vp, vp, vp, vp]),
        "mtts_durations_per_utterance": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
        "mtts_align_pool": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
        "mtts_set_frame_limits": (i32, [vp, vp]),
        "mtts_decoder_workspace_bytes": (i64, [vp, i32, i32]),
        "mtts_decoder_forward": (i32, [vp, vp, vp, vp, f32, i32, i32, vp, vp, i64, vp]),
        "mtts_cfm_solve": (i32, [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp, i32, f32, f32, vp, i64, vp]),
        "mtts_fold_rows": (i32, [vp, i32, i32]),
        "mtts_cfm_solve_folded": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, i32, f32, f32, vp, i64, vp]),
        "mtts_gemm_packed_bytes": (i64, [i32, i32, i32]),
        "mtts_attention_p16": (i32, [vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp]),
        "mtts_gemm_p16_scratch_bytes": (i64, [i32, i32, i32, i32, i32]),
        "mtts_gemm_p16": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp,
                                i32, vp, f32, vp, i32, vp, f32, vp, i32, vp, vp]),
        "mtts_gemm_f32": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp,
                                i32, vp, f32, vp, i32, vp, i32, vp]),
        "mtts_attention_f32": (i32, [vp, vp, i32, i32, i32, i32, f32, i32, vp, vp]),
        "mtts_chain_stream_frags": (i64, [i32, i32, i32, i32]),
        "mtts_chain_plan": (i32, [i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mtts_chain_stream_frags_pair": (i64, [i32, i32, i32, i32]),
        "mtts_chain_stream_pack_pair": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp]),
        "mtts_chain_stream_pack": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp]),
        "mtts_tblock_chain_scratch_bytes": (i64, [i32, i32, i32, i32, i32]),
        "mtts_tblock_chain": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]),
        "mtts_tblock_chain_timed": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp,
                                          i32, C.POINTER(C.c_float)]),
        "mtts_tblock_chain_pair_timed": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp,
                                          i32, C.POINTER(C.c_float)]),
        "mtts_row_stats": (i32, [vp, i32, i32, i32, f32, vp, vp, vp]),
        "mtts_channel_layernorm": (i32, [vp, i32, i32, i32, vp, vp, f32, i32, vp, vp, vp, vp]),
        "mtts_groupnorm_scratch_bytes": (i64, [i32, i32, i32]),
        "mtts_groupnorm_mish": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
        "mtts_vocos_create": (vp, [i32, i32, i32, i32, i32, i32]),
        "mtts_vocos_destroy": (None, [vp]),
        "mtts_vocos_set_tensor": (i32, [vp, C.c_char_p, vp, i64]),
        "mtts_vocos_weights_bytes": (i64, [vp]),
        "mtts_vocos_upload_weights": (i32, [vp, vp, i64]),
        "mtts_vocos_workspace_bytes": (i64, [vp, i32, i32]),
        "mtts_vocos_decode": (i32, [vp, vp, i32, i32, vp, vp, i64, vp]),
        "mtts_gemm_terms": (i32, [vp]),
        "mtts_set_arithmetic": (i32, [vp, i32]),
        "mtts_weights_saturate": (i32, [vp]),
        "mtts_prof_enable": (i32, [vp, i32]),
        "mtts_prof_reset": (i32, [vp]),
        "mtts_prof_records": (i64, [vp, C.POINTER(C.c_double), i64]),
        "mtts_prof_tags": (i64, [vp, C.c_char_p, i64]),
        "mtts_prof_read": (i32, [vp, i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mtts_abi_version() != 2:
        raise RuntimeError("libmtts_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("mtts: " + load().mtts_last_error().decode("utf-8", "replace"))


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous HIP tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mtts: tensor is not on a HIP device; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("mtts: tensor must be contiguous")
    return t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def rope_tables(d: int, n: int = 4000):
    """cos/sin caches exactly as the reference builds them (text_encoder.py:138-146), [n, d] fp32 on the CPU."""
    theta = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    idx = torch.einsum("n,d->nd", torch.arange(n).float(), theta)
    idx2 = torch.cat([idx, idx], dim=1)
    return idx2.cos().contiguous(), idx2.sin().contiguous()


def time_freqs(dim: int) -> torch.Tensor:
    """Frequency table of SinusoidalPosEmb in the reference's fp32 arithmetic (decoder.py:24-26)."""
    import math
    half = dim // 2
    c = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half).float() * -c).contiguous()


class HipModel:
    """One mtts_ctx: packed weights on a device + cached workspaces + the path's entry points on torch tensors."""
    _uids = 0

    def __init__(self, hp: PathHParams, terms: Optional[int] = None):
        """``terms``: GEMM arithmetic (mtts_set_arithmetic); None = the library default (fp16 two-term split, MTTS_GEMM_TERMS)."""
        self.lib = load()
        self.hp = hp
        cfg = MttsConfig()
        e, d = hp.encoder, hp.decoder
        cfg.n_feats, cfg.n_spks, cfg.spk_emb_dim, cfg.n_vocab = hp.n_feats, hp.n_spks, hp.spk_emb_dim, hp.n_vocab
        cfg.enc_channels, cfg.enc_filter, cfg.enc_heads, cfg.enc_layers = e.n_channels, e.filter_channels, e.n_heads, e.n_layers
        cfg.enc_kernel, cfg.prenet_layers, cfg.prenet_kernel = e.kernel_size, e.prenet_layers, e.prenet_kernel_size
        cfg.dp_filter, cfg.dp_kernel, cfg.dp_layers = e.dp_filter_channels, e.dp_kernel_size, e.dp_n_layers
        if len(d.channels) > 4:
            raise ValueError("at most 4 decoder levels")
        cfg.dec_levels = len(d.channels)
        for i, ch in enumerate(d.channels):
            cfg.dec_channels[i] = ch
        cfg.dec_head_dim, cfg.dec_heads, cfg.dec_n_blocks, cfg.dec_mid_blocks = (d.attention_head_dim, d.num_heads, d.n_blocks,
                                                                                d.num_mid_blocks)
        self.ctx = self.lib.mtts_create(C.byref(cfg))
        if not self.ctx:
            raise RuntimeError("mtts_create: " + self.lib.mtts_last_error().decode())
        if terms is not None:
            check(self.lib.mtts_set_arithmetic(self.ctx, int(terms)))
        self.weights: Optional[torch.Tensor] = None
        HipModel._uids += 1
        self.uid = HipModel._uids    # unique per process (id() is re-used after garbage collection)
        self.generation = 0          # bumped by every load_state_dict: whatever captured the weights' address is stale afterwards
        self.device: Optional[torch.device] = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._last_ws: Dict[str, torch.Tensor] = {}     # workspace of the latest call per kind: its first word = range flag

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.mtts_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _set(self, key: str, t: torch.Tensor) -> None:
        a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
        check(self.lib.mtts_set_tensor(self.ctx, key.encode(), a.ctypes.data, a.size))

    def weights_signature(self) -> str:
        """Everything the packed image's layout depends on (mtts_weights_signature): the key of a packed-image cache."""
        buf = C.create_string_buffer(1024)
        if self.lib.mtts_weights_signature(self.ctx, buf, 1024) < 0:
            check(-1)
        return buf.value.decode()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device, cache_dir=None) -> None:
        """Register every tensor of a reference-format state dict (keys of SURVEY appendix A; torch.compile's
        ``_orig_mod.`` infix is ignored), add the host-precomputed tables, pack and upload.
        ``cache_dir`` (a converted checkpoint's directory, checkpoint.py): the packed image is read from / written to a cache
        file there, keyed by the library's layout signature and a digest of the tensors (``checkpoint.packed_cache``)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("mtts: weights must live on a HIP device; the HIP path has no CPU fallback")
        for k, v in sd.items():
            k = k.replace("_orig_mod.", "")
            if k in ("mel_mean", "mel_std") or "rope." in k:
                continue
            self._set(k, v)
            if k.endswith("ff.net.0.alpha"):     # SnakeBeta parameters in the reference's own fp32 ops (transformer.py:68-75)
                self._set(k + "_exp", torch.exp(v.detach().float().cpu()))
            elif k.endswith("ff.net.0.beta"):
                self._set(k[:-4] + "inv_beta", 1.0 / (torch.exp(v.detach().float().cpu()) + 0.000000001))
        e = self.hp.encoder
        dh = (e.n_channels + self.hp.spk_emb_dim) // e.n_heads
        cos, sin = rope_tables(int(dh * 0.5))
        self._set("aux.rope_cos", cos)
        self._set("aux.rope_sin", sin)
        self._set("aux.time_freqs", time_freqs(2 * self.hp.n_feats))
        self.cache_hit = False
        cache = None
        if cache_dir is not None:
            from . import checkpoint as ck
            cache = ck.packed_cache(cache_dir, self.weights_signature(), sd)
            image = cache.read()
            if image is not None:
                try:
                    check(self.lib.mtts_import_weights(self.ctx, image["data"].ctypes.data, image["data"].nbytes, int(image["saturates"])))
                    self.cache_hit = True
                except RuntimeError:
                    self.cache_hit = False       # (a stale or foreign file: pack from the tensors below and rewrite it)
        nbytes = self.lib.mtts_weights_bytes(self.ctx)
        if nbytes < 0:
            check(-1)
        if cache is not None and not self.cache_hit:
            host = np.empty(nbytes, dtype=np.uint8)
            sat = C.c_int(0)
            check(self.lib.mtts_export_weights(self.ctx, host.ctypes.data, nbytes, C.byref(sat)))
            cache.write(host, bool(sat.value))
        self.weights = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(self.lib.mtts_upload_weights(self.ctx, self.weights.data_ptr(), nbytes))
        self.device = device
        self.generation += 1
        self._ws.clear()
        self._last_ws.clear()

    def _workspace(self, kind: str, a: int, b: int) -> torch.Tensor:
        """One GROW-ONLY scratch buffer per (kind, stream): a serving process sees a new (B, T_pad) with almost every request,
        so buffers keyed by shape would pin HBM without bound.  The C side takes (pointer, byte count) and bump-allocates what
        the call needs from the front.  Concurrent calls on different streams must not share scratch, hence the stream key;
        a buffer being replaced stays alive until the work queued on its stream has drained (torch's caching allocator frees
        a block for re-use in stream order)."""
        fn = self.lib.mtts_decoder_workspace_bytes if kind == "dec" else self.lib.mtts_encoder_workspace_bytes
        n = fn(self.ctx, a, b)
        if n < 0:
            check(-1)
        key = (kind, stream_ptr())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < n:
            ws = None
            self._ws.pop(key, None)
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        self._last_ws[key] = ws
        return ws

    def range_flags(self) -> torch.Tensor:
        """Sticky range flags (include/mtts.h "range guard") of this stream's latest encoder and estimator calls as a device
        int32 tensor [2]; non-zero = an operand left the fp16 range and saturated.  Reading it (``.any().item()``) synchronises."""
        z = torch.zeros(1, dtype=torch.int32, device=self.device)
        sp = stream_ptr()
        parts = [self._last_ws[(k, sp)][:4].view(torch.int32) if (k, sp) in self._last_ws else z for k in ("enc", "dec")]
        return torch.cat(parts)

    def pair_timeouts(self) -> torch.Tensor:
        """Second word of this stream's latest estimator workspace header: non-zero = a workgroup of a pair-form chain launch waited in
        vain for its partner (csrc/tblock_chain.hip) -- the call's results are void.  A device int32 tensor [1]."""
        sp = stream_ptr()
        if ("dec", sp) not in self._last_ws:
            return torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._last_ws[("dec", sp)][4:8].view(torch.int32)

    def weights_saturate(self) -> bool:
        r = self.lib.mtts_weights_saturate(self.ctx)
        if r < 0:
            check(-1)
        return bool(r)

    def decoder_workspace_bytes(self, B: int, T: int) -> int:
        n = self.lib.mtts_decoder_workspace_bytes(self.ctx, int(B), int(T))
        if n < 0:
            check(-1)
        return n

    def note_workspace(self, kind: str, ws: torch.Tensor) -> None:
        """Make ``ws`` the buffer ``range_flags`` reads for this stream (a replayed HIP graph ran on it)."""
        self._last_ws[(kind, stream_ptr())] = ws

    def workspace_bytes_held(self) -> int:
        return sum(int(w.numel()) for w in self._ws.values())

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda:
            raise RuntimeError("mtts: input tensor is not on a HIP device; the HIP path has no CPU fallback")
        return t.detach().to(torch.float32).contiguous()

    # ------------------------------------------------------------------ the path
    def text_encoder(self, x, x_lengths, e_enc, e_dur):
        B, Tx = x.shape
        x = x.detach().to(torch.int64).contiguous()
        x_lengths = x_lengths.detach().to(torch.int64).contiguous()
        e_enc, e_dur = self._f32(e_enc), self._f32(e_dur)
        if e_enc.shape[0] != B:
            e_enc, e_dur = e_enc.expand(B, -1).contiguous(), e_dur.expand(B, -1).contiguous()
        nf = self.hp.n_feats
        mu_x = torch.empty(B, nf, Tx, dtype=torch.float32, device=x.device)
        logw = torch.empty(B, 1, Tx, dtype=torch.float32, device=x.device)
        x_mask = torch.empty(B, 1, Tx, dtype=torch.float32, device=x.device)
        ws = self._workspace("enc", B, Tx)
        check(self.lib.mtts_text_encoder_forward(self.ctx, ptr(x), ptr(x_lengths), ptr(e_enc), ptr(e_dur), B, Tx, ptr(mu_x),
                                                 ptr(logw), ptr(x_mask), ws.data_ptr(), ws.numel(), stream_ptr()))
        return mu_x, logw, x_mask

    def speaker_embedding(self, table: int, ids: torch.Tensor) -> torch.Tensor:
        ids = ids.detach().to(torch.int64).contiguous()
        out = torch.empty(ids.numel(), self.hp.spk_emb_dim, dtype=torch.float32, device=ids.device)
        check(self.lib.mtts_speaker_embedding(self.ctx, table, ptr(ids), ids.numel(), ptr(out), stream_ptr()))
        return out

    def durations(self, logw, x_mask, scale_correction, length_scale):
        """``scale_correction`` / ``length_scale``: floats, or per-utterance sequences / tensors of B values."""
        logw, x_mask = self._f32(logw), self._f32(x_mask)
        B, _, Tx = logw.shape
        dur = torch.empty(B, Tx, dtype=torch.float32, device=logw.device)
        cum = torch.empty(B, Tx, dtype=torch.int32, device=logw.device)
        yfl = torch.empty(B, dtype=torch.int64, device=logw.device)
        if not (isinstance(scale_correction, (int, float)) and isinstance(length_scale, (int, float))):
            def per_utt(v):
                t = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
                t = t.expand(B) if t.numel() == 1 else t
                if t.numel() != B:
                    raise ValueError("per-utterance scale factors need one value per utterance")
                return t.to(logw.device).contiguous()
            sc, ls = per_utt(scale_correction), per_utt(length_scale)
            check(self.lib.mtts_durations_per_utterance(ptr(logw), ptr(x_mask), ptr(sc), ptr(ls), B, Tx, ptr(dur), ptr(cum),
                                                        ptr(yfl), stream_ptr()))
            return dur, cum, yfl
        check(self.lib.mtts_durations(ptr(logw), ptr(x_mask), float(scale_correction), float(length_scale), B, Tx, ptr(dur),
                                      ptr(cum), ptr(yfl), stream_ptr()))
        return dur, cum, yfl

    def align_pool(self, mu_x, cum, y_fine_lengths, t_pad: int):
        mu_x = self._f32(mu_x)
        B, nf, Tx = mu_x.shape
        mu_y = torch.empty(B, nf, t_pad, dtype=torch.float32, device=mu_x.device)
        y_mask = torch.empty(B, 1, t_pad, dtype=torch.float32, device=mu_x.device)
        y_len = torch.empty(B, dtype=torch.int64, device=mu_x.device)
        check(self.lib.mtts_align_pool(ptr(mu_x), ptr(cum), ptr(y_fine_lengths), B, nf, Tx, t_pad, ptr(mu_y), ptr(y_mask),
                                       ptr(y_len), stream_ptr()))
        return mu_y, y_mask, y_len

    def set_frame_limits(self, t_len: Optional[torch.Tensor]) -> None:
        """Per-utterance frame limits (int32 [B] on the device, even) for the following estimator calls; None clears."""
        if t_len is not None:
            if t_len.dtype != torch.int32 or not t_len.is_cuda or not t_len.is_contiguous():
                raise RuntimeError("mtts: frame limits must be a contiguous int32 tensor on the HIP device")
        self._t_len = t_len                     # keep it alive: the library stores the pointer
        check(self.lib.mtts_set_frame_limits(self.ctx, None if t_len is None else t_len.data_ptr()))

    def decoder_forward(self, x, mask, mu, t: float):
        x, mask, mu = self._f32(x), self._f32(mask), self._f32(mu)
        B, nf, T = x.shape
        out = torch.empty_like(x)
        ws = self._workspace("dec", B, T)
        check(self.lib.mtts_decoder_forward(self.ctx, ptr(x), ptr(mask), ptr(mu), float(t), B, T, ptr(out), ws.data_ptr(),
                                            ws.numel(), stream_ptr()))
        return out

    def fold_rows(self, y_max: int, align: int) -> int:
        """Rows per utterance the folded estimator needs for valid lengths up to y_max (mtts_fold_rows)."""
        n = self.lib.mtts_fold_rows(self.ctx, int(y_max), int(align))
        if n < 0:
            check(-1)
        return n

    def cfm_solve(self, x0, mu, mask, t_span, solver: str, add_mu: bool = False, t_out: Optional[int] = None,
                  out_scale: float = 1.0, out_shift: float = 0.0, y_lengths=None, y_max: Optional[int] = None,
                  t_fold: Optional[int] = None, ws: Optional[torch.Tensor] = None):
        """``y_lengths`` (int64 [B] on the device) + ``y_max`` + ``t_fold``: prefix masks on folded padding
        (mtts_cfm_solve_folded; ``mask`` is then not read).  ``ws``: caller-owned scratch (HIP-graph capture: the buffer must
        belong to the graph, not to the per-stream cache)."""
        x0, mu = self._f32(x0), self._f32(mu)
        B, nf, T = x0.shape
        if solver not in SOLVERS:
            raise ValueError(f"unsupported solver {solver!r} (euler, midpoint, rk4)")
        ts = np.ascontiguousarray(torch.as_tensor(t_span).detach().to("cpu", torch.float32).numpy())
        t_out = T if t_out is None else int(t_out)
        out = torch.empty(B, nf, t_out, dtype=torch.float32, device=x0.device)
        if t_fold is not None:
            y_lengths = y_lengths.detach().to(torch.int64).contiguous()
            if ws is None:
                ws = self._workspace("dec", B, int(t_fold))
            else:
                self._last_ws[("dec", stream_ptr())] = ws
            check(self.lib.mtts_cfm_solve_folded(self.ctx, ptr(x0), ptr(mu), ptr(y_lengths), int(y_max), int(bool(add_mu)),
                                                 ts.ctypes.data, len(ts) - 1, SOLVERS[solver], B, T, int(t_fold), ptr(out), t_out,
                                                 float(out_scale), float(out_shift), ws.data_ptr(), ws.numel(), stream_ptr()))
            return out
        mask = self._f32(mask)
        ws = self._workspace("dec", B, T)
        check(self.lib.mtts_cfm_solve(self.ctx, ptr(x0), ptr(mu), ptr(mask), int(bool(add_mu)), ts.ctypes.data, len(ts) - 1,
                                      SOLVERS[solver], B, T, ptr(out), t_out, float(out_scale), float(out_shift),
                                      ws.data_ptr(), ws.numel(), stream_ptr()))
        return out

    # ------------------------------------------------------------------ measurement
    def gemm_terms(self) -> int:
        return self.lib.mtts_gemm_terms(self.ctx)

    def prof_enable(self, on: bool) -> None:
        check(self.lib.mtts_prof_enable(self.ctx, int(on)))

    def prof_reset(self) -> None:
        check(self.lib.mtts_prof_reset(self.ctx))

    def prof_read(self, klass: int):
        n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        check(self.lib.mtts_prof_read(self.ctx, klass, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return n.value, ms.value, fl.value, by.value


    def prof_records(self, max_records: int = 1 << 16):
        """[(class, ms, flops, bytes)] per launch of the event pass, in launch order."""
        buf = (C.c_double * (4 * max_records))()
        n = self.lib.mtts_prof_records(self.ctx, buf, max_records)
        if n < 0:
            check(-1)
        return [(int(buf[4 * i]), buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n)]

    def prof_tags(self, max_bytes: int = 1 << 24):
        """Kernel instantiation name of each record of ``prof_records`` ("-" where the launcher does not tag)."""
        buf = C.create_string_buffer(max_bytes)
        n = self.lib.mtts_prof_tags(self.ctx, buf, max_bytes)
        if n < 0:
            check(-1)
        return buf.value.decode().split("\n")[:n]


# ---------------------------------------------------------------------- single kernels (used by the parity tests)
def gemm_f32(a, w, bias=None, *, B, T_in, T_out=None, tap_off=None, in_stride=1, a_mask=None, a_mean=None, a_rstd=None,
             a_part=None, act=0, p0=None, p1=None, res=None, out_mask=None, out_scale=1.0, stats_out=False, terms=-1):
    """a [B*T_in, C]; w Linear [N, C] or Conv1d [N, C, k]."""
    lib = load()
    N, Cc = w.shape[0], w.shape[1]
    ntaps = w.shape[2] if w.dim() == 3 else 1
    T_out = T_in if T_out is None else T_out
    taps = (C.c_int * ntaps)(*(tap_off if tap_off is not None else [j - ntaps // 2 for j in range(ntaps)]))
    packed = torch.empty(lib.mtts_gemm_packed_bytes(N, Cc, ntaps), dtype=torch.uint8, device=a.device)
    out = torch.empty(B * T_out, N, dtype=torch.float32, device=a.device)
    stats = torch.empty(B * T_out, N // 64, 2, dtype=torch.float32, device=a.device) if stats_out else None
    check(lib.mtts_gemm_f32(ptr(a), a.shape[1], B, T_in, Cc, ntaps, taps, in_stride, T_out, ptr(a_mask), ptr(a_mean), ptr(a_rstd),
                            ptr(a_part), a_part.shape[1] if a_part is not None else 0,
                            ptr(w.contiguous()), packed.data_ptr(), ptr(bias), N, act, ptr(p0), ptr(p1), ptr(res),
                            res.shape[1] if res is not None else 0, ptr(out_mask), float(out_scale), ptr(out), N, ptr(stats), terms,
                            stream_ptr()))
    return (out, stats) if stats_out else out


def gemm_p16(a, w, bias=None, *, B, T_in, T_out=None, tap_off=None, in_stride=1, a_mask=None, a_mean=None, a_rstd=None,
             a_part=None, act=0, p0=None, p1=None, res=None, out_mask=None, out_scale=1.0, stats_out=False, want_f32=True,
             want_p16=False, lscale=2048.0, force_bm=0):
    """P16-operand GEMM (csrc/gemm_p16.hip); a [B*T_in, C] fp32 is converted to its P16 image first.  Returns a dict."""
    lib = load()
    N, Cc = w.shape[0], w.shape[1]
    ntaps = w.shape[2] if w.dim() == 3 else 1
    T_out = T_in if T_out is None else T_out
    taps = (C.c_int * ntaps)(*(tap_off if tap_off is not None else [j - ntaps // 2 for j in range(ntaps)]))
    packed = torch.empty(lib.mtts_gemm_packed_bytes(N, Cc, ntaps), dtype=torch.uint8, device=a.device)
    scratch = torch.empty(lib.mtts_gemm_p16_scratch_bytes(B, T_in, Cc, T_out, N), dtype=torch.uint8, device=a.device)
    out = torch.empty(B * T_out, N, dtype=torch.float32, device=a.device) if want_f32 else None
    out16 = torch.empty(B * T_out, N, dtype=torch.float32, device=a.device) if want_p16 else None
    stats = torch.empty(B * T_out, N // 64, 2, dtype=torch.float32, device=a.device) if stats_out else None
    check(lib.mtts_gemm_p16(ptr(a), a.shape[1], B, T_in, Cc, ntaps, taps, in_stride, T_out, ptr(a_mask), ptr(a_mean), ptr(a_rstd),
                            ptr(a_part), a_part.shape[1] if a_part is not None else 0,
                            ptr(w.contiguous()), packed.data_ptr(), ptr(bias), N, act, ptr(p0), ptr(p1), ptr(res),
                            res.shape[1] if res is not None else 0, ptr(out_mask), float(out_scale), ptr(out), N,
                            ptr(out16), float(lscale), ptr(stats), force_bm, scratch.data_ptr(), stream_ptr()))
    return {"out": out, "out16": out16, "stats": stats}


def _host(t):
    """Host fp32 array of a tensor (or None) and the pointer ctypes passes for it."""
    if t is None:
        return None, None
    a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
    return a, a.ctypes.data


def tblock_chain(att, x, w_out, b_out, w1, b1, p0, p1, w2, b2, w_qkv=None, b_qkv=None, out_mask=None, qb=64, ch=128, repeat=0, pair=False):
    """Row-local chain of a transformer block (csrc/tblock_chain.hip, include/mtts.h mtts_tblock_chain).  att [M, inner] (or None:
    FeedForward only), x [M, C] on the device; panels / vectors anywhere (copied to the host).  Returns (x_out, qkv or None)."""
    lib = load()
    M, Cc = x.shape
    inner = att.shape[1] if att is not None else 0
    n_qkv = w_qkv.shape[0] if w_qkv is not None else 0
    n = lib.mtts_tblock_chain_scratch_bytes(M, Cc, inner, n_qkv, ch)
    if n < 0:
        raise RuntimeError("mtts_tblock_chain: unsupported shape")
    scratch = torch.empty(n, dtype=torch.uint8, device=x.device)
    x_out = torch.empty(M, Cc, dtype=torch.float32, device=x.device)
    qkv = torch.empty(M, n_qkv, dtype=torch.float32, device=x.device) if n_qkv else None
    keep = [_host(t) for t in (w_out, b_out, w1, b1, p0, p1, w2, b2, w_qkv, b_qkv)]
    hp = [k[1] for k in keep]
    ms = C.c_float(0.0)
    check((lib.mtts_tblock_chain_pair_timed if pair else lib.mtts_tblock_chain_timed)(ptr(att), ptr(x), M, Cc, inner, hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7], hp[8], hp[9],
                                      n_qkv, ptr(out_mask), qb, ch, ptr(x_out), ptr(qkv), scratch.data_ptr(), stream_ptr(), repeat,
                                      C.byref(ms)))
    torch.cuda.synchronize()
    if repeat:
        return x_out, qkv, ms.value
    return x_out, qkv


def attention_f32(qkv, mask, B, T, H, D, scale, mask_mode):
    lib = load()
    out = torch.empty(B * T, H * D, dtype=torch.float32, device=qkv.device)
    check(lib.mtts_attention_f32(ptr(qkv), ptr(mask), B, T, H, D, float(scale), mask_mode, ptr(out), stream_ptr()))
    return out


def attention_p16(qkv, mask, B, T, H, D, scale, mask_mode):
    lib = load()
    out = torch.empty(B * T, H * D, dtype=torch.float32, device=qkv.device)
    scratch = torch.empty(16 * B * T * H * D, dtype=torch.uint8, device=qkv.device)
    check(lib.mtts_attention_p16(ptr(qkv), ptr(mask), B, T, H, D, float(scale), mask_mode, ptr(out), scratch.data_ptr(), stream_ptr()))
    return out


def row_stats(x, eps=1e-5):
    lib = load()
    M, Cc = x.shape
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib.mtts_row_stats(ptr(x), M, Cc, Cc, eps, ptr(mean), ptr(rstd), stream_ptr()))
    return mean, rstd


def channel_layernorm(x, gamma, beta, B, T, *, act=0, film=None, mask=None, eps=1e-5):
    """x [B*T, C] rows; film [B, 2C] = gamma | beta of the DurationPredictor's speaker FiLM; mask [B*T]."""
    lib = load()
    y = torch.empty_like(x)
    check(lib.mtts_channel_layernorm(ptr(x), B, T, x.shape[1], ptr(gamma), ptr(beta), float(eps), act, ptr(film), ptr(mask), ptr(y),
                                     stream_ptr()))
    return y


def groupnorm_mish(y, gamma, beta, mask, B, T, G=8, eps=1e-5):
    lib = load()
    Cc = y.shape[1]
    scratch = torch.empty(lib.mtts_groupnorm_scratch_bytes(B, T, G), dtype=torch.uint8, device=y.device)
    out = torch.empty_like(y)
    check(lib.mtts_groupnorm_mish(ptr(y), ptr(gamma), ptr(beta), ptr(mask), B, T, Cc, G, eps, ptr(out), scratch.data_ptr(),
                                  stream_ptr()))
    return out
