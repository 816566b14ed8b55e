"""Checkpoint formats of the mel-synthesis path.

The reference ships Lightning checkpoints: a pickled dict with ``hyper_parameters`` (OmegaConf containers) and a
``state_dict`` whose keys may carry ``_orig_mod.`` infixes from ``torch.compile`` (reference inference.py:186-197,
matcha/utils/prepare_ckpt_for_release.py).  Unpickling those needs lightning / omegaconf on the loading side.

``convert_lightning_checkpoint`` rewrites one, once, on a machine that can unpickle it, into a directory that needs neither:

    <out>/model.safetensors   flat fp32 tensors under the reference's state-dict names (``_orig_mod.`` removed), only the
                              tensors of the inference path (encoder, speaker tables, decoder estimator, mel statistics)
    <out>/hparams.json        the flattened path hyper-parameters (hparams.PathHParams) + format version

    <out>/packed-<key>.bin    (written on first load on a GPU box) the library's packed weight image -- GEMM panels, fp16
                              planes, fragment streams -- so that later loads skip the ~7 s host packing pass; <key> digests
                              the library's layout signature (ABI / image revision, architecture, arithmetic, layout
                              switches) and the tensors themselves, so a stale file is simply never looked up

``load_converted`` reads that directory back; ``inference.load_matcha`` accepts either form.
"""
from __future__ import annotations

import dataclasses
import json
from pathlib import Path
from typing import Dict, Tuple

import torch

from . import hparams as H
from .synthetic import state_dict_spec

FORMAT_VERSION = 1
WEIGHTS = "model.safetensors"
HPARAMS = "hparams.json"


def _plain(obj):
    """OmegaConf containers -> plain python, when omegaconf is importable."""
    try:
        from omegaconf import OmegaConf  # type: ignore
        if OmegaConf.is_config(obj):
            return OmegaConf.to_container(obj, resolve=True)
    except Exception:
        pass
    return obj


def strip_compile_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``a._orig_mod.b`` -> ``a.b`` (keys written by a model whose sub-modules went through torch.compile)."""
    return {k.replace("_orig_mod.", ""): v for k, v in sd.items()}


def hparams_to_json(hp: H.PathHParams) -> dict:
    d = dataclasses.asdict(hp)
    d["decoder"]["channels"] = list(d["decoder"]["channels"])
    return {"format_version": FORMAT_VERSION, "path_hparams": d}


def hparams_from_json(d: dict) -> H.PathHParams:
    if d.get("format_version") != FORMAT_VERSION:
        raise ValueError(f"unsupported converted-checkpoint version {d.get('format_version')!r}")
    p = dict(d["path_hparams"])
    enc = H.EncoderHParams(**p.pop("encoder"))
    dec_d = dict(p.pop("decoder"))
    dec_d["channels"] = tuple(dec_d["channels"])
    dec = H.DecoderHParams(**dec_d)
    return H.PathHParams(encoder=enc, decoder=dec, **p)


def select_path_tensors(hp: H.PathHParams, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The tensors the path reads, checked against the spec (a missing or misshaped one is an error here, not at run time)."""
    out = {}
    for key, shape, _kind in state_dict_spec(hp):
        if key not in sd:
            raise KeyError(f"checkpoint has no tensor {key!r}")
        t = sd[key]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(t.shape)} != expected {tuple(shape)}")
        out[key] = t.detach().to(torch.float32).contiguous().cpu()
    return out


def convert_lightning_checkpoint(ckpt_path, out_dir) -> Path:
    """One-shot conversion (needs whatever the checkpoint's pickle needs: lightning / omegaconf for real ones)."""
    from safetensors.torch import save_file
    ckpt = torch.load(str(ckpt_path), map_location="cpu", weights_only=False)
    kw = dict(_plain(ckpt["hyper_parameters"]))
    kw.pop("optimizer", None)
    kw.pop("scheduler", None)
    hp = H.from_reference_kwargs(**kw)
    sd = select_path_tensors(hp, strip_compile_prefix(ckpt["state_dict"]))
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    save_file(sd, str(out / WEIGHTS), metadata={"format": "matcha-tts-24k_amd", "version": str(FORMAT_VERSION)})
    (out / HPARAMS).write_text(json.dumps(hparams_to_json(hp), indent=1, sort_keys=True))
    return out


def is_converted(path) -> bool:
    p = Path(path)
    return p.is_dir() and (p / WEIGHTS).exists() and (p / HPARAMS).exists()


def load_converted(path) -> Tuple[H.PathHParams, Dict[str, torch.Tensor]]:
    from safetensors.torch import load_file
    p = Path(path)
    hp = hparams_from_json(json.loads((p / HPARAMS).read_text()))
    sd = load_file(str(p / WEIGHTS), device="cpu")
    return hp, select_path_tensors(hp, sd)


class packed_cache:
    """The packed-image cache file of one (library signature, tensors) pair inside a converted checkpoint's directory."""
    MAGIC = b"MTTSIMG1"

    def __init__(self, directory, signature: str, sd: Dict[str, torch.Tensor]):
        import hashlib
        h = hashlib.sha256(signature.encode())
        for k in sorted(sd):
            t = sd[k].detach().to("cpu", torch.float32).contiguous()
            h.update(k.encode())
            h.update(str(tuple(t.shape)).encode())
            h.update(t.numpy().tobytes())
        self.signature = signature
        self.path = Path(directory) / f"packed-{h.hexdigest()[:24]}.bin"

    def read(self):
        """{"data": uint8 array, "saturates": bool} or None (no file / damaged file)."""
        import numpy as np
        try:
            with open(self.path, "rb") as f:
                head = f.read(24)
                if len(head) != 24 or head[:8] != self.MAGIC:
                    return None
                nbytes = int.from_bytes(head[8:16], "little")
                sat = int.from_bytes(head[16:24], "little")
                data = np.fromfile(f, dtype=np.uint8, count=nbytes)
            return {"data": data, "saturates": bool(sat)} if data.size == nbytes else None
        except OSError:
            return None

    def write(self, image, saturates: bool) -> None:
        """Best effort (a read-only model directory is not an error); written under a temporary name, then renamed."""
        import os
        tmp = self.path.with_suffix(".tmp%d" % os.getpid())
        try:
            with open(tmp, "wb") as f:
                f.write(self.MAGIC + int(image.nbytes).to_bytes(8, "little") + int(bool(saturates)).to_bytes(8, "little"))
                image.tofile(f)
            os.replace(tmp, self.path)
        except OSError:
            try:
                os.unlink(tmp)
            except OSError:
                pass


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Lightning checkpoint -> flat safetensors + JSON hyper-parameters")
    ap.add_argument("checkpoint")
    ap.add_argument("out_dir")
    a = ap.parse_args(argv)
    print(convert_lightning_checkpoint(a.checkpoint, a.out_dir))


if __name__ == "__main__":
    main()
