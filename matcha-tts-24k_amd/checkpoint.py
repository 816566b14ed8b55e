"""Checkpoint formats of the mel-synthesis path.

The reference ships Lightning checkpoints: a pickled dict with ``hyper_parameters`` (OmegaConf containers) and a
``state_dict`` whose keys may carry ``_orig_mod.`` infixes from ``torch.compile`` (reference inference.py:186-197,
matcha/utils/prepare_ckpt_for_release.py).  Unpickling those needs lightning / omegaconf on the loading side.

``convert_lightning_checkpoint`` rewrites one, once, on a machine that can unpickle it, into a directory that needs neither:

    <out>/model.safetensors   flat fp32 tensors under the reference's state-dict names (``_orig_mod.`` removed), only the
                              tensors of the inference path (encoder, speaker tables, decoder estimator, mel statistics)
    <out>/hparams.json        the flattened path hyper-parameters (hparams.PathHParams) + format version

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


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Lightning checkpoint -> flat safetensors + JSON hyper-parameters")
    ap.add_argument("checkpoint")
    ap.add_argument("out_dir")
    a = ap.parse_args(argv)
    print(convert_lightning_checkpoint(a.checkpoint, a.out_dir))


if __name__ == "__main__":
    main()
