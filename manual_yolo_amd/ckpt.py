"""Checkpoint I/O: read an Ultralytics ``.pt`` without Ultralytics, store as safetensors.

The reference loads weights with ``YOLO("rank_classifier.pt")`` / ``YOLO("poker_model.pt")``
(reference ``detect.py:20-21``, ``pipe.py:147``, ``yolo.py:354``): a ``torch.save`` pickle of
``{'model': nn.Module, 'train_args': ..., ...}`` whose classes live in ``ultralytics.*``
and ``torchvision.*``.  Neither package is needed to read the numbers: a restricted
unpickler maps those classes to inert stand-ins and only lets ``torch``/``collections``/
builtins through, so the file yields its state dict, ``names``, ``yaml`` spec and task.

``save_bundle``/``load_bundle`` keep the same information as one ``.safetensors`` file
(raw tensors, original dtypes) with the metadata JSON in its header - that is the format
that travels to GPU machines (no pickle, no third-party class paths).
"""
from __future__ import annotations

import json
import pickle
from typing import Dict, Tuple

import torch
import torch.nn as nn

_ALLOWED_EXACT = {"collections", "__builtin__", "builtins", "_codecs",
                  "numpy", "numpy.core.multiarray", "numpy._core.multiarray"}


class _Inert:
    """Stand-in for torchvision transform objects: keeps state, does nothing."""

    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("ultralytics."):
            return type(name, (nn.Module,), {"__module__": module})
        if module.startswith("torchvision."):
            return type(name, (_Inert,), {"__module__": module})
        if module == "torch" or module.startswith("torch.") or module in _ALLOWED_EXACT:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"blocked global {module}.{name}")


class _PickleModule:
    __name__ = "manual_yolo_amd_restricted_pickle"
    Unpickler = _RestrictedUnpickler

    @staticmethod
    def load(f, **kw):
        return _RestrictedUnpickler(f, **kw).load()


def _bn_eps(model: nn.Module) -> float:
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            return float(m.eps)
    return 1e-3


def read_ultralytics_pt(path: str) -> Tuple[Dict[str, torch.Tensor], dict]:
    """Returns (state_dict with original dtypes, meta).

    meta: task ('detect'|'classify'), nc, scale, names {int: str}, bn_eps, imgsz, spec
    (the yaml dict embedded in the checkpoint), version."""
    ck = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    model = ck["ema"] if ck.get("ema") is not None else ck["model"]
    cls_name = type(model).__name__
    task = {"ClassificationModel": "classify", "DetectionModel": "detect"}.get(cls_name)
    if task is None:
        raise ValueError(f"unsupported Ultralytics model class {cls_name}")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = getattr(model, "names", None) or {}
    if isinstance(names, (list, tuple)):
        names = dict(enumerate(names))
    spec = getattr(model, "yaml", {}) or {}
    train_args = ck.get("train_args") or {}
    imgsz = train_args.get("imgsz", 640 if task == "detect" else 224)
    meta = {
        "task": task,
        "nc": int(spec.get("nc", len(names))),
        "scale": spec.get("scale", "n"),
        "names": {int(k): str(v) for k, v in names.items()},
        "bn_eps": _bn_eps(model),
        "imgsz": int(imgsz if isinstance(imgsz, int) else imgsz[0]),
        "spec": {k: spec[k] for k in ("scales", "backbone", "head") if k in spec},
        "version": ck.get("version", ""),
        "nc_quirk": True,
    }
    return sd, meta


def save_bundle(path: str, sd: Dict[str, torch.Tensor], meta: dict) -> None:
    from safetensors.torch import save_file
    m = dict(meta)
    m["names"] = {str(k): v for k, v in meta["names"].items()}
    save_file({k: v.contiguous() for k, v in sd.items()}, path, metadata={"manual_yolo_amd": json.dumps(m)})


def load_bundle(path: str) -> Tuple[Dict[str, torch.Tensor], dict]:
    from safetensors import safe_open
    sd = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = json.loads(f.metadata()["manual_yolo_amd"])
        for k in f.keys():
            sd[k] = f.get_tensor(k)
    meta["names"] = {int(k): v for k, v in meta["names"].items()}
    return sd, meta


def load_weights(path: str) -> Tuple[Dict[str, torch.Tensor], dict]:
    """``.safetensors`` bundle or Ultralytics ``.pt`` (a sibling ``.safetensors`` wins)."""
    import os
    if path.endswith(".safetensors"):
        return load_bundle(path)
    sib = os.path.splitext(path)[0] + ".safetensors"
    if os.path.exists(sib):
        return load_bundle(sib)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return read_ultralytics_pt(path)
