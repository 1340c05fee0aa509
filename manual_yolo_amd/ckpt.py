"""Checkpoint I/O: read an Ultralytics ``.pt`` without Ultralytics, store as safetensors.

The reference loads weights with ``YOLO("rank_classifier.pt")`` / ``YOLO("poker_model.pt")``
(reference ``detect.py:20-21``, ``pipe.py:147``, ``yolo.py:354``): a ``torch.save`` pickle of
``{'model': nn.Module, 'train_args': ..., ...}`` whose classes live in ``ultralytics.*``
and ``torchvision.*``.  Neither package is needed to read the numbers: a restricted
unpickler maps those classes to inert stand-ins and only lets ``torch``/``collections``/
an explicit allowlist of builtins through, so the file yields its state dict, ``names``, ``yaml`` spec and task.

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

# Explicit (module, name) allowlist: exactly the globals an Ultralytics 8.x checkpoint references (probe of
# rank_classifier.pt's pickle stream) plus what torch's own tensor rebuilding needs.  Anything else - builtins.eval,
# torch.utils.collect_env.run, torch.hub.load, os.system, ... - raises: a crafted .pt cannot execute code through this
# loader (tests/test_ckpt_safety.py).
_ALLOWED = {
    ("collections", "OrderedDict"),
    ("builtins", "set"), ("builtins", "slice"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "complex"), ("builtins", "frozenset"), ("builtins", "range"), ("builtins", "bytearray"),
    ("__builtin__", "set"), ("__builtin__", "slice"), ("__builtin__", "dict"), ("__builtin__", "list"),
    ("__builtin__", "tuple"), ("__builtin__", "int"), ("__builtin__", "float"), ("__builtin__", "bool"),
    ("__builtin__", "str"), ("__builtin__", "frozenset"), ("__builtin__", "range"),
    ("_codecs", "encode"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch", "Tensor"),
    ("torch.nn.parameter", "Parameter"), ("torch.serialization", "_get_layout"),
}
_ALLOWED_STORAGES = {"DoubleStorage", "FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
                     "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage"}
_ALLOWED_DTYPES = {"float16", "float32", "float64", "bfloat16", "int8", "int16", "int32", "int64", "uint8", "bool"}


class _Inert:
    """Stand-in for torchvision transform objects: keeps state, does nothing."""

    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


def _allowed_global(module: str, name: str):
    if (module, name) in _ALLOWED:
        return True
    if module == "torch" and (name in _ALLOWED_STORAGES or name in _ALLOWED_DTYPES):
        return True
    if module == "torch.storage" and name in ("_load_from_bytes", "UntypedStorage", "TypedStorage"):
        return name != "_load_from_bytes"      # _load_from_bytes unpickles again with the default unpickler: refused
    # plain torch.nn layer classes (Conv2d, BatchNorm2d, SiLU, Sequential, ...): classes defined under torch.nn.modules only
    if module.startswith("torch.nn.modules.") and name[:1].isupper():
        import importlib
        try:
            obj = getattr(importlib.import_module(module), name)
        except Exception:
            return False
        return isinstance(obj, type) and issubclass(obj, nn.Module)
    return False


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("ultralytics."):
            return type(name, (nn.Module,), {"__module__": module})
        if module.startswith("torchvision."):
            return type(name, (_Inert,), {"__module__": module})
        if _allowed_global(module, name):
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
