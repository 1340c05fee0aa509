"""The .pt reader must not execute code from a crafted checkpoint (the old module-prefix rule let
builtins.eval / torch.utils.collect_env.run through).  No GPU needed."""
import io
import pickle
import zipfile

import pytest
import torch

from manual_yolo_amd import ckpt


class _Evil:
    def __init__(self, module, name, args):
        self.m, self.n, self.a = module, name, args

    def __reduce__(self):
        import importlib
        return (getattr(importlib.import_module(self.m), self.n), self.a)


def _save_pt(tmp_path, payload_bytes):
    """A torch.save-style zip whose data.pkl is `payload_bytes`."""
    p = tmp_path / "evil.pt"
    with zipfile.ZipFile(p, "w") as z:
        z.writestr("archive/data.pkl", payload_bytes)
        z.writestr("archive/version", "3\n")
        z.writestr("archive/byteorder", "little")
    return str(p)


@pytest.mark.parametrize("module,name,args", [
    ("builtins", "eval", ("__import__('os').getpid()",)),
    ("builtins", "exec", ("x = 1",)),
    ("builtins", "__import__", ("os",)),
    ("builtins", "getattr", ("abc", "upper")),
    ("os", "system", ("true",)),
    ("subprocess", "check_output", (["true"],)),
    ("torch.utils.collect_env", "run", ("true",)),
    ("torch.hub", "load", ("x/y", "z")),
    ("torch.storage", "_load_from_bytes", (b"",)),
    ("torch", "load", ("/nonexistent",)),
])
def test_crafted_pickle_is_rejected(tmp_path, module, name, args):
    payload = pickle.dumps({"model": _Evil(module, name, args), "ema": None}, protocol=2)
    path = _save_pt(tmp_path, payload)
    with pytest.raises(pickle.UnpicklingError, match="blocked global"):
        ckpt.read_ultralytics_pt(path)


def test_find_class_allowlist_is_explicit():
    u = ckpt._RestrictedUnpickler(io.BytesIO(b""))
    assert u.find_class("collections", "OrderedDict") is __import__("collections").OrderedDict
    assert u.find_class("torch._utils", "_rebuild_tensor_v2") is torch._utils._rebuild_tensor_v2
    assert u.find_class("torch", "HalfStorage") is torch.HalfStorage
    assert u.find_class("torch.nn.modules.conv", "Conv2d") is torch.nn.Conv2d
    assert issubclass(u.find_class("ultralytics.nn.modules.conv", "Conv"), torch.nn.Module)      # inert stand-in
    for module, name in [("builtins", "eval"), ("builtins", "open"), ("torch", "load"), ("torch.nn.modules.module", "register_module_forward_hook"),
                         ("torch.utils.collect_env", "run"), ("numpy", "load"), ("posix", "system")]:
        with pytest.raises(pickle.UnpicklingError):
            u.find_class(module, name)
