"""CPU-only: the C-ABI library builds/loads and exports exactly what include/miyolo.h declares;
host-side packing agrees with the library's layout constants."""
import ctypes
import os
import re

import pytest
import torch

from manual_yolo_amd import engine
from manual_yolo_amd.arch import build_program
from manual_yolo_amd.synth import synth_state_dict
from manual_yolo_amd.weights import K_ALIGN, build_weight_tensors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "miyolo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(miyolo_[a-z_0-9]+)\s*\(", src))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(engine.lib_path()):
        import __graft_entry__ as g
        g.build()
    return engine.load_library()


def test_header_and_binding_agree():
    assert _header_functions() == set(engine.SYMBOLS)


def test_library_exports_every_symbol(lib):
    raw = ctypes.CDLL(engine.lib_path())
    for name in _header_functions():
        assert hasattr(raw, name), name
    assert lib.miyolo_abi_version() == 1
    assert lib.miyolo_k_align(0) == K_ALIGN["f32"] and lib.miyolo_k_align(1) == K_ALIGN["f16"]


def test_struct_sizes_match_header():
    # miyolo_view 4 ints; miyolo_op = 7 + 3*4 + 4 + 4 + 2 + 3 + 5 ints
    assert ctypes.sizeof(engine._View) == 16
    assert ctypes.sizeof(engine._Buf) == 16
    assert ctypes.sizeof(engine._Op) == 4 * (7 + 12 + 4 + 4 + 2 + 3 + 5)
    assert ctypes.sizeof(engine._Desc) == 4 * 16


def test_create_without_gpu_fails_loudly(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    prog = build_program("classify", 13, "n")
    with pytest.raises(engine.MiyoloError):
        engine.Engine(prog, synth_state_dict("classify", 13, "n"), 1e-5, "f16")


def test_program_and_weight_layouts():
    prog = build_program("detect", 64, "m", nc_quirk=False)
    assert prog.conv_macs_per_image(640, 640) == 39_442_022_400      # 39.442 GMAC
    assert sum(1 for o in prog.ops if o.kind in (0, 1)) == 83        # 83 convolutions
    assert prog.strides == (8, 16, 32) and prog.max_stride == 32
    sd = synth_state_dict("detect", 64, "m")
    for dt in ("f16", "f32"):
        ws = build_weight_tensors(prog, sd, 1e-3, dt)
        assert len(ws) == len(prog.weights)
        for r, w in zip(prog.weights, ws):
            if r.kind == "conv":
                assert w.shape[1] % K_ALIGN[dt] == 0 and w.dtype == (torch.float16 if dt == "f16" else torch.float32)
            if r.kind == "stem":
                assert tuple(w.shape) == (48, 32)
        # scalar bias loads run up to a widest channel tile past cout: every bias carries a multiple of 128 + 256 floats
        for op in prog.ops:
            if op.kind in (0, 1):
                assert ws[op.bias].numel() >= (op.cout + 127) // 128 * 128 + 256 and float(ws[op.bias][op.cout:].abs().max()) == 0.0
    # upsample / concat never become ops: only 3 op kinds besides conv appear
    assert {o.kind for o in prog.ops} == {0, 1, 2, 3}
    cls = build_program("classify", 13, "n")
    assert sum(1 for o in cls.ops if o.kind in (0, 1)) == 26 and cls.conv_macs_per_image(64, 64) > 16_000_000
