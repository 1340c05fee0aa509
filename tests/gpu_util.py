"""Helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np
import torch

from manual_yolo_amd.arch import OP_CONV, OP_MAXPOOL5, Op, Program, View, WeightRecipe
from manual_yolo_amd.engine import Engine


def conv_program(srcs, cout, k, s, act, with_res, dst_ld=None, dst_off=0, out_f32=False):
    """One CONV op.  srcs: list of (ld, ch_off, ch_cnt, up).  Buffers: 0 input(u8, unused),
    1..n sources, then dst, then residual."""
    bufs = [(3, 1, -2)]
    views = []
    for (ld, off, cnt, up) in srcs:
        bufs.append((ld, 2 if up else 1, -1))
        views.append(View(len(bufs) - 1, off, cnt, up))
    bufs.append((dst_ld or cout, s, 0 if out_f32 else -1))
    dst = View(len(bufs) - 1, dst_off, cout)
    res = None
    if with_res:
        bufs.append((cout, s, -1))
        res = View(len(bufs) - 1, 0, cout)
    cin = sum(v.ch_cnt for v in views)
    op = Op(OP_CONV, k, s, int(act), cin, cout, views, dst, res, 0, 1, name="t", down_in=1, down_out=s)
    w = [WeightRecipe("conv", "t", False, tuple(v.ch_cnt for v in views)), WeightRecipe("bias", "t", False)]
    return Program("classify", 1, bufs, [op], w, 2, {})


def run_conv(dtype, x_list, w, b, srcs, k, s, act, res=None, B=1, H=8, W=8, force=None, dst_ld=None, dst_off=0,
             out_f32=False, impl=1, opts=None):
    """x_list: fp32 NHWC arrays for the source buffers.  Returns fp32 [B,Ho,Wo,dst_ld]."""
    cout = w.shape[0]
    prog = conv_program(srcs, cout, k, s, act, res is not None, dst_ld, dst_off, out_f32)
    sd = {"t.weight": torch.from_numpy(w), "t.bias": torch.from_numpy(b)}
    eng = Engine(prog, sd, 1e-3, dtype, 0)
    try:
        eng.set_option("conv_impl", impl)
    except Exception as e:           # experimental kernels (conv_impl 2, 4, 5, 6) are not in the shipped build
        if "experimental" in str(e):
            import pytest
            pytest.skip(str(e))
        raise
    for k_, v_ in (opts or {}).items():
        eng.set_option(k_, v_)
    if force:
        eng.set_option("force_wc", force[0]); eng.set_option("force_tc", force[1])
    for i, x in enumerate(x_list):
        eng.write_buffer(1 + i, torch.from_numpy(x), H, W)
    nb = len(prog.bufs)
    dst_buf = nb - 2 if res is not None else nb - 1
    if dst_ld:   # pre-fill so untouched channels are checkable
        eng.write_buffer(dst_buf, torch.full(eng.buffer_shape(dst_buf, B, H, W), 7.0), H, W)
    if res is not None:
        eng.write_buffer(nb - 1, torch.from_numpy(res), H, W)
    eng.run_ops(0, 1, None, B, H, W)
    y = eng.read_buffer(dst_buf, B, H, W)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def q(x, dtype):
    """Round to the activation dtype (what the GPU buffers hold)."""
    return x.astype(np.float16).astype(np.float32) if dtype == "f16" else x.astype(np.float32)


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
