#!/usr/bin/env python3
"""Per-layer error of the fp8 engine against the f16 engine (same weights, same frames): relative RMS error of every op's
output slice.  usage: python tools/fp8_layer_error.py [imgsz] [scale]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict  # noqa: E402

sz = int(sys.argv[1]) if len(sys.argv) > 1 else 640
scale = sys.argv[2] if len(sys.argv) > 2 else "m"
sd, meta = synth_state_dict("detect", 64, scale, 0), synth_meta("detect", 64, scale)
calib = torch.from_numpy(np.concatenate([synth_frames(4, sz, sz, seed=101), synth_frames(2, sz, sz, seed=102, kind="blocks")]))
e8 = engine_from_weights(sd, meta, "f8", 0, bgr_input=False, calib_frames=calib)
e16 = engine_from_weights(sd, meta, "f16", 0, bgr_input=False, fuse_head=False)   # same op list as the fp8 program
frames = torch.from_numpy(synth_frames(2, sz, sz, seed=1)).cuda()
y8 = e8.head_raw(frames); y16 = e16.head_raw(frames)
for i, op in enumerate(e16.prog.ops):
    if op.dst is None:
        continue
    a16 = e16.read_buffer(op.dst.buf, 2, sz, sz)[..., op.dst.ch_off:op.dst.ch_off + op.dst.ch_cnt]
    a8 = e8.read_buffer(op.dst.buf, 2, sz, sz)[..., op.dst.ch_off:op.dst.ch_off + op.dst.ch_cnt]
    rel = float((a8 - a16).norm() / (a16.norm() + 1e-12))
    gain = float((a8 * a16).sum() / ((a16 * a16).sum() + 1e-12))
    sc = e8.quant.out_scale.get(i, 0.0)
    print(f"op {i:3d} {op.name:22s} k{op.ksize} {op.cin:4d}->{op.cout:4d} rel rms err {rel:.4f} gain {gain:.4f}  |x|max f16 {float(a16.abs().max()):8.3f} fp8 {float(a8.abs().max()):8.3f}  out_scale*448 {sc * 448:8.3f}")
print("head y: max |dscore|", float((y8[:, 4:] - y16[:, 4:]).abs().max()), "mean", float((y8[:, 4:] - y16[:, 4:]).abs().mean()))
