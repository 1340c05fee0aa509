#!/usr/bin/env python3
"""In-kernel cycle stamps of the halo-slab conv kernel (conv_h2.h); needs a stamps build:
   OUT=libmiyolo_stamps.so csrc/build.sh stamps ; MIYOLO_LIB=.../libmiyolo_stamps.so python tools/stamp_h2.py [op ...]
(yolov8m 640x640 batch 64 f16).  Per wave of the first 512 workgroups: total cycles, and per tap step the cycles in
[vmcnt(0) + barrier], [weight DMA issue], [fragment reads + MFMAs]; per chunk the slab barrier + issue; the epilogue."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict  # noqa: E402

argv = sys.argv[1:]
opts = [x for x in argv if "=" in x]
ops = [int(x) for x in argv if "=" not in x] or [10, 21, 71]
sd, meta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
for kv in opts:
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
frames = torch.from_numpy(synth_frames(64, 640, 640, seed=1)).cuda()
eng.head_raw(frames); torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 256 * 8 * 8))()
for op in ops:
    eng.set_option("dbg_op", op)
    eng.head_raw(frames); torch.cuda.synchronize()
    eng.lib.miyolo_debug_stamps(eng.h, buf)
    a = np.frombuffer(buf, dtype=np.uint64)[:512 * 4 * 8].reshape(512, 4, 8).astype(np.float64)
    ntile = a[:, :, 7]
    o = eng.prog.ops[op]
    ok = a[:, :, 7] > 0
    tot, wait, iss, comp, slab, epi, steps, sbar = (a[:, :, k][ok] for k in range(8))
    nt = float(ok.sum()); tiles = np.ones_like(tot)
    print(f"op {op} {o.name} cin {o.cin} cout {o.cout} down {o.down_out}: waves {ok.sum()} tiles/workgroup {tiles.mean():.2f} tap steps/tile {steps.sum()/nt:.0f}")
    print(f"   cycles per wave per TILE: total {tot.sum()/nt:.0f} | per tap step: wait+barrier {wait.sum()/steps.sum():.0f}  weight dma issue {iss.sum()/steps.sum():.0f}  "
          f"reads+mfma {comp.sum()/steps.sum():.0f} | slab barrier per tile {sbar.sum()/nt:.0f} + slab issue per tile {slab.sum()/nt:.0f} | epilogue(+next slab issue) {epi.sum()/nt:.0f} | "
          f"prologue+unaccounted {(tot-wait-iss-comp-slab-epi).sum()/nt:.0f}")
    e = np.frombuffer(buf, dtype=np.uint64)[16384:16384 + 512 * 4 * 4].reshape(512, 4, 4).astype(np.float64)
    print(f"   prologue: setup {e[:, :, 0][ok].mean():.0f} | slab + weight DMA issue {e[:, :, 1][ok].mean():.0f} | bias + wait until everything landed {e[:, :, 2][ok].mean():.0f}")
    t0 = e[:, :, 3][ok]
    print(f"   workgroup start times (cycles after the first): median {np.median(t0 - t0.min()):.0f}  90% {np.percentile(t0 - t0.min(), 90):.0f}  max {(t0 - t0.min()).max():.0f}")
