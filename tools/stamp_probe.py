#!/usr/bin/env python3
"""In-kernel cycle stamps of the persistent conv kernel (needs `csrc/build.sh ablate`).
usage: python tools/stamp_probe.py [--impl N] [--ablate A] [op_index ...]   (yolov8m 640x640 batch 64 f16)
impl 5 (conv_ws.h) stamps consumer waves (role 1) and producer waves (role 2) separately."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict  # noqa: E402

argv = sys.argv[1:]
impl = ablate = None
trace = 0
while argv and argv[0].startswith("--"):
    if argv[0] == "--impl": impl = int(argv[1])
    elif argv[0] == "--trace": trace = int(argv[1])
    elif argv[0] == "--ablate": ablate = int(argv[1])
    argv = argv[2:]
ops = [int(x) for x in argv] or [10, 21, 71, 18, 29]
sd, meta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
if impl is not None: eng.set_option("conv_impl", impl)
if ablate is not None: eng.set_option("ablate", ablate)
frames = torch.from_numpy(synth_frames(64, 640, 640, seed=1)).cuda()
eng.head_raw(frames); torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 256 * 8 * 8))()
for op in ops:
    eng.set_option("dbg_op", op)
    eng.head_raw(frames); torch.cuda.synchronize()
    rc = eng.lib.miyolo_debug_stamps(eng.h, buf)
    a = np.frombuffer(buf, dtype=np.uint64)[:256 * 64].reshape(256, 8, 8).astype(np.float64)
    o = eng.prog.ops[op]
    if impl == 5 and trace:
        tr = np.frombuffer(buf, dtype=np.uint64)[16384:].astype(np.int64)
        pt, ct = tr[:480 * 8].reshape(480, 8), tr[480 * 8:480 * 8 + 240 * 8].reshape(240, 8)
        print("   producer wave 4 of workgroup 0, per stage: start | slot granted | issued | published older | end   (cycles since wave start)")
        for i in range(min(trace, 480)):
            print("   P %3d  %8d %8d %8d %8d %8d" % (i, *pt[i, :5]))
        print("   consumer wave 0, per K step (2 stages): start | even stage full | even MFMAs done | odd stage full | odd MFMAs done")
        for i in range(min(trace // 2, 240)):
            print("   C %3d  %8d %8d %8d %8d %8d" % (i, *ct[i, :5]))
    if impl == 5:
        con, pro = a[:, :, 7] == 1, a[:, :, 7] == 2
        print(f"op {op} {o.name} k{o.ksize} cin {o.cin} cout {o.cout} down {o.down_out}: steps/wave {a[:, :, 5][con].mean():.1f} tiles/wg {a[:, :, 6][con].mean():.2f}")
        st, tl = a[:, :, 5][con].sum(), a[:, :, 6][con].sum()
        print(f"   consumers: total {a[:, :, 0][con].mean():.0f} | per K step: reads+mfma {a[:, :, 3][con].sum()/st:.0f}  wait for stage {a[:, :, 1][con].sum()/st:.0f} | epilogue per tile {a[:, :, 4][con].sum()/tl:.0f}")
        st = a[:, :, 5][pro].sum()
        print(f"   producers: total {a[:, :, 0][pro].mean():.0f} | per K step: dma issue {a[:, :, 2][pro].sum()/st:.0f}  vmcnt wait {a[:, :, 1][pro].sum()/st:.0f}  wait for slot {a[:, :, 3][pro].sum()/st:.0f} | tile setup per tile {a[:, :, 4][pro].sum()/a[:, :, 6][pro].sum():.0f}")
        continue
    ok = a[:, :, 7] > 0
    tot, wait, iss, comp, epi, steps, tiles = (a[:, :, k][ok] for k in range(7))
    print(f"op {op} {o.name} k{o.ksize} cin {o.cin} cout {o.cout} down {o.down_out}: waves {ok.sum()} steps/wave {steps.mean():.1f} tiles/wg {tiles.mean():.2f}")
    print(f"   cycles per wave: total {tot.mean():.0f} | per K step: wait+barrier {wait.sum()/steps.sum():.0f}  dma issue {iss.sum()/steps.sum():.0f}  "
          f"reads+mfma {comp.sum()/steps.sum():.0f} | epilogue per tile {epi.sum()/tiles.sum():.0f} | unaccounted {(tot-wait-iss-comp-epi).mean():.0f}")
