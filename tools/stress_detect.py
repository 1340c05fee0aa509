"""Replay stress of the detect path: the head output of one batch must come out bit-identical launch after launch (every
conv kernel stages its tiles by LDS-DMA behind hand-counted waits).  usage (GPU box): python tools/stress_detect.py [replays]"""
import sys

import torch

sys.path.insert(0, ".")
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sd, meta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
bad = 0
for dtype, B, H, W, n in (("f16", 64, 640, 640, reps), ("f16", 16, 1280, 1280, reps // 2), ("f32", 16, 640, 640, reps // 4), ("f8", 16, 1280, 1280, reps // 2)):
    eng = engine_from_weights(sd, meta, dtype, 0, bgr_input=False)
    frames = torch.from_numpy(synth_frames(B, H, W, seed=5, kind="noise")).cuda()
    ref = eng.head_raw(frames).clone()
    d0 = [t.clone() for t in eng.detect(frames, 0.25, 0.7)]
    nb = 0
    for i in range(n):
        if not torch.equal(eng.head_raw(frames), ref):
            nb += 1
        d = eng.detect(frames, 0.25, 0.7)
        if not all(torch.equal(a, b) for a, b in zip(d, d0)):
            nb += 1
    print(dtype, B, H, W, "replays", n, "differing launches", nb)
    bad += nb
    del eng
sys.exit(1 if bad else 0)
