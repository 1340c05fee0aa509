#!/usr/bin/env python3
"""Letterboxed real frames as detection fixtures ("letterboxed real frames from
roadmap1.v3i.yolov8/valid/images precomputed here (a few, as fixtures) so NMS sees clustered boxes").

Reads DATA files of the reference only (JPEG screenshots and their YOLO label files), decodes them with Pillow, converts
RGB -> BGR (the reference passes cv2/mss BGR frames, detect.py:535-541), letterboxes them with the oracle's LetterBox
restatement (oracle/pre_ref.py, rect padding to a multiple of 32 as `auto=True` does) and writes
tests/golden/real_frames.npz:  frames uint8 [N, 384, 640, 3] (BGR),  orig_hw [N, 2],  labels (object array flattened:
label_frame [M], label_rows [M, 5] = cls cx cy w h normalised, unlabel.py:44,54-57).  Run in the build container only
(the reference is not on the GPU box); the .npz travels.   usage: python tools/make_frames_golden.py [n_frames]"""
import glob
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pre_ref import letterbox  # noqa: E402

REF = "/root/reference/roadmap1.v3i.yolov8/valid"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
files = sorted(glob.glob(os.path.join(REF, "images", "*.jpg")))
step = max(1, len(files) // n)
files = [f for f in files if Image.open(f).size == (1600, 900)]          # one shape -> one batch (1920 x 1200 frames letterbox to 416 rows)
pick = files[::max(1, len(files) // n)][:n]
frames, orig, lf, lr = [], [], [], []
for i, f in enumerate(pick):
    rgb = np.asarray(Image.open(f).convert("RGB"), dtype=np.uint8)
    bgr = np.ascontiguousarray(rgb[..., ::-1])
    out, _ = letterbox(bgr, (640, 640), auto=True, stride=32)
    frames.append(out)
    orig.append(rgb.shape[:2])
    lab = os.path.join(REF, "labels", os.path.splitext(os.path.basename(f))[0] + ".txt")
    if os.path.exists(lab):
        for line in open(lab):
            p = line.split()
            if len(p) == 5:
                lf.append(i); lr.append([float(v) for v in p])
shapes = {x.shape for x in frames}
assert len(shapes) == 1, shapes
out = os.path.join(ROOT, "tests", "golden", "real_frames.npz")
np.savez_compressed(out, frames=np.stack(frames), orig_hw=np.array(orig, np.int32), label_frame=np.array(lf, np.int32),
                    label_rows=np.array(lr, np.float32))
print("wrote", out, np.stack(frames).shape, f"{os.path.getsize(out) / 1e6:.2f} MB,", len(lr), "label boxes")
