#!/usr/bin/env python3
"""Throughput of the device LetterBox (miyolo_letterbox): B frames of the reference's capture size
(detect.py:18, 930 x 1130 BGR uint8, resident in HBM) -> 640-letterboxed uint8.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.preprocess import letterbox_batch_gpu  # noqa: E402

B, h, w = 64, 930, 1130
x = torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    y = letterbox_batch_gpu(x, (640, 640), 32, auto=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    y = letterbox_batch_gpu(x, (640, 640), 32, auto=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
by = x.numel() + y.numel()
print(json.dumps({"op": "letterbox 930x1130 -> %dx%d" % tuple(y.shape[1:3]), "batch": B, "ms": round(ms, 4),
                  "frames_per_s": round(B / ms * 1e3), "algorithmic_GBps": round(by / ms / 1e6, 1), "hbm_peak_GBps": 8000}))
