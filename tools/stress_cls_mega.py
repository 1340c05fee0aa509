"""Replay stress of the one-launch classifier against the layered path (bit-identity), for hunting rare LDS-DMA hazards.
usage (GPU box): python tools/stress_cls_mega.py [rounds]"""
import sys

import torch

sys.path.insert(0, ".")
from manual_yolo_amd.ckpt import load_bundle  # noqa: E402
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sd, meta = load_bundle("tests/golden/rank_best.safetensors")
eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
g = torch.Generator().manual_seed(3)
bad = tot = 0
for B in (67, 256, 1, 1024, 300):
    x = torch.randint(0, 256, (B, 64, 64, 3), dtype=torch.uint8, generator=g).cuda()
    eng.set_option("cls_mega", 0)
    ref = eng.classify(x)[0].clone()
    eng.set_option("cls_mega", 1)
    for r in range(rounds if B > 1 else rounds * 10):
        out = eng.classify(x)[0]
        nb = int(((out - ref).abs().max(1).values > 0).sum())
        bad += nb; tot += B
        if nb:
            print("batch", B, "replay", r, "images differing", nb)
print("images checked", tot, "differing", bad)
sys.exit(1 if bad else 0)
