#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference's artefacts.

Run in the build container only (needs /root/reference; the GPU box has neither it nor
any need for it):

    python tools/make_golden.py

Inputs (all data files, no reference source):
  /root/reference/rank_classifier.pt                        (== runs/.../weights/best.pt)
  /root/reference/runs/rank_classifier/weights/last.pt
  /root/reference/rank_classifier/valid/<class>/*.jpg       (67 crops)
Outputs:
  tests/golden/rank_best.safetensors, rank_last.safetensors  raw fp16 tensors + meta
  tests/golden/rank_valid.npz   raw RGB crops (ragged), preprocessed u8 64x64x3, labels,
                                oracle logits/probs for best and last (fp32, BN fused)
The known answers these must reproduce (checked in tests/test_oracle_kat.py) come from the
reference itself: runs/rank_classifier/results.csv:21-22, confusion_matrix.png.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from manual_yolo_amd.ckpt import read_ultralytics_pt, save_bundle  # noqa: E402
from oracle.pre_ref import classify_transform  # noqa: E402
from oracle.yolo_ref import RefYolo  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from PIL import Image
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    bundles = {}
    for tag, p in (("best", f"{REF}/rank_classifier.pt"),
                   ("last", f"{REF}/runs/rank_classifier/weights/last.pt")):
        sd, meta = read_ultralytics_pt(p)
        save_bundle(os.path.join(OUT, f"rank_{tag}.safetensors"), sd, meta)
        bundles[tag] = (sd, meta)
        print(tag, meta["task"], meta["nc"], meta["scale"], meta["bn_eps"], meta["imgsz"], len(sd))

    names = bundles["best"][1]["names"]
    name_to_idx = {v: k for k, v in names.items()}
    raw, shapes, pre, labels, files = [], [], [], [], []
    vdir = f"{REF}/rank_classifier/valid"
    for cls in sorted(os.listdir(vdir)):
        for fn in sorted(os.listdir(os.path.join(vdir, cls))):
            im = np.asarray(Image.open(os.path.join(vdir, cls, fn)).convert("RGB"), dtype=np.uint8)
            raw.append(im.reshape(-1)); shapes.append(im.shape[:2])
            pre.append(classify_transform(im, 64))
            labels.append(name_to_idx[cls]); files.append(f"{cls}/{fn}")
    pre = np.stack(pre)
    x = torch.from_numpy(pre).permute(0, 3, 1, 2).float() / 255.0
    out = {"raw_rgb_flat": np.concatenate(raw), "raw_shapes": np.asarray(shapes, np.int32),
           "pre_u8": pre, "labels": np.asarray(labels, np.int32), "files": np.asarray(files)}
    for tag, (sd, meta) in bundles.items():
        ref = RefYolo(sd, "classify", meta["nc"], meta["scale"], meta["bn_eps"], fuse=True)
        probs, logits = ref.forward(x)
        top1 = probs.argmax(1).numpy()
        print(tag, "top1 acc", (top1 == np.asarray(labels)).sum(), "/", len(labels))
        out[f"logits_{tag}"] = logits.numpy().astype(np.float32)
        out[f"probs_{tag}"] = probs.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "rank_valid.npz"), **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
