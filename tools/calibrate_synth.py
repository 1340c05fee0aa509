#!/usr/bin/env python3
"""Choose the per-layer gains of the synthetic (random-init) detect model.

Random conv weights at a single global gain either collapse to constants or blow up after
~80 layers, so each conv gets a scalar gain that makes its raw output unit-variance on a
calibration batch (two seed-1 uniform-noise frames), layer by layer (LSUV style), and the
Detect class branch gets a bias that lets a few hundred anchors per frame clear conf=0.25.
Writes manual_yolo_amd/synth_gains.json (data; regenerate only if synth.py changes).

Uses the CPU oracle for the forward passes: this is offline tooling, not the product path.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manual_yolo_amd.synth as S  # noqa: E402
from manual_yolo_amd.arch import build_program  # noqa: E402
from oracle.yolo_ref import RefYolo  # noqa: E402


def main(task="detect", nc=64, scale="m", seed=0):
    torch.set_num_threads(8)
    x = S.synth_frames(2, 640, 640, seed=1)
    x = torch.from_numpy(x).permute(0, 3, 1, 2).float() / 255
    prog = build_program(task, nc, scale, nc_quirk=False)
    names = [r.prefix for r in prog.weights if r.kind in ("conv", "stem") and r.fused_bn]
    gains = {}
    S._GAINS_OVERRIDE = gains
    for n in names:
        sd = S.synth_state_dict(task, nc, scale, seed, calibrate=True)
        ref = RefYolo(sd, task, nc, scale, 1e-3, fuse=False, nc_quirk=False)
        got = {}

        class Stop(Exception):
            pass

        def hook(prefix, y, n=n, got=got):
            if prefix == n:
                got["std"] = float(y.std())
                raise Stop()
        ref.stats_hook = hook
        try:
            ref.forward(x)
        except Stop:
            pass
        gains[n] = 1.0 / max(got["std"], 1e-6)
        print(n, "raw std %.4f -> gain %.4f" % (got["std"], gains[n]), flush=True)
    # head: scale the final 1x1 convs so logits have std ~1.5 (cls) / ~1.0 (box)
    sd = S.synth_state_dict(task, nc, scale, seed, calibrate=True)
    ref = RefYolo(sd, task, nc, scale, 1e-3, nc_quirk=False)
    (y, raws) = ref.forward(x)
    for l, r in enumerate(raws):
        b, c = r[:, :64], r[:, 64:]
        bb = sd[f"model.22.cv2.{l}.2.bias"]; cb = sd[f"model.22.cv3.{l}.2.bias"]
        bstd = float((b - bb.view(1, -1, 1, 1)).std()); cstd = float((c - cb.view(1, -1, 1, 1)).std())
        gains[f"model.22.cv2.{l}.2"] = 1.0 / bstd
        gains[f"model.22.cv3.{l}.2"] = 1.5 / cstd
        print("level", l, "box std", bstd, "cls std", cstd)
    out = {"task": task, "nc": nc, "scale": scale, "seed": seed, "gains": gains}
    with open(os.path.join(ROOT, "manual_yolo_amd", "synth_gains.json"), "w") as f:
        json.dump(out, f, indent=0)
    S._GAINS_OVERRIDE = None
    S._load_gains.cache_clear()
    sd = S.synth_state_dict(task, nc, scale, seed)
    ref = RefYolo(sd, task, nc, scale, 1e-3, nc_quirk=False)
    (y, raws) = ref.forward(x)
    mx = y[:, 4:].amax(1)
    for t in (0.1, 0.25, 0.35, 0.5, 0.9):
        print("anchors with max score >", t, (mx > t).sum(1).tolist())


if __name__ == "__main__":
    main()
