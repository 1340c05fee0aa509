"""Replay stress of the one-launch classifier against the layered path (bit-identity), for hunting rare LDS-DMA hazards.
usage (GPU box): python tools/stress_cls_mega.py [rounds] [--with-detect]
--with-detect: a yolov8m f16 detect engine runs batches of 16 frames 640 x 640 on ANOTHER stream the whole time, so the
classifier's LDS-DMA rings share the chip (L2, fabric, CUs) with conv kernels - the situation of manual_yolo_amd/chain.py
(detect.py:541 -> detect.py:121), which an idle-chip replay does not exercise."""
import sys

import torch

sys.path.insert(0, ".")
from manual_yolo_amd.ckpt import load_bundle  # noqa: E402
from manual_yolo_amd.engine import engine_from_weights  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
with_detect = "--with-detect" in sys.argv
rounds = int(argv[0]) if argv else 40
sd, meta = load_bundle("tests/golden/rank_best.safetensors")
eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
g = torch.Generator().manual_seed(3)
bad = tot = 0
det = None
if with_detect:
    from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
    dsd, dmeta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
    det = engine_from_weights(dsd, dmeta, "f16", 0, bgr_input=False)
    dframes = torch.from_numpy(synth_frames(16, 640, 640, seed=5)).cuda()
    dstream = torch.cuda.Stream()
    dout = None
    with torch.cuda.stream(dstream):
        dref = [t.clone() for t in det.detect(dframes)]
    dbad = dcalls = 0


def keep_detect_busy(n=3):
    """Enqueue n detect calls on the side stream (asynchronous: they run under the classifier calls that follow)."""
    global dbad, dcalls
    if det is None:
        return
    with torch.cuda.stream(dstream):
        for _ in range(n):
            out = det.detect(dframes)
            dcalls += 1
            dbad += int(not all(torch.equal(a, b) for a, b in zip(out, dref)))
for B in (67, 256, 1, 1024, 300):
    x = torch.randint(0, 256, (B, 64, 64, 3), dtype=torch.uint8, generator=g).cuda()
    eng.set_option("cls_mega", 0)
    ref = eng.classify(x)[0].clone()
    eng.set_option("cls_mega", 1)
    for r in range(rounds if B > 1 else rounds * 10):
        if r % 4 == 0:
            keep_detect_busy()
        out = eng.classify(x)[0]
        nb = int(((out - ref).abs().max(1).values > 0).sum())
        bad += nb; tot += B
        if nb:
            print("batch", B, "replay", r, "images differing", nb)
print("images checked", tot, "differing", bad)
if det is not None:
    torch.cuda.synchronize()
    print("detect calls beside them", dcalls, "differing", dbad)
    bad += dbad
sys.exit(1 if bad else 0)
