import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.engine import engine_from_weights
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
sd, meta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
eng = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
frames = torch.from_numpy(synth_frames(64, 640, 640, seed=1)).cuda()
y = eng.head_raw(frames)
d, c, a = eng.detect(frames)
print("kept per image: mean %.1f max %d" % (c.float().mean().item(), c.max().item()))
cand = (y[:, 4:].amax(1) > 0.25).sum(1)
print("candidates per image: mean %.1f max %d min %d" % (cand.float().mean().item(), cand.max().item(), cand.min().item()))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for md in (300, 50, 1):
    print("nms max_det", md, "%.3f ms" % t(lambda: eng.nms(y, 640, 640, 0.25, 0.7, False, md)))
print("nms conf 0.9 (few candidates) %.3f ms" % t(lambda: eng.nms(y, 640, 640, 0.9, 0.7, False, 300)))
cls = d[..., 5].long()
worst = 0
for b in range(8):
    h = torch.bincount(cls[b, : int(c[b])], minlength=64)
    worst = max(worst, int(h.max()))
    if b < 2:
        print("image", b, "kept", int(c[b]), "largest class chain", int(h.max()), "classes used", int((h > 0).sum()))
print("nms agnostic %.3f ms" % t(lambda: eng.nms(y, 640, 640, 0.25, 0.7, True, 300)))
