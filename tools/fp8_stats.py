import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manual_yolo_amd.engine import engine_from_weights
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
from oracle.post_ref import non_max_suppression
from oracle.yolo_ref import RefYolo
sd, meta = synth_state_dict("detect", 64, "m", 0), synth_meta("detect", 64, "m")
calib = torch.from_numpy(np.concatenate([synth_frames(4, 640, 640, seed=101), synth_frames(2, 640, 640, seed=102, kind="blocks")]))
e8 = engine_from_weights(sd, meta, "f8", 0, bgr_input=False, calib_frames=calib)
e16 = engine_from_weights(sd, meta, "f16", 0, bgr_input=False)
frames = synth_frames(4, 640, 640, seed=1)
ref = RefYolo(sd, "detect", 64, "m", 1e-3, nc_quirk=False)
(y, raws) = ref.forward(torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255)
y = y.numpy()
x = torch.from_numpy(frames).cuda()
for name, e in (("f8", e8), ("f16", e16)):
    gy = e.head_raw(x).cpu().numpy()
    dec = [op for op in e.prog.ops if op.kind == 3][0]
    for lvl, v in enumerate(dec.src):
        got = e.read_buffer(v.buf, 4, 640, 640).cpu().numpy()
        want = raws[lvl].permute(0, 2, 3, 1).numpy()
        for nm, sl in (("box", slice(0, 64)), ("cls", slice(64, 128))):
            g, w = got[..., sl].ravel(), want[..., sl].ravel()
            sl_ = np.polyfit(w, g, 1)
            qs = [0.5, 0.99, 0.999, 0.9999]
            print(f"   slope {sl_[0]:.4f} icpt {sl_[1]:.4f} std got {g.std():.3f} | quantiles ref {np.quantile(w, qs).round(3)} got {np.quantile(g, qs).round(3)}")
            print(f"{name} level {lvl} {nm}: rel rms {np.linalg.norm(g - w) / np.linalg.norm(w - w.mean()):.3f} corr {np.corrcoef(g, w)[0, 1]:.4f} std {w.std():.3f}")
    outs, idxs = non_max_suppression(y, 0.25, 0.7)
    d, c, a = e.detect(x)
    for b in range(4):
        n = int(c[b]); ga = a[b, :n].cpu().numpy()
        cm = np.intersect1d(ga, idxs[b])
        # scores of the oracle at fp8-kept anchors, and fp8 scores at oracle-kept anchors
        so = y[b, 4:, :].max(0)
        sg = gy[b, 4:, :].max(0)
        print(f"{name} img {b}: kept {n} / oracle {len(idxs[b])} common {len(cm)} | oracle score at engine-kept anchors: min {so[ga].min() if n else 0:.3f} median {np.median(so[ga]) if n else 0:.3f} | engine score at oracle-kept: median {np.median(sg[idxs[b]]):.3f} min {sg[idxs[b]].min():.3f}")
