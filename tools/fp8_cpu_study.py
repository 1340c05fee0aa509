#!/usr/bin/env python3
"""CPU-only study of the static e4m3 scheme (oracle/quant_ref.py): where does the fp8 network's signal go?

For every stored tensor of the fake-quant walk, against the fp32 walk on the same frames:
  slope   least-squares slope of the fp8 tensor on the fp32 one  <a8,a16>/<a16,a16>   (1 = no shrink)
  invg    the engine's gain-correction estimate                   <a16,a8>/<a8,a8>    (what quant.gain_correction fits)
  rel     relative rms error
  sat     fraction of stored values at +-448 (saturated)
  ftz     fraction of stored values that are exactly 0 where the fp32 value is not
Switches: --weights-only / --acts-only quantise one side only; --bias-corr applies the bias correction of quant.py;
--amplitude runs NO quantisation at all: the exact-fp32 network twice, the second time with the stem's output scaled by
0.99, and prints how much smaller every later tensor comes out (the network's own amplification of an amplitude change -
what turns sub-percent effects of rounding noise into a visible "shrink" on an ill-conditioned random-init net);
--gains FILE uses another synth_gains.json (e.g. the round-2 unit-variance table, kept as tools/synth_gains_r2.json).

usage: python tools/fp8_cpu_study.py [detect-n|detect-m|classify] [imgsz] [--acts-only|--weights-only|--bias-corr|--amplitude] [--gains=FILE]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manual_yolo_amd.ckpt import load_bundle  # noqa: E402
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict  # noqa: E402
from oracle.quant_ref import RefYoloQuant, calibrate  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = {a for a in sys.argv[1:] if a.startswith("--")}
    what = args[0] if args else "detect-n"
    qa, qw = "--weights-only" not in flags, "--acts-only" not in flags
    torch.set_num_threads(8)
    for f in flags:
        if f.startswith("--gains="):
            import json
            import manual_yolo_amd.synth as S
            S._GAINS_OVERRIDE = json.load(open(f.split("=", 1)[1]))["gains"]
    if what.startswith("detect"):
        scale = what.split("-")[1]
        sz = int(args[1]) if len(args) > 1 else 320
        sd, meta = synth_state_dict("detect", 64, scale, 0), synth_meta("detect", 64, scale)
        task, nc, eps, quirk = "detect", 64, meta["bn_eps"], False
        cal = np.concatenate([synth_frames(4, sz, sz, seed=101), synth_frames(2, sz, sz, seed=102, kind="blocks")])
        test = synth_frames(2, sz, sz, seed=1)
    else:
        sd, meta = load_bundle(os.path.join(ROOT, "tests", "golden", "rank_best.safetensors"))
        val = np.load(os.path.join(ROOT, "tests", "golden", "rank_valid.npz"))
        task, nc, scale, eps, quirk = "classify", meta["nc"], meta["scale"], meta["bn_eps"], True
        cal, test = val["pre_u8"][:32], val["pre_u8"]
    tonchw = lambda a: torch.from_numpy(a).permute(0, 3, 1, 2).contiguous()  # noqa: E731
    if "--amplitude" in flags:
        from oracle.quant_ref import QT

        class Scaled(RefYoloQuant):
            k = 1.0

            def _stem(self, u8, prefix, ks, st):
                o = super()._stem(u8, prefix, ks, st)
                return QT(o.q * self.k, o.s)
        a = Scaled(sd, task, nc, scale, eps, nc_quirk=quirk, mode="calib"); a.keep_taps = True
        b = Scaled(sd, task, nc, scale, eps, nc_quirk=quirk, mode="calib"); b.keep_taps = True; b.k = 0.99
        a.forward_u8(tonchw(test[:4])); b.forward_u8(tonchw(test[:4]))
        print(f"{what}: exact fp32, stem output x 0.99 -> least-squares slope of every later tensor on the unscaled run")
        for i, k in enumerate(a.taps):
            x, y = a.taps[k].double().flatten(), b.taps[k].double().flatten()
            sl = float((x * y).sum() / (x * x).sum())
            print(f"{k:28s} slope {sl:.4f}   amplification of the 1 % change: {(1 - sl) / 0.01:6.1f}x")
        return
    scales = calibrate(sd, task, nc, scale, eps, tonchw(cal), nc_quirk=quirk)
    f32 = RefYoloQuant(sd, task, nc, scale, eps, nc_quirk=quirk, mode="calib"); f32.keep_taps = True
    q8 = RefYoloQuant(sd, task, nc, scale, eps, scales, nc_quirk=quirk, quant_weights=qw, quant_acts=qa,
                      in_mean=calibrate.in_mean if "--bias-corr" in flags else None); q8.keep_taps = True
    o32 = f32.forward_u8(tonchw(test))
    o8 = q8.forward_u8(tonchw(test))
    print(f"{what}: quantise weights={qw} activations={qa}")
    print(f"{'tensor':28s} {'slope':>7s} {'invg':>7s} {'rel':>7s} {'sat':>8s} {'ftz':>8s} {'amax32':>8s} {'448*s':>8s}")
    for k in f32.taps:
        a16, a8 = f32.taps[k].double().flatten(), q8.taps[k].double().flatten()
        slope = float((a8 * a16).sum() / (a16 * a16).sum())
        invg = float((a8 * a16).sum() / (a8 * a8).sum())
        rel = float((a8 - a16).norm() / a16.norm())
        s = scales[k]
        sat = float(((a8.abs() / s) >= 447.9).double().mean())
        ftz = float(((a8 == 0) & (a16 != 0)).double().mean())
        print(f"{k:28s} {slope:7.4f} {invg:7.4f} {rel:7.4f} {sat:8.5f} {ftz:8.5f} {float(a16.abs().max()):8.3f} {448 * s:8.3f}")
    if task == "detect":
        for lvl, (r32, r8) in enumerate(zip(o32[1], o8[1])):
            for nm, sl in (("box", slice(0, 64)), ("cls", slice(64, None))):
                w, g = r32[:, sl].double().flatten(), r8[:, sl].double().flatten()
                wc, gc = w - w.mean(), g - g.mean()
                print(f"head level {lvl} {nm}: slope {float((gc * wc).sum() / (wc * wc).sum()):.4f} corr {float((gc * wc).sum() / (gc.norm() * wc.norm())):.4f} "
                      f"rel rms {float((g - w).norm() / wc.norm()):.4f}")
    else:
        lab = val["labels"]
        p32, p8 = o32[1].argmax(1).numpy(), o8[1].argmax(1).numpy()
        print(f"top-1 fp32 {int((p32 == lab).sum())}/{len(lab)}  fake-quant fp8 {int((p8 == lab).sum())}/{len(lab)}  "
              f"same arg-max {int((p32 == p8).sum())}/{len(lab)}  max |dlogit| {float((o32[1] - o8[1]).abs().max()):.3f}")


if __name__ == "__main__":
    main()
