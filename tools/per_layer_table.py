#!/usr/bin/env python3
"""Per-layer table from bench.py --profile-out: time, TFLOP/s, GB/s and the fraction of each layer's own roofline
bound  t_min = max(flops / dense MFMA peak, compulsory bytes / HBM peak)  (SURVEY.md 8d).
usage: python tools/per_layer_table.py perop.json dtype(f16|f32) out.md"""
import json
import sys

PEAK = {"f16": 2500e12, "f32": 157.3e12, "f8": 5000e12}
HBM = 8e12
src, dt, out = sys.argv[1], sys.argv[2], sys.argv[3]
ops = json.load(open(src))["by_op"]
kinds = {0: "stem", 1: "conv", 2: "maxpool5", 3: "decode", 4: "cls_head"}
lines = [f"# Per-layer times and roofline fractions ({dt}, yolov8m, one MI355X, the workload of the json; from `bench.py --profile-out`)", "",
         "| layer | op | k | cin->cout | /stride | kernel cfg | us | TFLOP/s | GB/s (compulsory) | bound | frac of own roofline |", "|" + "---|" * 11]
tot_ms = tot_min = 0.0
for o in ops:
    ms = o["avg_ms"]
    tf, tb = o["flop"] / PEAK[dt], o["bytes"] / HBM
    tmin = max(tf, tb)
    tot_ms += ms
    tot_min += tmin * 1e3
    lines.append(f"| {o['name']} | {kinds.get(o['kind'], o['kind'])} | {o['k']} | {o['cin']}->{o['cout']} | {o['down']} | {o['cfg']} | "
                 f"{ms * 1e3:.1f} | {o['tflops']:.0f} | {o['gbs']:.0f} | {'mfma' if tf >= tb else 'hbm'} | {tmin * 1e3 / ms:.3f} |")
lines += ["", f"Sum of kernel times {tot_ms:.3f} ms per step; sum of per-layer t_min {tot_min:.3f} ms; layer-wise mixed roofline fraction "
          f"{tot_min / tot_ms:.3f}.  cfg = impl*1000 + ksize*100 + WC*10 + TC (3 = conv_dmap, 7 = conv_t2d, 8 = conv_h2 with WC = 4 waves, 9 = conv_bneck: a fused Bottleneck, its row carries both convs' flops; 94xx / 95xx = conv_stem2: the stem and model.1 (95xx: and model.2.cv1) in one launch, the row carries all their flops and the frame + the last layer's output as bytes)."]
open(out, "w").write("\n".join(lines) + "\n")
print(lines[-1])
