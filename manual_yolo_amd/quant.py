"""Static fp8 (e4m3, OCP "fn") quantisation of the detect path - BASELINE.json config 5.

The reference runs fp32 on the CPU (``runs/rank_classifier/args.yaml:40`` ``half: false``); fp8 is the MI355X-side
precision option north_star asks for ("yolov8m detect fp8 weights (CDNA4 fp8 MFMA), 1280x1280").  Scheme:

* **Activations**: every activation buffer slice holds e4m3 values ``q = x / s`` with ONE scale ``s`` per producing op
  (per-tensor scale), ``s = headroom * amax / 448`` where ``amax`` is the largest |activation| the op produced on the
  calibration frames (run once through the f16 engine).  SPPF max-pool outputs inherit their source slice's scale
  (max commutes with a positive scale).
* **Weights**: e4m3 with one scale per OUTPUT channel.  The input scales are folded in first, per input channel -
  ``W_eff[n, k] = W[n, k] * s_in[k]`` - so a conv over a concat of slices with different scales (C2f.cv2, the FPN 1x1s)
  needs no per-segment handling in the kernel: ``x[n] = acc[n] * qscale[n] + bias[n]``.
* **Arithmetic**: ``v_mfma_scale_f32_16x16x128_f8f6f4`` with unit block scales, fp32 accumulate; bias, SiLU and the
  residual add in fp32; the head's last 1x1 convs write fp32 raw maps, so DFL/sigmoid/NMS are unchanged.
* The stem (uint8 -> 48 channels, K = 27) stays an f16 MFMA and stores e4m3.
* **Bias correction** (round 3; replaces round 2's fitted per-op gains): rounding a weight row to e4m3 leaves an error
  ``dW[n, k]`` that is fixed, so its product with the MEAN input is a constant shift of output channel n,
  ``eps[n] = sum_k dW[n, k] * E[x_k]``.  SiLU outputs have a large positive mean, so this shift is the dominant part of the
  weight-rounding error on a trained net (the reference's rank classifier, CPU fake-quant walk, tools/fp8_cpu_study.py:
  last-layer relative error 0.54 -> 0.32, top-1 61/67 -> 63/67 = the fp32 result).  ``E[x_k]`` comes from the same
  calibration pass as the ranges; ``eps`` is subtracted from the bias on the host - no kernel change.
* What round 2 called "rounding shrinks the signal" is NOT a property of round-to-nearest (which is unbiased: slope 1.000 per
  layer with exact inputs).  The round-2 synthetic yolov8m amplified any amplitude change of its input ~60x by construction
  (unit-variance LSUV gains put every SiLU at its most super-linear point: a 1 % smaller stem output gave 47 % smaller head
  activations in exact fp32, tools/fp8_cpu_study.py --amplitude); sub-percent second-order effects of the rounding noise
  then arrive at the head as a 20-50 % shrink.  The trained classifier's factor is 0.3, the re-conditioned synthetic
  detector's (synth.py, pre-activation std 4) is 2.  ``gain_correction`` is kept for reference, off by default.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from .arch import OP_CONV, OP_MAXPOOL5, OP_STEM, Program

FP8_MAX = 448.0
HEADROOM = 1.25          # activations up to 1.25 x the calibration maximum are representable; beyond that they saturate


@dataclass
class QuantSpec:
    buf_scale: Dict[int, np.ndarray]                  # activation buffer -> per-channel scale (float32 [channels])
    out_scale: Dict[int, float] = field(default_factory=dict)   # op index -> scale of the slice it writes
    buf_mean: Dict[int, np.ndarray] = field(default_factory=dict)   # activation buffer -> per-channel mean activation (bias correction)


def save_spec(spec: "QuantSpec", path: str):
    """QuantSpec -> .npz (so that a profiled run can skip the calibration pass: bench.py --quant-cache)."""
    d = {f"bs{k}": v for k, v in spec.buf_scale.items()}
    d.update({f"bm{k}": v for k, v in spec.buf_mean.items()})
    d["os_idx"] = np.asarray(sorted(spec.out_scale), np.int64)
    d["os_val"] = np.asarray([spec.out_scale[k] for k in sorted(spec.out_scale)], np.float64)
    np.savez(path, **d)


def load_spec(path: str) -> "QuantSpec":
    z = np.load(path)
    bs = {int(k[2:]): z[k] for k in z.files if k.startswith("bs")}
    bm = {int(k[2:]): z[k] for k in z.files if k.startswith("bm")}
    return QuantSpec(bs, {int(i): float(v) for i, v in zip(z["os_idx"], z["os_val"])}, bm)


def spec_from_amax(prog: Program, amax: Dict[int, float], headroom: float = HEADROOM) -> QuantSpec:
    """amax: op index -> max |activation| written by that op (ops writing activation-dtype buffers)."""
    buf_scale = {i: np.ones(c, np.float32) for i, (c, d, dt) in enumerate(prog.bufs) if dt == -1}
    out_scale: Dict[int, float] = {}
    for i, op in enumerate(prog.ops):
        if op.kind in (OP_CONV, OP_STEM) and prog.bufs[op.dst.buf][2] == -1:
            s = max(float(amax.get(i, 1.0)), 1e-6) * headroom / FP8_MAX
            out_scale[i] = s
            buf_scale[op.dst.buf][op.dst.ch_off:op.dst.ch_off + op.dst.ch_cnt] = s
        elif op.kind == OP_MAXPOOL5:
            src, dst = op.src[0], op.dst
            buf_scale[dst.buf][dst.ch_off:dst.ch_off + dst.ch_cnt] = buf_scale[src.buf][src.ch_off:src.ch_off + src.ch_cnt]
    return QuantSpec(buf_scale, out_scale)


def calibrate(prog: Program, sd, bn_eps: float, frames: torch.Tensor, device: Optional[int] = None, bgr_input: bool = True,
              headroom: float = HEADROOM, eng16=None, bias_correction: bool = True) -> QuantSpec:
    """Run the calibration frames (uint8 [N,H,W,3], on the GPU or not) through the f16 engine and read every op's output
    range from the activation buffers (miyolo_read_buffer)."""
    from .engine import Engine
    eng = eng16 if eng16 is not None else Engine(prog, sd, bn_eps, "f16", device, bgr_input)
    restore = _materialise_all(eng)
    frames = frames.to(eng.device)
    N, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
    amax: Dict[int, float] = {}
    msum: Dict[int, torch.Tensor] = {}
    mcnt = 0
    step = max(1, min(N, eng.chunk(N, H, W), 4))
    for b0 in range(0, N, step):
        x = frames[b0:b0 + step].contiguous()
        eng.head_raw(x) if prog.task == "detect" else eng.classify(x)
        cache: Dict[int, torch.Tensor] = {}
        seen = set()
        for i, op in enumerate(prog.ops):
            if op.kind not in (OP_CONV, OP_STEM, OP_MAXPOOL5) or prog.bufs[op.dst.buf][2] != -1:
                continue
            if op.dst.buf not in cache:
                cache = {op.dst.buf: eng.read_buffer(op.dst.buf, x.shape[0], H, W)}      # one buffer at a time (memory)
            if op.dst.buf not in seen:           # a C2f's buffer comes round once per branch: count its means once per batch
                seen.add(op.dst.buf)
                msum[op.dst.buf] = msum.get(op.dst.buf, 0) + cache[op.dst.buf].double().mean((1, 2)).sum(0)
            if op.kind == OP_MAXPOOL5:
                continue
            v = cache[op.dst.buf][..., op.dst.ch_off:op.dst.ch_off + op.dst.ch_cnt]
            amax[i] = max(amax.get(i, 0.0), float(v.abs().max()))
        mcnt += x.shape[0]
    restore()
    spec = spec_from_amax(prog, amax, headroom)
    if bias_correction:
        spec.buf_mean = {b: (v / mcnt).float().cpu().numpy() for b, v in msum.items()}
    return spec


def _materialise_all(eng):
    """The f16 engine's fused launches (a narrow Bottleneck as one launch, stem + layer 1 as one, the one-launch classifier)
    keep the first op's output in LDS: its buffer in the workspace is never written (round-2 ADVICE: calibration read uninitialised memory for
    yolov8m's model.2.m.*.cv1 - garbage scales for two layers).  Switch those fusions off while taps are read; returns a
    function that restores the options.  (miyolo_read_buffer now also refuses such a buffer.)"""
    prev = {k: eng.options.get(k, d) for k, d in (("bneck_fuse", 1), ("stem_fuse", 1), ("cls_mega", 1))}
    for k in prev:                       # cls_mega: the one-launch classifier keeps EVERY activation in LDS
        eng.set_option(k, 0)

    def restore():
        for k, v in prev.items():
            eng.set_option(k, v)
    return restore


def gain_correction(eng8, eng16, frames: torch.Tensor, lo: float = 0.9, hi: float = 1.2) -> Dict[int, float]:
    """Round-to-nearest quantisation leaves an error that is uncorrelated with the QUANTISED value, not with the exact one:
    every rounding stage lowers the variance of what it stores by the variance of its error (3.6 % rms per e4m3 tensor),
    and over ~80 layers the fp8 activations end up 10-20 % smaller than the f16 ones (measured: slope 0.79-0.89 of the
    head logits against the oracle's, tools/fp8_layer_error.py) - which costs most of the detections, whose scores sit on
    the far tail.  Standard post-training remedy, done once at build time: walk the ops in order and scale each conv's
    dequantisation factors by the least-squares gain g = <a16, a8> / <a8, a8> between its fp8 output and the f16
    engine's output on calibration frames (the bias is exact and is not scaled), then re-run the op so that the next
    one sees the corrected activations.  Returns {op index: g}."""
    frames = frames.to(eng8.device).contiguous()
    B, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
    assert B <= eng8.chunk(B, H, W) and B <= eng16.chunk(B, H, W)
    restore = _materialise_all(eng16)
    if eng16.prog.task == "detect":
        eng16.head_raw(frames)
    else:
        eng16.classify(frames)
    gains: Dict[int, float] = {}
    ops16 = eng16.prog.ops
    for i, op in enumerate(eng8.prog.ops):
        eng8.run_ops(i, i + 1, frames if i == 0 else None, B, H, W)
        if op.kind != OP_CONV or i not in eng8.f8_extra:
            continue
        d8, d16 = op.dst, ops16[i].dst
        a8 = eng8.read_buffer(d8.buf, B, H, W)[..., d8.ch_off:d8.ch_off + d8.ch_cnt]
        a16 = eng16.read_buffer(d16.buf, B, H, W)[..., d16.ch_off:d16.ch_off + d16.ch_cnt]
        if op.res is not None:                       # the correction applies to the conv branch: take the residual out
            r8 = eng8.read_buffer(op.res.buf, B, H, W)[..., op.res.ch_off:op.res.ch_off + op.res.ch_cnt]
            r16 = eng16.read_buffer(ops16[i].res.buf, B, H, W)[..., ops16[i].res.ch_off:ops16[i].res.ch_off + ops16[i].res.ch_cnt]
            a8, a16 = a8 - r8, a16 - r16
        den = float((a8 * a8).sum())
        g = float((a16 * a8).sum()) / den if den > 0 else 1.0
        g = min(max(g, lo), hi)
        gains[i] = g
        qi, bi = eng8.f8_extra[i]
        eng8.weights[qi].mul_(g)
        eng8.weights[bi].div_(g)
        eng8.run_ops(i, i + 1, None, B, H, W)
    torch.cuda.synchronize(eng8.device)
    restore()
    return gains


def quantize_conv_weight(wf: torch.Tensor, s_in: np.ndarray):
    """wf [cout,cin,kh,kw] fp32 (BN folded), s_in [cin] -> (uint8 e4m3 [cout,kpad] in (ky,kx,cin) order padded to 128,
    qscale [cout] fp32)."""
    cout, cin, kh, kw = wf.shape
    weff = wf * torch.from_numpy(s_in.astype(np.float32)).view(1, cin, 1, 1)
    flat = weff.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
    am = flat.abs().amax(1)
    qs = torch.where(am > 0, am / FP8_MAX, torch.ones_like(am))
    q = (flat / qs.view(-1, 1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    kpad = (flat.shape[1] + 127) // 128 * 128
    out = torch.zeros((cout, kpad), dtype=torch.uint8)
    out[:, :flat.shape[1]] = q.view(torch.uint8)
    return out.contiguous(), qs.float()


def weight_rounding_shift(wf: torch.Tensor, q_u8: torch.Tensor, qs: torch.Tensor, s_in: np.ndarray, m_in: np.ndarray) -> torch.Tensor:
    """eps[n] = sum_k (dequantised weight - exact weight)[n, k] * E[x_k]: the constant output shift the e4m3 rounding of
    row n causes on inputs of mean ``m_in`` (real units, per input channel; every tap of a 3x3 kernel sees the same mean -
    the zero padding at the border is ignored).  wf [cout,cin,kh,kw] exact folded weights, q_u8 / qs as returned by
    quantize_conv_weight (so in the stored domain the weight is W * s_in and the input x / s_in)."""
    cout, cin, kh, kw = wf.shape
    K = cin * kh * kw
    s = torch.from_numpy(s_in.astype(np.float32))
    wq_real = (dequant_fp8_bytes(q_u8)[:, :K] * qs.view(-1, 1)).view(cout, kh, kw, cin) / s.view(1, 1, 1, cin)
    dw = wq_real - wf.permute(0, 2, 3, 1)
    return (dw.sum((1, 2)) * torch.from_numpy(m_in.astype(np.float32)).view(1, cin)).sum(1)


def dequant_fp8_bytes(u8: torch.Tensor) -> torch.Tensor:
    return u8.view(torch.float8_e4m3fn).float()


def fp8_round(x: torch.Tensor) -> torch.Tensor:
    """Round-trip through e4m3 (saturating), for test references."""
    return x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()
