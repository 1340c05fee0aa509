"""CPU fake-quant restatement of the fp8 (e4m3) execution of the YOLOv8 forward pass - BASELINE.json config 5.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) - never imported by the product package.

The reference runs fp32 (``runs/rank_classifier/args.yaml:40`` ``half: false``; call sites ``detect.py:541``,
``pipe.py:41-42,179`` for the 1280x1280 geometry, ``detect.py:121`` for the classifier).  fp8 is the MI355X-side
precision option north_star asks for; nothing in the reference pins it, so this file is "parity unpinned" as far as
ACCURACY goes.  What it pins is the ARITHMETIC of the HIP fp8 engine at model level: the same network walked module by
module as ``yolo_ref.RefYolo`` does ([3P] ultralytics Conv / C2f / Bottleneck / SPPF / Detect / Classify), with every
stored tensor held as OCP e4m3 values ``q`` plus one scale per channel ``s`` (real value = q * s), exactly the scheme
``manual_yolo_amd/quant.py`` documents and ``csrc/common.h`` (``epilogue_fast``) executes:

* conv input: the stored bytes ``q_x`` of the producing ops (a concat = concat of (q, s) pairs; nearest upsample and
  max-pool act on q, max commutes with a positive scale);
* weights: BN folded in fp32 (``W * g/sqrt(var+eps)``, the product's op order so the e4m3 bytes agree), input scales
  folded per input channel ``W_eff = W * s_in[c]``, one scale per output channel ``qs[n] = max|W_eff[n]| / 448``,
  ``q_w = e4m3(W_eff / qs)``;
* ``acc = sum q_w * q_x`` in fp32 (the MFMA accumulates fp32; only the summation ORDER differs from the GPU);
* ``v = acc * (qs[n] * gain) + bias[n]``; SiLU in fp32; ``+ q_res * s_res`` for a Bottleneck shortcut;
* stored as ``e4m3(clamp(v * (1 / s_out), +-448))``, round-to-nearest-even; Detect's last 1x1 convs and the classifier's
  Linear stay fp32;
* stem: f16 arithmetic on the uint8 frame (``(u * (1/255))`` rounded to half, f16 weights, fp32 accumulate), e4m3 store.

Scales and gains are INPUTS (dicts keyed by the conv's state-dict prefix), normally the HIP engine's own calibration,
so that a test compares two executions of the same quantised network.  ``calibrate()`` derives scales from an fp32 pass
the way the engine's calibration does (``amax * 1.25 / 448`` of every stored tensor, residual included) for CPU-only
studies (tools/fp8_cpu_study.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .yolo_ref import REG_MAX, RefYolo

FP8_MAX = 448.0
HEADROOM = 1.25


def e4m3(x: torch.Tensor) -> torch.Tensor:
    """Round to OCP e4m3fn (saturating at +-448, round-to-nearest-even, subnormals kept), returned as fp32 values."""
    return x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32)


class QT:
    """A stored activation: e4m3 values ``q`` [B,C,H,W] (as fp32) and the per-channel scale ``s`` [C]."""
    __slots__ = ("q", "s")

    def __init__(self, q: torch.Tensor, s: torch.Tensor):
        self.q, self.s = q, s

    def real(self) -> torch.Tensor:
        return self.q * self.s.view(1, -1, 1, 1)

    def chunk2(self):
        c = self.q.shape[1] // 2
        return QT(self.q[:, :c], self.s[:c]), QT(self.q[:, c:], self.s[c:])


def qcat(xs: List[QT]) -> QT:
    return QT(torch.cat([x.q for x in xs], 1), torch.cat([x.s for x in xs]))


class RefYoloQuant(RefYolo):
    """``mode="quant"``: the fake-quant walk described in the module docstring.  ``mode="calib"``: the same walk in plain
    fp32 (q holds real values, s = 1) that records ``amax[prefix]`` of every tensor the quantised walk would store.
    ``weight_only`` / ``act_only`` switch one half of the quantisation off (studies only)."""

    def __init__(self, sd, task: str, nc: int, scale: str, bn_eps: float, out_scale: Optional[Dict[str, float]] = None,
                 gains: Optional[Dict[str, float]] = None, nc_quirk: bool = True, mode: str = "quant",
                 quant_weights: bool = True, quant_acts: bool = True, in_mean: Optional[Dict[str, torch.Tensor]] = None):
        super().__init__(sd, task, nc, scale, bn_eps, fuse=True, nc_quirk=nc_quirk)
        self.out_scale = dict(out_scale or {})
        self.gains = dict(gains or {})
        self.mode = mode
        self.quant_weights, self.quant_acts = quant_weights, quant_acts
        self.amax: Dict[str, float] = {}
        # bias correction (quant.py): in_mean[prefix] = mean real input of the conv per input channel over the calibration
        # pixels; the expected output shift of the weight rounding, sum_k dW[n,k] * E[x_k], is taken out of the bias
        self.in_mean = dict(in_mean or {})
        self._msum: Dict[str, torch.Tensor] = {}
        self._mcnt: Dict[str, float] = {}
        # acc_noise > 0: every accumulator is multiplied by (1 + acc_noise * N(0,1)) - a stand-in for ANOTHER fp32 summation
        # order (the GPU's MFMA tree vs this CPU conv differ by ~1e-6 relative).  A value within that distance of a rounding
        # boundary then lands on the other e4m3 code, a 6-12 % change of that element, which moves ~1000 downstream sums by
        # enough to flip ~5 % of THEIR roundings: the flips avalanche, and two executions of the same quantised network
        # that differ only in summation order end up as far apart as the quantisation noise itself (tests/test_gpu_fp8.py
        # uses this as the yardstick for the end-to-end comparison; single layers on identical input bytes agree code for code).
        self.acc_noise = 0.0
        self.taps: Dict[str, torch.Tensor] = {}      # prefix -> real-valued stored output (filled when keep_taps)
        self.keep_taps = False
        self._qw: Dict[tuple, tuple] = {}

    # ---- product-order BN fold (manual_yolo_amd/weights.py fold_conv: W * scale.view(-1,1,1,1)); RefYolo's torch.mm
    # form can differ in the last bit, which would flip e4m3 rounding ties of single weights
    def _fold(self, prefix: str, fused_bn: bool = True):
        if fused_bn:
            w = self.sd[prefix + ".conv.weight"]
            g, beta = self.sd[prefix + ".bn.weight"], self.sd[prefix + ".bn.bias"]
            mean, var = self.sd[prefix + ".bn.running_mean"], self.sd[prefix + ".bn.running_var"]
            sc = g.div(torch.sqrt(self.eps + var))
            return w * sc.view(-1, 1, 1, 1), beta - g.mul(mean).div(torch.sqrt(var + self.eps))
        return self.sd[prefix + ".weight"], self.sd[prefix + ".bias"]

    def _noise_gen(self):
        if not hasattr(self, "_ng"):
            self._ng = torch.Generator().manual_seed(1234)
        return self._ng

    def _store(self, prefix: str, v: torch.Tensor) -> QT:
        c = v.shape[1]
        if self.mode == "calib":
            self.amax[prefix] = max(self.amax.get(prefix, 0.0), float(v.abs().max()))
            if self.keep_taps:
                self.taps[prefix] = v
            return QT(v, torch.ones(c))
        so = float(self.out_scale[prefix])
        inv = float(torch.tensor(1.0 / so, dtype=torch.float32))      # the host passes 1/s as a float
        q = e4m3(v * inv) if self.quant_acts else v * inv
        out = QT(q, torch.full((c,), so, dtype=torch.float32))
        if self.keep_taps:
            self.taps[prefix] = out.real()
        return out

    def _qconv(self, x: QT, prefix: str, k: int, s: int, act: bool = True, res: Optional[QT] = None,
               fused_bn: bool = True, out_f32: bool = False):
        fw, fb = self._fold(prefix, fused_bn)
        if self.mode == "calib":
            v = F.conv2d(x.q, fw, fb, stride=s, padding=k // 2)
            self._msum[prefix] = self._msum.get(prefix, 0) + x.q.double().sum((0, 2, 3))
            self._mcnt[prefix] = self._mcnt.get(prefix, 0.0) + float(x.q.shape[0] * x.q.shape[2] * x.q.shape[3])
        else:
            key = (prefix, tuple(x.s.tolist()))
            if key not in self._qw:
                cout, cin = fw.shape[0], fw.shape[1]
                weff = fw * x.s.view(1, cin, 1, 1)
                flat = weff.permute(0, 2, 3, 1).reshape(cout, -1)                 # (ky, kx, cin): the product's K order
                am = flat.abs().amax(1)
                qs = torch.where(am > 0, am / FP8_MAX, torch.ones_like(am))
                qf = flat / qs.view(-1, 1)
                qf = e4m3(qf) if self.quant_weights else qf
                qw4 = qf.view(cout, k, k, cin).permute(0, 3, 1, 2).contiguous()
                bc = torch.zeros(cout)
                if prefix in self.in_mean:
                    dw = qw4 * qs.view(-1, 1, 1, 1) / x.s.view(1, cin, 1, 1) - fw          # real-weight error of the rounding
                    bc = (dw.sum((2, 3)) * self.in_mean[prefix].view(1, cin)).sum(1)
                self._qw[key] = (qw4, qs, bc)
            qw, qs, bc = self._qw[key]
            acc = F.conv2d(x.q, qw, None, stride=s, padding=k // 2)
            if self.acc_noise:
                acc = acc * (1.0 + self.acc_noise * torch.randn(acc.shape, generator=self._noise_gen()))
            g = float(self.gains.get(prefix, 1.0))
            v = acc * (qs * g).view(1, -1, 1, 1) + (fb - bc).view(1, -1, 1, 1)
        if act:
            v = F.silu(v)
        if res is not None:
            v = v + res.real()
        if out_f32:
            return v
        return self._store(prefix, v)

    # ---- stem: uint8 frame, f16 arithmetic (kernels_misc.h stem_kernel: (float)u * (1/255) rounded to half)
    def _stem(self, u8: torch.Tensor, prefix: str, k: int, s: int) -> QT:
        fw, fb = self._fold(prefix)
        if self.mode == "calib":
            v = F.conv2d(u8.float() / 255, fw, fb, stride=s, padding=k // 2)
        else:
            x = (u8.float() * (1.0 / 255.0)).to(torch.float16).float()
            v = F.conv2d(x, fw.to(torch.float16).float(), fb, stride=s, padding=k // 2)
        return self._store(prefix, F.silu(v))

    def _qbottleneck(self, x: QT, prefix: str, add: bool) -> QT:
        t = self._qconv(x, prefix + ".cv1", 3, 1)
        return self._qconv(t, prefix + ".cv2", 3, 1, res=x if add else None)

    def _qc2f(self, x: QT, prefix: str, n: int, shortcut: bool) -> QT:
        y = list(self._qconv(x, prefix + ".cv1", 1, 1).chunk2())
        for j in range(n):
            y.append(self._qbottleneck(y[-1], f"{prefix}.m.{j}", shortcut))
        return self._qconv(qcat(y), prefix + ".cv2", 1, 1)

    def _qsppf(self, x: QT, prefix: str, k: int) -> QT:
        y = [self._qconv(x, prefix + ".cv1", 1, 1)]
        for _ in range(3):
            y.append(QT(F.max_pool2d(y[-1].q, k, 1, k // 2), y[-1].s))
        return self._qconv(qcat(y), prefix + ".cv2", 1, 1)

    def _qdetect(self, xs: List[QT], prefix: str):
        outs = []
        for l, x in enumerate(xs):
            b = self._qconv(self._qconv(x, f"{prefix}.cv2.{l}.0", 3, 1), f"{prefix}.cv2.{l}.1", 3, 1)
            b = self._qconv(b, f"{prefix}.cv2.{l}.2", 1, 1, act=False, fused_bn=False, out_f32=True)
            c = self._qconv(self._qconv(x, f"{prefix}.cv3.{l}.0", 3, 1), f"{prefix}.cv3.{l}.1", 3, 1)
            c = self._qconv(c, f"{prefix}.cv3.{l}.2", 1, 1, act=False, fused_bn=False, out_f32=True)
            outs.append(torch.cat((b, c), 1))
        return self._detect_inference(outs, prefix), outs

    def _qclassify(self, x: QT, prefix: str):
        h = self._qconv(x, prefix + ".conv", 1, 1)
        p = F.adaptive_avg_pool2d(h.real(), 1).flatten(1)
        lg = F.linear(p, self.sd[prefix + ".linear.weight"], self.sd[prefix + ".linear.bias"])
        return lg.softmax(1), lg

    @torch.no_grad()
    def forward_u8(self, u8_nchw: torch.Tensor):
        """uint8 [B,3,H,W] (RGB) -> what ``RefYolo.forward`` returns ((y, raws) for detect, (probs, logits) for classify)."""
        ys: List = []
        x = None
        for l in self.layers:
            f, t, a, i = l["f"], l["type"], l["args"], l["i"]
            if i > 0 and f != -1:
                x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
            p = f"model.{i}"
            if t == "Conv":
                x = self._stem(u8_nchw, p, a[2], a[3]) if i == 0 else self._qconv(x, p, a[2], a[3])
            elif t == "C2f":
                x = self._qc2f(x, p, a[2], a[3] if len(a) > 3 else False)
            elif t == "SPPF":
                x = self._qsppf(x, p, a[2])
            elif t == "nn.Upsample":
                x = QT(F.interpolate(x.q, scale_factor=a[1], mode=a[2]), x.s)
            elif t == "Concat":
                x = qcat(x)
            elif t == "Classify":
                x = self._qclassify(x, p)
            elif t == "Detect":
                x = self._qdetect(x, p)
            ys.append(x if i in self.save else None)
        return x


def calibrate(sd, task: str, nc: int, scale: str, bn_eps: float, frames_u8_nchw: torch.Tensor, nc_quirk: bool = True,
              headroom: float = HEADROOM, batch: int = 4) -> Dict[str, float]:
    """{conv prefix: activation scale} from an fp32 walk over the calibration frames: ``amax * headroom / 448`` of every
    stored tensor (quant.py spec_from_amax)."""
    m = RefYoloQuant(sd, task, nc, scale, bn_eps, nc_quirk=nc_quirk, mode="calib")
    for b0 in range(0, frames_u8_nchw.shape[0], batch):
        m.forward_u8(frames_u8_nchw[b0:b0 + batch])
    calibrate.in_mean = {k: (v / m._mcnt[k]).float() for k, v in m._msum.items()}      # side result: bias-correction means
    return {k: max(v, 1e-6) * headroom / FP8_MAX for k, v in m.amax.items()}
