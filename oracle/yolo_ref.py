"""CPU restatement (plain torch fp32 ops) of the Ultralytics YOLOv8 forward pass.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) - never imported by the
product package.

Each function cites what it follows.  ``[3P]`` marks third-party code that is not
under /root/reference (ultralytics==8.3.176, reference ``requirements.txt:95``);
those are restated from the published algorithm and anchored on the reference's
call sites (``detect.py:121,541``; ``pipe.py:179``; ``yolo.py:361``) and on the
module tree pickled inside ``rank_classifier.pt``.

The model graph is driven by the same yaml-style spec Ultralytics embeds in its
checkpoints (``ckpt['model'].yaml``): ``[from, repeats, module, args]`` rows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- specs
# [3P] ultralytics/cfg/models/v8/yolov8-cls.yaml - identical to the copy embedded in
# rank_classifier.pt (ckpt['model'].yaml, printed in SURVEY.md section 0).
YOLOV8_CLS_SPEC = {
    "scales": {"n": [0.33, 0.25, 1024], "s": [0.33, 0.50, 1024], "m": [0.67, 0.75, 1024],
               "l": [1.00, 1.00, 1024], "x": [1.00, 1.25, 1024]},
    "backbone": [
        [-1, 1, "Conv", [64, 3, 2]],
        [-1, 1, "Conv", [128, 3, 2]],
        [-1, 3, "C2f", [128, True]],
        [-1, 1, "Conv", [256, 3, 2]],
        [-1, 6, "C2f", [256, True]],
        [-1, 1, "Conv", [512, 3, 2]],
        [-1, 6, "C2f", [512, True]],
        [-1, 1, "Conv", [1024, 3, 2]],
        [-1, 3, "C2f", [1024, True]],
    ],
    "head": [[-1, 1, "Classify", ["nc"]]],
}

# [3P] ultralytics/cfg/models/v8/yolov8.yaml (detect).  Layer table cross-checked in
# SURVEY.md section 8a against the published 25.9 M params / 78.9 GFLOPs of yolov8m.
YOLOV8_DET_SPEC = {
    "scales": {"n": [0.33, 0.25, 1024], "s": [0.33, 0.50, 1024], "m": [0.67, 0.75, 768],
               "l": [1.00, 1.00, 512], "x": [1.00, 1.25, 512]},
    "backbone": [
        [-1, 1, "Conv", [64, 3, 2]],
        [-1, 1, "Conv", [128, 3, 2]],
        [-1, 3, "C2f", [128, True]],
        [-1, 1, "Conv", [256, 3, 2]],
        [-1, 6, "C2f", [256, True]],
        [-1, 1, "Conv", [512, 3, 2]],
        [-1, 6, "C2f", [512, True]],
        [-1, 1, "Conv", [1024, 3, 2]],
        [-1, 3, "C2f", [1024, True]],
        [-1, 1, "SPPF", [1024, 5]],
    ],
    "head": [
        [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
        [[-1, 6], 1, "Concat", [1]],
        [-1, 3, "C2f", [512]],
        [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
        [[-1, 4], 1, "Concat", [1]],
        [-1, 3, "C2f", [256]],
        [-1, 1, "Conv", [256, 3, 2]],
        [[-1, 12], 1, "Concat", [1]],
        [-1, 3, "C2f", [512]],
        [-1, 1, "Conv", [512, 3, 2]],
        [[-1, 9], 1, "Concat", [1]],
        [-1, 3, "C2f", [1024]],
        [[15, 18, 21], 1, "Detect", ["nc"]],
    ],
}

REG_MAX = 16  # [3P] Detect.reg_max


def make_divisible(x: float, divisor: int) -> int:
    """[3P] ultralytics.utils.ops.make_divisible."""
    return int(math.ceil(x / divisor) * divisor)


def parse_spec(spec: dict, nc: int, scale: str, ch: int = 3, nc_quirk: bool = True) -> List[dict]:
    """[3P] ultralytics.nn.tasks.parse_model, reduced to the modules YOLOv8 uses.

    Returns one dict per layer: ``{"i", "f", "type", "args", "c2"}`` with args as the
    constructor would receive them (C2f: [c1, c2, n, shortcut]).

    ``nc_quirk``: upstream only width-scales a layer ``if c2 != nc`` (the test is meant
    for Classify outputs), so a spec channel count that happens to equal ``nc`` is left
    unscaled - with nc=64 (reference ``roadmap1.v3i.yolov8/data.yaml:5``) a checkpoint
    trained by Ultralytics gets a 64-wide stem instead of yolov8m's 48.  ``True``
    reproduces that; ``False`` gives the nominal yolov8{n,s,m,l,x} widths that
    SURVEY.md/BASELINE.md count (78.88 GFLOP/frame for m).
    """
    depth, width, max_channels = spec["scales"][scale]
    chs = [ch]
    layers = []
    for i, (f, n, m, args) in enumerate(spec["backbone"] + spec["head"]):
        args = [nc if a == "nc" else a for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        if m in ("Conv", "C2f", "SPPF"):
            c1, c2 = chs[f], args[0]
            if c2 != nc or not nc_quirk:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m == "C2f":
                args.insert(2, n)
                n = 1
        elif m == "Classify":
            c1, c2 = chs[f], args[0]
            args = [c1, c2]
        elif m == "nn.Upsample":
            c2 = chs[f]
        elif m == "Concat":
            c2 = sum(chs[x] for x in f)
        elif m == "Detect":
            args = [args[0], [chs[x] for x in f]]
            c2 = None
        else:
            raise ValueError(f"unsupported module {m}")
        assert n == 1, "repeats > 1 only occur on C2f in YOLOv8 specs"
        layers.append({"i": i, "f": f, "type": m, "args": args, "c2": c2})
        if i == 0:
            chs = []
        chs.append(c2)
    return layers


# --------------------------------------------------------------------------- modules
def fuse_conv_bn(w: torch.Tensor, bn_w, bn_b, bn_mean, bn_var, eps: float):
    """[3P] ultralytics.utils.torch_utils.fuse_conv_and_bn (what AutoBackend(fuse=True)
    applies before the reference's ``model(frame)`` runs): W' = diag(g/sqrt(var+eps)) W,
    b' = beta - g*mean/sqrt(var+eps).  Same op order as upstream (torch.mm with a diag)."""
    cout = w.shape[0]
    w_bn = torch.diag(bn_w.div(torch.sqrt(eps + bn_var)))
    fw = torch.mm(w_bn, w.reshape(cout, -1)).reshape(w.shape)
    b_conv = torch.zeros(cout, dtype=w.dtype)
    b_bn = bn_b - bn_w.mul(bn_mean).div(torch.sqrt(bn_var + eps))
    fb = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
    return fw, fb


class RefYolo:
    """Functional YOLOv8 (detect or classify) over an Ultralytics-named state dict.

    ``sd`` keys follow ``model.<i>.<...>`` exactly as ``ckpt['model'].state_dict()``.
    All tensors are converted to fp32 (reference: ``model.float()`` in AutoBackend [3P]).
    """

    def __init__(self, sd: Dict[str, torch.Tensor], task: str, nc: int, scale: str,
                 bn_eps: float, fuse: bool = True, nc_quirk: bool = True):
        self.task = task
        self.nc = nc
        self.eps = bn_eps
        self.fuse = fuse
        spec = YOLOV8_DET_SPEC if task == "detect" else YOLOV8_CLS_SPEC
        self.layers = parse_spec(spec, nc, scale, nc_quirk=nc_quirk)
        self.sd = {k: v.detach().to(torch.float32) for k, v in sd.items()
                   if v.is_floating_point()}
        self.stride = [8, 16, 32]
        self._fused: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.stats_hook = None  # optional callable(prefix, raw_conv_output) (fuse=False only)
        self.save = sorted({x % 1000 for l in self.layers
                            for x in ([l["f"]] if isinstance(l["f"], int) else l["f"]) if x != -1})

    # ---- ultralytics.nn.modules.conv.Conv [3P]: conv(bias=False, pad=k//2) -> BN -> SiLU
    def _conv(self, x, prefix: str, k: int, s: int, act: bool = True):
        w = self.sd[prefix + ".conv.weight"]
        if self.fuse:
            if prefix not in self._fused:
                self._fused[prefix] = fuse_conv_bn(
                    w, self.sd[prefix + ".bn.weight"], self.sd[prefix + ".bn.bias"],
                    self.sd[prefix + ".bn.running_mean"], self.sd[prefix + ".bn.running_var"], self.eps)
            fw, fb = self._fused[prefix]
            y = F.conv2d(x, fw, fb, stride=s, padding=k // 2)
        else:
            y = F.conv2d(x, w, None, stride=s, padding=k // 2)
            if self.stats_hook is not None:
                self.stats_hook(prefix, y)
            y = F.batch_norm(y, self.sd[prefix + ".bn.running_mean"], self.sd[prefix + ".bn.running_var"],
                             self.sd[prefix + ".bn.weight"], self.sd[prefix + ".bn.bias"], False, 0.0, self.eps)
        return F.silu(y) if act else y

    # ---- ultralytics.nn.modules.block.Bottleneck [3P] (k=((3,3),(3,3)), e=1.0 inside C2f)
    def _bottleneck(self, x, prefix: str, add: bool):
        y = self._conv(self._conv(x, prefix + ".cv1", 3, 1), prefix + ".cv2", 3, 1)
        return x + y if add else y

    # ---- ultralytics.nn.modules.block.C2f [3P]
    def _c2f(self, x, prefix: str, n: int, shortcut: bool):
        y = list(self._conv(x, prefix + ".cv1", 1, 1).chunk(2, 1))
        for j in range(n):
            y.append(self._bottleneck(y[-1], f"{prefix}.m.{j}", shortcut))
        return self._conv(torch.cat(y, 1), prefix + ".cv2", 1, 1)

    # ---- ultralytics.nn.modules.block.SPPF [3P]
    def _sppf(self, x, prefix: str, k: int):
        y = [self._conv(x, prefix + ".cv1", 1, 1)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], k, 1, k // 2))
        return self._conv(torch.cat(y, 1), prefix + ".cv2", 1, 1)

    # ---- ultralytics.nn.modules.head.Classify [3P]: returns (softmax, logits) in eval
    def _classify(self, x, prefix: str):
        x = self._conv(x, prefix + ".conv", 1, 1)
        x = F.adaptive_avg_pool2d(x, 1).flatten(1)
        x = F.linear(x, self.sd[prefix + ".linear.weight"], self.sd[prefix + ".linear.bias"])
        return x.softmax(1), x

    # ---- ultralytics.nn.modules.head.Detect [3P] (legacy=True branch used by v8 specs)
    def _detect(self, xs: List[torch.Tensor], prefix: str):
        outs = []
        for l, x in enumerate(xs):
            b = self._conv(self._conv(x, f"{prefix}.cv2.{l}.0", 3, 1), f"{prefix}.cv2.{l}.1", 3, 1)
            b = F.conv2d(b, self.sd[f"{prefix}.cv2.{l}.2.weight"], self.sd[f"{prefix}.cv2.{l}.2.bias"])
            c = self._conv(self._conv(x, f"{prefix}.cv3.{l}.0", 3, 1), f"{prefix}.cv3.{l}.1", 3, 1)
            c = F.conv2d(c, self.sd[f"{prefix}.cv3.{l}.2.weight"], self.sd[f"{prefix}.cv3.{l}.2.bias"])
            outs.append(torch.cat((b, c), 1))
        return self._detect_inference(outs, prefix), outs

    def _detect_inference(self, x: List[torch.Tensor], prefix: str):
        """[3P] Detect._inference + make_anchors + DFL + dist2bbox(xywh=True)."""
        no = self.nc + 4 * REG_MAX
        bsz = x[0].shape[0]
        x_cat = torch.cat([xi.view(bsz, no, -1) for xi in x], 2)
        anchors, strides = make_anchors(x, self.stride, 0.5)
        anchors, strides = anchors.transpose(0, 1), strides.transpose(0, 1)
        box, cls = x_cat.split((4 * REG_MAX, self.nc), 1)
        # DFL: softmax over the 16 bins, 1x1 conv with weight arange(16)
        a = box.shape[2]
        dflw = self.sd.get(prefix + ".dfl.conv.weight",
                           torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1))
        dist = F.conv2d(box.view(bsz, 4, REG_MAX, a).transpose(2, 1).softmax(1), dflw).view(bsz, 4, a)
        lt, rb = dist.chunk(2, 1)
        x1y1 = anchors.unsqueeze(0) - lt
        x2y2 = anchors.unsqueeze(0) + rb
        c_xy = (x1y1 + x2y2) / 2
        wh = x2y2 - x1y1
        dbox = torch.cat((c_xy, wh), 1) * strides
        return torch.cat((dbox, cls.sigmoid()), 1)

    # ---- BaseModel._predict_once [3P]
    @torch.no_grad()
    def forward(self, x: torch.Tensor, return_feats: bool = False):
        ys: List[torch.Tensor] = []
        feats = {}
        for l in self.layers:
            f, t, a, i = l["f"], l["type"], l["args"], l["i"]
            if f != -1:
                x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
            p = f"model.{i}"
            if t == "Conv":
                x = self._conv(x, p, a[2], a[3])
            elif t == "C2f":
                x = self._c2f(x, p, a[2], a[3] if len(a) > 3 else False)
            elif t == "SPPF":
                x = self._sppf(x, p, a[2])
            elif t == "nn.Upsample":
                x = F.interpolate(x, scale_factor=a[1], mode=a[2])
            elif t == "Concat":
                x = torch.cat(x, 1)
            elif t == "Classify":
                x = self._classify(x, p)
            elif t == "Detect":
                x = self._detect(x, p)
            ys.append(x if i in self.save else None)
            if return_feats:
                feats[i] = x
        return (x, feats) if return_feats else x


def make_anchors(feats: Sequence[torch.Tensor], strides: Sequence[int], offset: float = 0.5):
    """[3P] ultralytics.utils.tal.make_anchors."""
    pts, st = [], []
    for f, s in zip(feats, strides):
        h, w = f.shape[2:]
        sx = torch.arange(w, dtype=torch.float32) + offset
        sy = torch.arange(h, dtype=torch.float32) + offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
    return torch.cat(pts), torch.cat(st)


def count_params(sd: Dict[str, torch.Tensor]) -> int:
    return sum(v.numel() for k, v in sd.items()
               if v.is_floating_point() and not k.endswith(("running_mean", "running_var")))
