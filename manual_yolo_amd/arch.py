"""YOLOv8 model spec -> flat layer program for the HIP engine.

The reference builds its network by unpickling Ultralytics modules
(``YOLO(path)``, reference ``detect.py:20-21``); those modules were created by
[3P] ``ultralytics.nn.tasks.parse_model`` from the yaml spec embedded in the checkpoint.
This module walks the same spec (``[from, repeats, module, args]`` rows, width/depth
scaling, the ``c2 != nc`` rule) and emits what ``include/miyolo.h`` executes:

* activation buffers (NHWC, per image ``H/down x W/down x channels``),
* a list of ops (STEM / CONV / MAXPOOL5 / DECODE / CLS_HEAD) over channel-slice *views*,
* one weight recipe per conv (which state-dict tensors to fold and how to lay them out).

``chunk``, ``cat`` and ``nn.Upsample`` never become ops: a C2f writes its branches into
channel slices of one buffer, a Concat is a list of views, an Upsample is a flag on a view
that the consuming 1x1 conv folds into its load index.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

OP_STEM, OP_CONV, OP_MAXPOOL5, OP_DECODE, OP_CLS_HEAD = 0, 1, 2, 3, 4
REG_MAX = 16
CLASSIFY_HIDDEN = 1280  # [3P] Classify: c_ = 1280

# [3P] ultralytics/cfg/models/v8/yolov8.yaml and yolov8-cls.yaml (the cls one is also
# embedded in rank_classifier.pt); used when a weights bundle carries no spec of its own.
DEFAULT_SPECS = {
    "detect": {
        "scales": {"n": [0.33, 0.25, 1024], "s": [0.33, 0.50, 1024], "m": [0.67, 0.75, 768],
                   "l": [1.00, 1.00, 512], "x": [1.00, 1.25, 512]},
        "backbone": [[-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C2f", [128, True]],
                     [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C2f", [256, True]], [-1, 1, "Conv", [512, 3, 2]],
                     [-1, 6, "C2f", [512, True]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C2f", [1024, True]],
                     [-1, 1, "SPPF", [1024, 5]]],
        "head": [[-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 6], 1, "Concat", [1]], [-1, 3, "C2f", [512]],
                 [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 4], 1, "Concat", [1]], [-1, 3, "C2f", [256]],
                 [-1, 1, "Conv", [256, 3, 2]], [[-1, 12], 1, "Concat", [1]], [-1, 3, "C2f", [512]],
                 [-1, 1, "Conv", [512, 3, 2]], [[-1, 9], 1, "Concat", [1]], [-1, 3, "C2f", [1024]],
                 [[15, 18, 21], 1, "Detect", ["nc"]]],
    },
    "classify": {
        "scales": {"n": [0.33, 0.25, 1024], "s": [0.33, 0.50, 1024], "m": [0.67, 0.75, 1024],
                   "l": [1.00, 1.00, 1024], "x": [1.00, 1.25, 1024]},
        "backbone": [[-1, 1, "Conv", [64, 3, 2]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C2f", [128, True]],
                     [-1, 1, "Conv", [256, 3, 2]], [-1, 6, "C2f", [256, True]], [-1, 1, "Conv", [512, 3, 2]],
                     [-1, 6, "C2f", [512, True]], [-1, 1, "Conv", [1024, 3, 2]], [-1, 3, "C2f", [1024, True]]],
        "head": [[-1, 1, "Classify", ["nc"]]],
    },
}


@dataclass
class View:
    buf: int
    ch_off: int
    ch_cnt: int
    upsample: int = 0


@dataclass
class Value:
    """What a spec layer produces: a channel-concat of views at one resolution."""
    views: List[View]
    down: int  # H_in / H of this value as its consumer sees it

    @property
    def channels(self) -> int:
        return sum(v.ch_cnt for v in self.views)


@dataclass
class WeightRecipe:
    """How to produce one device tensor from the state dict (see weights.py)."""
    kind: str                       # 'conv' | 'bias' | 'stem' | 'linear' | 'linear_bias'
    prefix: str                     # state-dict prefix, e.g. 'model.2.m.0.cv1'
    fused_bn: bool = True           # Conv (conv+bn) vs bare nn.Conv2d (weight+bias)
    seg_channels: Tuple[int, ...] = ()  # input-channel split (views) for K padding


@dataclass
class Op:
    kind: int
    ksize: int = 0
    stride: int = 0
    act: int = 0
    cin: int = 0
    cout: int = 0
    src: List[View] = field(default_factory=list)
    dst: Optional[View] = None
    res: Optional[View] = None
    weight: int = -1
    bias: int = -1
    level_stride: Tuple[int, int, int] = (0, 0, 0)
    name: str = ""
    down_in: int = 1    # resolution of the op's input grid (H_in / H)
    down_out: int = 1


@dataclass
class Program:
    task: str
    nc: int
    bufs: List[Tuple[int, int, int]]   # (channels, down, dtype: -1 act / 0 f32)
    ops: List[Op]
    weights: List[WeightRecipe]
    max_stride: int
    layer_out: Dict[int, Value]        # spec layer index -> value (debug / parity taps)
    strides: Tuple[int, ...] = ()

    def conv_macs_per_image(self, h: int, w: int) -> int:
        tot = 0
        for op in self.ops:
            if op.kind in (OP_CONV, OP_STEM):
                tot += (h // op.down_out) * (w // op.down_out) * op.cout * op.cin * op.ksize * op.ksize
        return tot


def make_divisible(x: float, divisor: int) -> int:
    return int(math.ceil(x / divisor) * divisor)


class _Builder:
    def __init__(self, task: str, nc: int):
        self.p = Program(task, nc, [(3, 1, -2)], [], [], 1, {})  # buf 0 = uint8 input

    def buf(self, channels: int, down: int, dtype: int = -1) -> int:
        self.p.bufs.append((channels, down, dtype))
        return len(self.p.bufs) - 1

    def weight(self, r: WeightRecipe) -> int:
        self.p.weights.append(r)
        return len(self.p.weights) - 1

    def conv(self, x: Value, prefix: str, cout: int, k: int, s: int, act: int = 1,
             dst: Optional[View] = None, res: Optional[View] = None, fused_bn: bool = True,
             out_dtype: int = -1) -> Value:
        cin = x.channels
        down_out = x.down * s
        if dst is None:
            dst = View(self.buf(cout, down_out, out_dtype), 0, cout)
        if k != 1 and (len(x.views) != 1 or x.views[0].upsample):
            raise NotImplementedError(f"{prefix}: 3x3 conv over a concat/upsampled input")
        if len(x.views) > 2:
            raise NotImplementedError(f"{prefix}: conv over more than two concatenated inputs")
        segs = tuple(v.ch_cnt for v in x.views)
        if x.views[0].buf == 0:  # reads the uint8 image
            assert k == 3 and s == 2 and cin == 3, "stem must be conv3x3 s2 on 3 channels"
            wi = self.weight(WeightRecipe("stem", prefix, fused_bn, segs))
            kind = OP_STEM
        else:
            wi = self.weight(WeightRecipe("conv", prefix, fused_bn, segs))
            kind = OP_CONV
        bi = self.weight(WeightRecipe("bias", prefix, fused_bn, segs))
        self.p.ops.append(Op(kind, k, s, act, cin, cout, list(x.views), dst, res, wi, bi,
                             name=prefix, down_in=x.down, down_out=down_out))
        return Value([dst], down_out)

    # [3P] ultralytics.nn.modules.block.C2f / Bottleneck
    def c2f(self, x: Value, prefix: str, c2: int, n: int, shortcut: bool) -> Value:
        c = int(c2 * 0.5)
        ybuf = self.buf((2 + n) * c, x.down)
        self.conv(x, prefix + ".cv1", 2 * c, 1, 1, dst=View(ybuf, 0, 2 * c))
        for j in range(n):
            src = View(ybuf, (1 + j) * c, c)
            t = self.conv(Value([src], x.down), f"{prefix}.m.{j}.cv1", c, 3, 1)
            self.conv(t, f"{prefix}.m.{j}.cv2", c, 3, 1, dst=View(ybuf, (2 + j) * c, c),
                      res=src if shortcut else None)
        return self.conv(Value([View(ybuf, 0, (2 + n) * c)], x.down), prefix + ".cv2", c2, 1, 1)

    # [3P] ultralytics.nn.modules.block.SPPF
    def sppf(self, x: Value, prefix: str, c2: int, k: int) -> Value:
        assert k == 5, "SPPF pool size other than 5 not implemented"
        c_ = x.channels // 2
        ybuf = self.buf(4 * c_, x.down)
        self.conv(x, prefix + ".cv1", c_, 1, 1, dst=View(ybuf, 0, c_))
        for j in range(3):
            self.p.ops.append(Op(OP_MAXPOOL5, 5, 1, 0, c_, c_, [View(ybuf, j * c_, c_)],
                                 View(ybuf, (j + 1) * c_, c_), name=f"{prefix}.m{j}",
                                 down_in=x.down, down_out=x.down))
        return self.conv(Value([View(ybuf, 0, 4 * c_)], x.down), prefix + ".cv2", c2, 1, 1)

    # [3P] ultralytics.nn.modules.head.Detect (legacy=True: plain Conv cls branch)
    def detect(self, xs: Sequence[Value], prefix: str, nc: int, fuse_first: bool = False):
        """``fuse_first``: cv2[l][0] and cv3[l][0] are both conv3x3 + BN + SiLU over the same level input; run them as ONE
        conv of c2 + c3 output channels (one launch, the input read once, 4x the output tiles on P5 where 64 channels
        alone cannot fill the chip) whose output the two second convs read as channel slices.  Same arithmetic per
        output channel, so results are unchanged bit for bit."""
        ch = [x.channels for x in xs]
        c2 = max(16, ch[0] // 4, REG_MAX * 4)
        c3 = max(ch[0], min(nc, 100))
        raws, strides = [], []
        for l, x in enumerate(xs):
            raw = self.buf(4 * REG_MAX + nc, x.down, 0)   # fp32 [.., 64 box | nc cls]
            if fuse_first:
                f = self.conv(x, f"{prefix}.cv2.{l}.0|{prefix}.cv3.{l}.0", c2 + c3, 3, 1)
                fb = f.views[0].buf
                b = Value([View(fb, 0, c2)], x.down)
            else:
                b = self.conv(x, f"{prefix}.cv2.{l}.0", c2, 3, 1)
            b = self.conv(b, f"{prefix}.cv2.{l}.1", c2, 3, 1)
            self.conv(b, f"{prefix}.cv2.{l}.2", 4 * REG_MAX, 1, 1, act=0, fused_bn=False,
                      dst=View(raw, 0, 4 * REG_MAX))
            c = Value([View(fb, c2, c3)], x.down) if fuse_first else self.conv(x, f"{prefix}.cv3.{l}.0", c3, 3, 1)
            c = self.conv(c, f"{prefix}.cv3.{l}.1", c3, 3, 1)
            self.conv(c, f"{prefix}.cv3.{l}.2", nc, 1, 1, act=0, fused_bn=False,
                      dst=View(raw, 4 * REG_MAX, nc))
            raws.append(View(raw, 0, 4 * REG_MAX + nc))
            strides.append(x.down)
        assert len(xs) == 3, "Detect with other than 3 levels not implemented"
        self.p.ops.append(Op(OP_DECODE, src=raws, level_stride=tuple(strides), name=prefix,
                             cin=4 * REG_MAX + nc, cout=4 + nc))
        self.p.strides = tuple(strides)

    # [3P] ultralytics.nn.modules.head.Classify
    def classify(self, x: Value, prefix: str, nc: int):
        h = self.conv(x, prefix + ".conv", CLASSIFY_HIDDEN, 1, 1)
        wi = self.weight(WeightRecipe("linear", prefix + ".linear", False))
        bi = self.weight(WeightRecipe("linear_bias", prefix + ".linear", False))
        self.p.ops.append(Op(OP_CLS_HEAD, cin=CLASSIFY_HIDDEN, cout=nc, src=list(h.views), weight=wi,
                             bias=bi, name=prefix, down_in=h.down, down_out=h.down))


def build_program(task: str, nc: int, scale: str, spec: Optional[dict] = None,
                  nc_quirk: bool = True, fuse_head: bool = False) -> Program:
    """[3P] parse_model restated: see module docstring.  ``nc_quirk``: upstream only scales a
    layer's width ``if c2 != nc``; a yolov8m trained by Ultralytics with nc=64 (reference
    ``roadmap1.v3i.yolov8/data.yaml:5``) therefore has a 64-wide stem.  ``False`` gives the
    nominal widths BASELINE.md counts its FLOPs for."""
    spec = spec if spec and "backbone" in spec else DEFAULT_SPECS[task]
    depth, width, max_ch = spec["scales"][scale]
    b = _Builder(task, nc)
    values: List[Value] = []
    x = Value([View(0, 0, 3)], 1)
    for i, (f, n, m, args) in enumerate(spec["backbone"] + spec["head"]):
        args = [nc if a == "nc" else a for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        if isinstance(f, int):
            xin = x if f == -1 else values[f]
        else:
            xin = [x if j == -1 else values[j] for j in f]
        prefix = f"model.{i}"
        if m in ("Conv", "C2f", "SPPF"):
            c2 = args[0]
            if c2 != nc or not nc_quirk:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            if m == "Conv":
                x = b.conv(xin, prefix, c2, args[1], args[2] if len(args) > 2 else 1)
            elif m == "C2f":
                x = b.c2f(xin, prefix, c2, n, bool(args[1]) if len(args) > 1 else False)
            else:
                x = b.sppf(xin, prefix, c2, args[1] if len(args) > 1 else 5)
        elif m == "nn.Upsample":
            assert args[1] == 2 and args[2] == "nearest", "only nearest x2 upsample is implemented"
            assert all(v.upsample == 0 for v in xin.views), "stacked upsamples not implemented"
            x = Value([View(v.buf, v.ch_off, v.ch_cnt, 1) for v in xin.views], xin.down // 2)
        elif m == "Concat":
            assert len({v.down for v in xin}) == 1, "Concat of different resolutions"
            x = Value([vw for v in xin for vw in v.views], xin[0].down)
        elif m == "Detect":
            b.detect(xin, prefix, nc, fuse_head)
            x = None
        elif m == "Classify":
            b.classify(xin, prefix, nc)
            x = None
        else:
            raise NotImplementedError(f"module {m} is not part of the YOLOv8 detect/classify path")
        values.append(x)
        if x is not None:
            b.p.layer_out[i] = x
    b.p.max_stride = max(op.down_out for op in b.p.ops) if b.p.ops else 1
    return b.p
