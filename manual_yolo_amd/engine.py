"""ctypes binding of ``include/miyolo.h`` (the gfx950 HIP library) + device plumbing.

This is the only place the product touches the native library.  PyTorch provides device
memory (weights, workspace, outputs) and the stream; every compute step runs inside
``libmiyolo.so``.  There is NO fallback: if the library is missing or no MI355X is visible,
constructing an :class:`Engine` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .arch import OP_DECODE, Op, Program, View, build_program
from .weights import K_ALIGN, build_weight_tensors

# MIYOLO_LIB: an alternative build of the same ABI (A/B timing of kernel variants on one box); still no CPU fallback
_LIB_PATH = os.environ.get("MIYOLO_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmiyolo.so")
DT_CODE = {"f32": 0, "f16": 1, "f8": 2}


class MiyoloError(RuntimeError):
    pass


class _View(C.Structure):
    _fields_ = [("buf", C.c_int32), ("ch_off", C.c_int32), ("ch_cnt", C.c_int32), ("upsample", C.c_int32)]


class _Buf(C.Structure):
    _fields_ = [("channels", C.c_int32), ("down", C.c_int32), ("dtype", C.c_int32), ("reserved", C.c_int32)]


class _Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("n_src", C.c_int32),
                ("src", _View * 3), ("dst", _View), ("res", _View),
                ("weight", C.c_int32), ("bias", C.c_int32),
                ("level_stride", C.c_int32 * 3), ("qscale", C.c_int32), ("bias_init", C.c_int32),
                ("out_inv_scale", C.c_float), ("res_scale", C.c_float), ("reserved", C.c_int32 * 1)]


class _Desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("task", C.c_int32), ("dtype", C.c_int32), ("nc", C.c_int32),
                ("reg_max", C.c_int32), ("max_stride", C.c_int32), ("n_bufs", C.c_int32),
                ("n_ops", C.c_int32), ("n_weights", C.c_int32), ("reserved", C.c_int32 * 7)]


# name -> (restype, argtypes): every symbol include/miyolo.h declares
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "miyolo_abi_version": (_i, []),
    "miyolo_k_align": (_i, [_i]),
    "miyolo_create": (_i, [C.POINTER(_Desc), C.POINTER(_Buf), C.POINTER(_Op), C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "miyolo_destroy": (None, [_vp]),
    "miyolo_last_error": (C.c_char_p, [_vp]),
    "miyolo_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "miyolo_chunk": (_i, [_vp, _i, _i, _i]),
    "miyolo_classify_launches": (_i, [_vp, _i, _i, C.POINTER(_sz)]),
    "miyolo_set_option": (_i, [_vp, C.c_char_p, _i]),
    "miyolo_wait_outputs": (_i, [_vp, _vp]),
    "miyolo_set_classes": (_i, [_vp, _vp, _i]),
    "miyolo_detect": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "miyolo_head_raw": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "miyolo_nms": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "miyolo_classify": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "miyolo_read_buffer": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "miyolo_write_buffer": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "miyolo_run_ops": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "miyolo_work": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "miyolo_op_work": (_i, [_vp, _i, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "miyolo_profile_read": (_i, [_vp, _i, _vp, _vp, _vp]),
    "miyolo_graph_info": (_i, [_vp, _vp]),
    "miyolo_debug_candidate_counts": (_i, [_vp, _vp, _i, _vp]),
    "miyolo_debug_stamps": (_i, [_vp, _vp]),
    "miyolo_letterbox": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "miyolo_crop_resize": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "miyolo_slice_batch": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "miyolo_merge_slices": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "miyolo_merge_slices_nmm": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen ``libmiyolo.so`` (built in-tree by ``__graft_entry__.build()`` / ``csrc/build.sh``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MiyoloError(f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.miyolo_abi_version() != 1:
        raise MiyoloError("libmiyolo.so ABI version mismatch")
    for k, v in K_ALIGN.items():
        if lib.miyolo_k_align(DT_CODE[k]) != v:
            raise MiyoloError("weights.K_ALIGN disagrees with miyolo_k_align()")
    _lib = lib
    return lib


def iou_threshold_f32(iou: float) -> float:
    """torchvision's CPU nms compares the fp32 IoU against a C++ double threshold; the largest
    float <= that double gives the identical decision in pure fp32 (oracle/post_ref.py)."""
    t = np.float32(iou)
    if float(t) > float(iou):
        t = np.nextafter(t, np.float32(-np.inf))
    return float(t)


def _view(v: Optional[View]) -> _View:
    if v is None:
        return _View(-1, 0, 0, 0)
    return _View(v.buf, v.ch_off, v.ch_cnt, v.upsample)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Engine:
    """One model on one GPU.  Not re-entrant (one call in flight), as the C ABI states."""

    def __init__(self, prog: Program, sd: Dict[str, torch.Tensor], bn_eps: float, dtype: str = "f16",
                 device: Optional[int] = None, bgr_input: bool = True, quant=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise MiyoloError("no GPU visible: the MI355X path has no CPU fallback")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.prog, self.dtype = prog, dtype
        self.nc = prog.nc
        self.quant = quant
        self.f8_extra = {}
        extra = {}
        if dtype == "f8":
            # the conv kernels switch K segments per 128-byte K step: the FIRST segment of a two-view 1x1 conv must be a
            # multiple of 128 channels.  Where only the second one is (yolov8m: cat(up(576), 384), cat(192, 384)), swap the
            # views - the weight columns are permuted to match in build_weight_tensors (op.swapped).
            import copy
            prog = copy.deepcopy(prog)
            for op in prog.ops:
                if op.kind == 1 and len(op.src) == 2 and op.src[0].ch_cnt % 128 and op.src[1].ch_cnt % 128 == 0:
                    op.src = [op.src[1], op.src[0]]
                    op.swapped = op.src[1].ch_cnt                 # channels of the ORIGINAL first segment
            self.prog = prog
            if quant is None:
                raise MiyoloError("dtype f8 needs a quant.QuantSpec (engine_from_weights calibrates one)")
            cpu_w, extra = build_weight_tensors(prog, sd, bn_eps, dtype, bgr_input, quant)
            self.f8_extra = extra
        else:
            cpu_w = build_weight_tensors(prog, sd, bn_eps, dtype, bgr_input)
        # the conv kernels fetch biases with scalar loads, 16 floats per channel tile, for every tile that starts below
        # cout rounded up to the widest tile - not range-checked by the hardware (round 2's intermittent memory access fault):
        # every bias array must carry the padding weights._pad128 gives it
        for op in prog.ops:
            if op.kind in (0, 1) and cpu_w[op.bias].numel() < (op.cout + 127) // 128 * 128 + 256:
                raise MiyoloError(f"{op.name}: bias array of {cpu_w[op.bias].numel()} floats lacks the padding the scalar loads need")
        self.weights = [w.to(self.device) for w in cpu_w]   # kept alive for the handle's lifetime
        bufs = (_Buf * len(prog.bufs))(*[_Buf(c, d, dt, 0) for (c, d, dt) in prog.bufs])
        ops = (_Op * len(prog.ops))()
        for i, op in enumerate(prog.ops):
            o = ops[i]
            o.kind, o.ksize, o.stride, o.act, o.cin, o.cout = op.kind, op.ksize, op.stride, op.act, op.cin, op.cout
            o.n_src = len(op.src)
            for j in range(3):
                o.src[j] = _view(op.src[j] if j < len(op.src) else None)
            o.dst, o.res = _view(op.dst), _view(op.res)
            o.weight, o.bias = op.weight, op.bias
            for j in range(3):
                o.level_stride[j] = op.level_stride[j]
            o.qscale, o.bias_init, o.out_inv_scale, o.res_scale = -1, -1, 1.0, 1.0
            if dtype == "f8":
                if i in extra:
                    o.qscale, o.bias_init = extra[i]
                if i in quant.out_scale:
                    o.out_inv_scale = 1.0 / quant.out_scale[i]
                if op.res is not None:
                    rs = quant.buf_scale[op.res.buf][op.res.ch_off:op.res.ch_off + op.res.ch_cnt]
                    if float(rs.max()) != float(rs.min()):
                        raise MiyoloError(f"{op.name}: residual slice spans several activation scales")
                    o.res_scale = float(rs[0])
        desc = _Desc(1, 0 if prog.task == "detect" else 1, DT_CODE[dtype], prog.nc, 16, prog.max_stride,
                     len(prog.bufs), len(prog.ops), len(self.weights))
        wptrs = (C.c_void_p * len(self.weights))(*[w.data_ptr() for w in self.weights])
        h = C.c_void_p()
        rc = self.lib.miyolo_create(C.byref(desc), bufs, ops, wptrs, self.device_index, C.byref(h))
        if rc != 0:
            raise MiyoloError(f"miyolo_create failed ({rc}): {self.lib.miyolo_last_error(None).decode()}")
        self.h = h
        self.options: Dict[str, int] = {}            # what set_option was called with (library defaults are not listed)
        self._ws: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ plumbing
    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.miyolo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise MiyoloError(f"{what} failed ({rc}): {self.lib.miyolo_last_error(self.h).decode()}")

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def set_option(self, key: str, value: int):
        self._check(self.lib.miyolo_set_option(self.h, key.encode(), int(value)), "miyolo_set_option")
        self.options[key] = int(value)
        if key == "graph":
            # hipGraph replay (library option "graph"): capture needs a non-default stream, and the replayed launches
            # write to the buffers of the captured call - so graph mode owns a side stream and per-shape output buffers
            self._graph = bool(value)
            self._gstream = torch.cuda.Stream(self.device) if value else None
            self._gout = {}

    def set_classes(self, classes=None):
        """`classes=` filter applied on the device BEFORE NMS (None / empty: no filter)."""
        cl = [int(c) for c in classes] if classes is not None else []
        arr = (C.c_int32 * max(len(cl), 1))(*cl)
        self._check(self.lib.miyolo_set_classes(self.h, arr if cl else None, len(cl)), "miyolo_set_classes")

    def _graph_call(self, fn):
        """Run fn() on the graph side stream, ordered after / before the caller's current stream."""
        cur = torch.cuda.current_stream(self.device)
        self._gstream.wait_stream(cur)
        with torch.cuda.stream(self._gstream):
            r = fn()
        cur.wait_stream(self._gstream)
        return r

    def workspace(self, B: int, H: int, W: int) -> torch.Tensor:
        need = self.lib.miyolo_workspace_bytes(self.h, B, H, W)
        if need == 0:
            raise MiyoloError(f"bad shape B={B} H={H} W={W}: {self.lib.miyolo_last_error(self.h).decode()}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def chunk(self, B: int, H: int, W: int) -> int:
        return self.lib.miyolo_chunk(self.h, B, H, W)

    def classify_launches(self, H: int, W: int) -> Tuple[int, int]:
        """(kernel launches per classify() call of HxW crops, LDS bytes per image of the one-launch kernel or 0)."""
        lds = _sz(0)
        n = self.lib.miyolo_classify_launches(self.h, H, W, C.byref(lds))
        if n < 0:
            self._check(n, "classify_launches")
        return n, int(lds.value)

    def _in(self, frames: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise MiyoloError("input must be uint8 [B,H,W,3]")
        if frames.device != self.device:
            frames = frames.to(self.device, non_blocking=True)
        frames = frames.contiguous()
        return frames, frames.shape[0], frames.shape[1], frames.shape[2]

    def num_anchors(self, H: int, W: int) -> int:
        return sum((H // s) * (W // s) for s in self.prog.strides)

    # ------------------------------------------------------------------ hot path
    def detect(self, frames: torch.Tensor, conf: float = 0.25, iou: float = 0.7, agnostic: bool = False,
               max_det: int = 300, scale: Optional[torch.Tensor] = None, want_anchor: bool = True,
               out: Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]] = None, defer: bool = False):
        """frames uint8 [B,H,W,3] on the GPU -> (dets [B,max_det,6], counts [B], anchor [B,max_det]).
        With option nms_async the outputs are complete in stream order only after `wait_outputs()`; `defer=False`
        (default) calls it here, so the call behaves as without the option; a pipelined caller passes `defer=True` and
        waits where it consumes."""
        x, B, H, W = self._in(frames)
        ws = self.workspace(B, H, W)
        if out is None:
            dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=self.device)
            counts = torch.empty((B,), dtype=torch.int32, device=self.device)
            anchor = torch.empty((B, max_det), dtype=torch.int32, device=self.device) if want_anchor else None
        else:
            dets, counts, anchor = out
        call = lambda: self._check(
            self.lib.miyolo_detect(self.h, x.data_ptr(), B, H, W, float(np.float32(conf)), iou_threshold_f32(iou), int(agnostic), max_det,
                                   _ptr(scale), dets.data_ptr(), counts.data_ptr(), _ptr(anchor), ws.data_ptr(), ws.numel(), self._stream()),
            "miyolo_detect")
        if getattr(self, "_graph", False) and out is not None:      # replay only makes sense with caller-owned, reused outputs
            self._graph_call(call)
        else:
            call()
        if not defer:
            self.wait_outputs()
        return dets, counts, anchor

    def wait_outputs(self, stream: Optional["torch.cuda.Stream"] = None):
        """Order `stream` (default: the current one) behind the most recent asynchronous NMS (option nms_async)."""
        st = (stream.cuda_stream if stream is not None else self._stream())
        self._check(self.lib.miyolo_wait_outputs(self.h, st), "miyolo_wait_outputs")

    def head_raw(self, frames: torch.Tensor) -> torch.Tensor:
        x, B, H, W = self._in(frames)
        ws = self.workspace(B, H, W)
        y = torch.empty((B, 4 + self.nc, self.num_anchors(H, W)), dtype=torch.float32, device=self.device)
        self._check(self.lib.miyolo_head_raw(self.h, x.data_ptr(), B, H, W, y.data_ptr(), ws.data_ptr(), ws.numel(),
                                             self._stream()), "miyolo_head_raw")
        return y

    def nms(self, y: torch.Tensor, H: int, W: int, conf: float = 0.25, iou: float = 0.7, agnostic: bool = False,
            max_det: int = 300, scale: Optional[torch.Tensor] = None):
        y = y.to(self.device, torch.float32).contiguous()
        B, _, A = y.shape
        ws = self.workspace(B, H, W)
        dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=self.device)
        counts = torch.empty((B,), dtype=torch.int32, device=self.device)
        anchor = torch.empty((B, max_det), dtype=torch.int32, device=self.device)
        rc = self.lib.miyolo_nms(self.h, y.data_ptr(), B, A, H, W, float(np.float32(conf)), iou_threshold_f32(iou),
                                 int(agnostic), max_det, _ptr(scale), dets.data_ptr(), counts.data_ptr(),
                                 anchor.data_ptr(), ws.data_ptr(), ws.numel(), self._stream())
        self._check(rc, "miyolo_nms")
        return dets, counts, anchor

    def detect_sliced(self, frame: torch.Tensor, boxes, slice_hw, conf: float = 0.25, iou: float = 0.7, agnostic: bool = False,
                      max_det: int = 300, extra: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                      merge: str = "GREEDYNMM", match_metric: str = "IOS", match_threshold: float = 0.5,
                      merge_agnostic: Optional[bool] = None):
        """One frame uint8 [H,W,3] (device) cut into slices `boxes` (list of x1,y1,x2,y2) of canvas slice_hw, ONE batched
        miyolo_detect, boxes shifted back and merged on the device: `merge="GREEDYNMM"` (sahi's default, what the reference's
        call reaches: miyolo_merge_slices_nmm, `match_metric` IOS / IOU, `match_threshold`) or `merge="NMS"` (the class-aware
        NMS of the per-frame post-process at `iou`, miyolo_merge_slices - limited to n_slices * max_det <= the anchors of one
        slice, 8400 at 640 x 640).  `extra`: optional (dets [1,max_det,6] in frame coordinates, counts [1]) of a full-frame
        pass, merged in as one more "slice" at (0,0) placed last, as sahi appends it.
        Returns (dets [max_det,6] frame coordinates, count, index [max_det] = slice * max_det + row of every kept box)."""
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise MiyoloError("frame must be uint8 [H,W,3]")
        frame = frame.to(self.device).contiguous()
        H, W = int(frame.shape[0]), int(frame.shape[1])
        sh, sw = slice_hw
        n = len(boxes)
        bx = torch.tensor(boxes, dtype=torch.int32, device=self.device).reshape(n, 4)
        batch = torch.empty((n, sh, sw, 3), dtype=torch.uint8, device=self.device)
        if self.lib.miyolo_slice_batch(frame.data_ptr(), H, W, bx.data_ptr(), n, batch.data_ptr(), sh, sw, 114, self._stream()):
            raise MiyoloError(f"miyolo_slice_batch failed: {self.lib.miyolo_last_error(None).decode()}")
        dets, counts, _ = self.detect(batch, conf, iou, agnostic, max_det, None, want_anchor=False)
        if extra is not None:
            dets = torch.cat([dets, extra[0].to(self.device).reshape(1, max_det, 6)])
            counts = torch.cat([counts, extra[1].to(self.device).reshape(1).to(torch.int32)])
            bx = torch.cat([bx, torch.zeros((1, 4), dtype=torch.int32, device=self.device)])
            n += 1
        if merge.upper() == "GREEDYNMM":
            od = torch.empty((max_det, 6), dtype=torch.float32, device=self.device)
            oc = torch.empty((2,), dtype=torch.int32, device=self.device)
            oi = torch.empty((max_det,), dtype=torch.int32, device=self.device)
            metric = {"IOS": 0, "IOU": 1}[match_metric.upper()]
            if self.lib.miyolo_merge_slices_nmm(dets.contiguous().data_ptr(), counts.contiguous().data_ptr(), bx.data_ptr(), n, max_det, H, W, metric,
                                                float(np.float32(match_threshold)), int(agnostic if merge_agnostic is None else merge_agnostic), max_det, od.data_ptr(), oc.data_ptr(),
                                                oi.data_ptr(), self._stream()):
                raise MiyoloError(f"miyolo_merge_slices_nmm failed: {self.lib.miyolo_last_error(None).decode()}")
            if int(oc[1]) < 0:
                raise MiyoloError(f"{-int(oc[1])} slice detections exceed the 4096 candidates the device merge holds: raise conf or lower max_det")
            return od, oc[0], oi
        if merge.upper() != "NMS":
            raise MiyoloError(f"unknown merge {merge!r} (GREEDYNMM or NMS)")
        ws = self.workspace(n if extra is None else n - 1, sh, sw)
        cap = n * max_det
        if cap > self.num_anchors(sh, sw):
            raise MiyoloError(f"{n} slices x max_det {max_det} exceed the {self.num_anchors(sh, sw)} candidate slots of a {sh}x{sw} workspace")
        ysc = torch.empty(((4 + self.nc) * cap,), dtype=torch.float32, device=self.device)
        od = torch.empty((1, max_det, 6), dtype=torch.float32, device=self.device)
        oc = torch.empty((1,), dtype=torch.int32, device=self.device)
        oi = torch.empty((1, max_det), dtype=torch.int32, device=self.device)
        self._check(self.lib.miyolo_merge_slices(self.h, dets.contiguous().data_ptr(), counts.contiguous().data_ptr(), bx.data_ptr(), n, max_det, sh, sw,
                                                 iou_threshold_f32(iou), int(agnostic), max_det, ysc.data_ptr(), od.data_ptr(), oc.data_ptr(),
                                                 oi.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), "miyolo_merge_slices")
        return od[0], oc[0], oi[0]

    def classify(self, frames: torch.Tensor):
        """uint8 [B,H,W,3] -> (logits, probs) [B,nc] f32.  In graph mode the returned tensors are the engine's own
        per-shape buffers: valid until the next call with the same shape."""
        x, B, H, W = self._in(frames)
        ws = self.workspace(B, H, W)
        if getattr(self, "_graph", False):
            if ("c", B, H, W) not in self._gout:
                self._gout[("c", B, H, W)] = (torch.empty((B, self.nc), dtype=torch.float32, device=self.device),
                                              torch.empty((B, self.nc), dtype=torch.float32, device=self.device))
            logits, probs = self._gout[("c", B, H, W)]
            self._graph_call(lambda: self._check(
                self.lib.miyolo_classify(self.h, x.data_ptr(), B, H, W, logits.data_ptr(), probs.data_ptr(), ws.data_ptr(), ws.numel(),
                                         self._stream()), "miyolo_classify"))
            return logits, probs
        logits = torch.empty((B, self.nc), dtype=torch.float32, device=self.device)
        probs = torch.empty((B, self.nc), dtype=torch.float32, device=self.device)
        self._check(self.lib.miyolo_classify(self.h, x.data_ptr(), B, H, W, logits.data_ptr(), probs.data_ptr(),
                                             ws.data_ptr(), ws.numel(), self._stream()), "miyolo_classify")
        return logits, probs

    # ------------------------------------------------------------------ parity taps
    def buffer_shape(self, buf: int, B: int, H: int, W: int):
        c, d, _ = self.prog.bufs[buf]
        return (B, H // d, W // d, c)

    def read_buffer(self, buf: int, B: int, H: int, W: int, raw: bool = False) -> torch.Tensor:
        """fp32 copy of an activation buffer [B, H/down, W/down, channels].  fp8 engines: the real-valued activations
        (stored e4m3 value x the buffer's scale), or with `raw` the stored e4m3 values themselves."""
        ws = self.workspace(B, H, W)
        out = torch.empty(self.buffer_shape(buf, B, H, W), dtype=torch.float32, device=self.device)
        self._check(self.lib.miyolo_read_buffer(self.h, buf, B, H, W, out.data_ptr(), ws.data_ptr(), self._stream()),
                    "miyolo_read_buffer")
        if self.dtype == "f8" and buf in self.quant.buf_scale and not raw:       # stored e4m3 values -> real activations
            out *= torch.from_numpy(self.quant.buf_scale[buf]).to(self.device)
        return out

    def write_buffer(self, buf: int, x: torch.Tensor, H: int, W: int):
        x = x.to(self.device, torch.float32).contiguous()
        if self.dtype == "f8" and buf in self.quant.buf_scale:
            x = (x / torch.from_numpy(self.quant.buf_scale[buf]).to(self.device)).contiguous()
        B = x.shape[0]
        assert tuple(x.shape) == self.buffer_shape(buf, B, H, W), (tuple(x.shape), self.buffer_shape(buf, B, H, W))
        ws = self.workspace(B, H, W)
        self._check(self.lib.miyolo_write_buffer(self.h, buf, B, H, W, x.data_ptr(), ws.data_ptr(), self._stream()),
                    "miyolo_write_buffer")

    def run_ops(self, first: int, last: int, frames: Optional[torch.Tensor], B: int, H: int, W: int):
        ws = self.workspace(B, H, W)
        xin = None
        if frames is not None:
            xin, B, H, W = self._in(frames)
        self._check(self.lib.miyolo_run_ops(self.h, first, last, _ptr(xin), B, H, W, ws.data_ptr(), ws.numel(),
                                            self._stream()), "miyolo_run_ops")

    def work(self, B: int, H: int, W: int) -> Tuple[float, float]:
        fl, by = C.c_double(), C.c_double()
        self._check(self.lib.miyolo_work(self.h, B, H, W, C.byref(fl), C.byref(by)), "miyolo_work")
        return fl.value, by.value


    def op_work(self, op: int, B: int, H: int, W: int) -> Tuple[float, float]:
        fl, by = C.c_double(), C.c_double()
        self._check(self.lib.miyolo_op_work(self.h, op, B, H, W, C.byref(fl), C.byref(by)), "miyolo_op_work")
        return fl.value, by.value

    def graph_info(self) -> dict:
        """Census of the hipGraph captures (option graph): see miyolo_graph_info."""
        v = (C.c_int32 * 6)()
        self._check(self.lib.miyolo_graph_info(self.h, v), "miyolo_graph_info")
        return dict(zip(("graphs", "nodes", "kernel_nodes", "launches", "rejected", "total_launches"), list(v)))

    def candidate_counts(self, B: int) -> List[int]:
        """Debug: anchors above conf per image as the last detect call's score filter counted them."""
        v = (C.c_int32 * B)()
        self._check(self.lib.miyolo_debug_candidate_counts(self.h, self._ws.data_ptr(), B, v), "miyolo_debug_candidate_counts")
        return list(v)

    def profile_read(self):
        """[(op_index, conv_variant, ms)] of every op launch since profiling was switched on."""
        n = self.lib.miyolo_profile_read(self.h, 0, None, None, None)
        if n <= 0:
            return []
        ops = (C.c_int32 * n)(); cfg = (C.c_int32 * n)(); ms = (C.c_float * n)()
        got = self.lib.miyolo_profile_read(self.h, n, ops, cfg, ms)
        if got < 0:
            self._check(got, "miyolo_profile_read")
        return [(ops[i], cfg[i], ms[i]) for i in range(n)]


def engine_from_weights(sd: Dict[str, torch.Tensor], meta: dict, dtype: str = "f16", device: Optional[int] = None,
                        bgr_input: bool = True, calib_frames: Optional[torch.Tensor] = None, quant=None,
                        gain_fix: bool = False, fuse_head: bool = True, bias_correction: bool = True) -> Engine:
    """dtype "f8": static e4m3 quantisation, calibrated on `calib_frames` (uint8 [N,H,W,3]; default: eight seeded synthetic
    frames at the model's image size - pass real frames / crops for a trained model) through the f16 engine: activation
    ranges and channel means, the latter for the bias correction of the weight rounding (quant.py).  `gain_fix` re-enables
    round 2's fitted per-op gains (off: see quant.py for what they were compensating)."""
    # Detect's two first convs per level run as one (f16 / f32; the fp8 build keeps them apart: one activation scale per op)
    prog = build_program(meta["task"], meta["nc"], meta["scale"], meta.get("spec"), meta.get("nc_quirk", True),
                         fuse_head=(dtype != "f8") and fuse_head)
    if dtype != "f8":
        return Engine(prog, sd, meta["bn_eps"], dtype, device, bgr_input, quant)
    from .quant import calibrate, gain_correction
    if calib_frames is None:
        from .synth import synth_frames
        sz = int(meta.get("imgsz", 640))
        calib_frames = torch.from_numpy(np.concatenate([synth_frames(4, sz, sz, seed=101, kind="noise"),
                                                        synth_frames(4, sz, sz, seed=102, kind="blocks")]))
    eng16 = Engine(prog, sd, meta["bn_eps"], "f16", device, bgr_input) if (quant is None or gain_fix) else None
    if quant is None:
        quant = calibrate(prog, sd, meta["bn_eps"], calib_frames, device, bgr_input, eng16=eng16, bias_correction=bias_correction)
    eng = Engine(prog, sd, meta["bn_eps"], "f8", device, bgr_input, quant)
    eng.gains = {}
    if gain_fix:
        n = max(1, min(4, calib_frames.shape[0], eng.chunk(4, calib_frames.shape[1], calib_frames.shape[2])))
        eng.gains = gain_correction(eng, eng16, calib_frames[:n])
    del eng16
    return eng
