"""Detect -> classify chaining on the device.

The reference loops over the detections of a frame on the host: ``crop = safe_crop(frame, x1, y1, x2, y2, pad=6)``
(``detect.py:100-113,586``) and, for rank classes, ``rank_model(crop)[0]`` one crop at a time
(``detect.py:121-125``).  Here the frame stays on the GPU, all boxes of the frame are cropped + resized by ONE kernel
launch (``miyolo_crop_resize``: Pillow's antialiased bilinear resize to 64 on the short side + centre crop, byte-exact)
and classified as ONE batch.  No CPU fallback: raises if the HIP extension is missing.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import MiyoloError, load_library


def safe_crop_box(shape_hw: Tuple[int, int], x1, y1, x2, y2, pad: int = 6) -> Optional[Tuple[int, int, int, int]]:
    """The index arithmetic of the reference's ``safe_crop`` (``detect.py:100-113``): padded, clamped box or None."""
    h, w = shape_hw
    x1 = max(0, min(w - 1, int(x1 - pad)))
    x2 = max(0, min(w, int(x2 + pad)))
    y1 = max(0, min(h - 1, int(y1 - pad)))
    y2 = max(0, min(h, int(y2 + pad)))
    if x2 <= x1 or y2 <= y1:
        return None
    return x1, y1, x2, y2


def crops_to_classifier_input(frame, boxes: Sequence[Tuple[int, int, int, int]], size: int = 64, device=None) -> torch.Tensor:
    """frame: HxWx3 uint8 (numpy or tensor, host or device); boxes: clamped integer x1,y1,x2,y2.
    Returns the classifier's input batch, uint8 [n, size, size, 3] on the device (same channel order as the frame)."""
    lib = load_library()
    f = frame if isinstance(frame, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame))
    if f.dtype != torch.uint8 or f.dim() != 3 or f.shape[2] != 3:
        raise ValueError("frame must be uint8 [H, W, 3]")
    dev = torch.device(device if device is not None else (f.device if f.is_cuda else "cuda:0"))
    f = f.to(dev).contiguous()
    H, W = int(f.shape[0]), int(f.shape[1])
    bx = np.asarray(boxes, dtype=np.int32).reshape(-1, 4)
    n = int(bx.shape[0])
    out = torch.empty((n, size, size, 3), dtype=torch.uint8, device=dev)
    if n == 0:
        return out
    if (bx[:, 0] < 0).any() or (bx[:, 1] < 0).any() or (bx[:, 2] > W).any() or (bx[:, 3] > H).any() or \
            (bx[:, 2] <= bx[:, 0]).any() or (bx[:, 3] <= bx[:, 1]).any():
        raise ValueError("boxes must be clamped to the frame with x2 > x1 and y2 > y1 (use safe_crop_box)")
    max_short = int(np.minimum(bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]).max())
    bd = torch.from_numpy(bx).to(dev)
    with torch.cuda.device(dev):
        rc = lib.miyolo_crop_resize(f.data_ptr(), H, W, bd.data_ptr(), n, size, max_short, out.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise MiyoloError(f"miyolo_crop_resize failed ({rc}): {lib.miyolo_last_error(None).decode()}")
    return out


def classify_boxes(rank_model, frame, xyxy, pad: int = 6, half: bool = False):
    """Drop-in for the per-detection loop of ``detect.py:580-588``: returns one entry per box - ``None`` where
    ``safe_crop`` would return None, else ``(top1: int, top1conf: float, probs: np.ndarray)`` as
    ``rank_model(crop)[0].probs`` gives them (``detect.py:122-124``)."""
    if rank_model.task != "classify":
        raise ValueError("rank_model must be a classification model")
    shape_hw = tuple(frame.shape[:2])
    boxes, keep = [], []
    for i, b in enumerate(np.asarray(xyxy).reshape(-1, 4)):
        sb = safe_crop_box(shape_hw, b[0], b[1], b[2], b[3], pad)
        if sb is not None:
            boxes.append(sb)
            keep.append(i)
    res: List[Optional[tuple]] = [None] * int(np.asarray(xyxy).reshape(-1, 4).shape[0])
    if not boxes:
        return res
    eng = rank_model.engine("f16" if half else "f32")
    size = int(rank_model.meta["imgsz"]) if not isinstance(rank_model.meta["imgsz"], (tuple, list)) else int(rank_model.meta["imgsz"][0])
    x = crops_to_classifier_input(frame, boxes, size, eng.device)
    logits, probs = eng.classify(x)
    p = probs.float().cpu().numpy()
    for j, i in enumerate(keep):
        t = int(p[j].argmax())
        res[i] = (t, float(p[j, t]), p[j])
    return res
