"""CPU restatement of the detect post-process: non_max_suppression, torchvision nms,
scale_boxes.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

[3P] = third-party code absent from /root/reference:
  ultralytics==8.3.176 ``utils/ops.py`` (non_max_suppression, xywh2xyxy, scale_boxes,
  clip_boxes) and torchvision==0.23.0 ``csrc/ops/cpu/nms_kernel.cpp`` (nms_kernel_impl).
Reference call sites whose results this decides: ``detect.py:541-542``,
``pipe.py:100-135,179``, ``yolo.py:361-373`` (``boxes.xyxy/.conf/.cls`` in NMS keep order).

All arithmetic is IEEE fp32, one rounding per operation (numpy never contracts to FMA),
so a GPU kernel that avoids contraction can be bit-identical.  PARITY UNPINNED: no
reference artefact holds a detection output (``poker_model.pt`` missing).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

MAX_WH = 7680      # [3P] non_max_suppression(max_wh=7680)
MAX_NMS = 30000    # [3P] non_max_suppression(max_nms=30000)


def nms_torchvision(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """[3P] torchvision nms_kernel_impl<float>: stable descending sort by score, greedy,
    suppress j iff ``ovr > iou_threshold`` where ovr is fp32 and the threshold is the C++
    ``double`` argument (the comparison promotes ovr to double), areas are
    ``(x2-x1)*(y2-y1)`` (no +1), ``ovr = inter / (iarea + areas[j] - inter)``.

    Vectorised over j for speed; every element sees exactly the scalar kernel's ops."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")  # stable, descending (ties: lower index first)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = float(iou_thres)
    zero = np.float32(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            rest = rest[~suppressed[rest]]
            if rest.size == 0:
                continue
            xx1 = np.maximum(x1[i], x1[rest])
            yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest])
            yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(zero, xx2 - xx1)
            h = np.maximum(zero, yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def nms_torchvision_scalar(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """Literal double loop of nms_kernel_impl (slow; used to validate the vectorised form)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")
    suppressed = np.zeros(n, dtype=np.uint8)
    keep = []
    f0 = np.float32(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i] == 1:
                continue
            keep.append(i)
            for _j in range(_i + 1, n):
                j = order[_j]
                if suppressed[j] == 1:
                    continue
                xx1 = max(x1[i], x1[j]); yy1 = max(y1[i], y1[j])
                xx2 = min(x2[i], x2[j]); yy2 = min(y2[i], y2[j])
                w = max(f0, np.float32(xx2 - xx1)); h = max(f0, np.float32(yy2 - yy1))
                inter = np.float32(w * h)
                ovr = np.float32(inter / np.float32(np.float32(areas[i] + areas[j]) - inter))
                if float(ovr) > float(iou_thres):
                    suppressed[j] = 1
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(pred: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.7,
                        classes: Optional[Sequence[int]] = None, agnostic: bool = False,
                        max_det: int = 300, max_nms: int = MAX_NMS, max_wh: float = MAX_WH
                        ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """[3P] ultralytics.utils.ops.non_max_suppression (multi_label=False, no masks, no
    time limit - the wall-clock break is deliberately not restated).

    ``pred``: (B, 4+nc, A) fp32 - Detect output ``y`` (xywh in letterboxed px, sigmoid
    class scores).  Returns per image ``(n,6)`` rows ``x1,y1,x2,y2,conf,cls`` in keep
    order and the ``(n,)`` anchor indices (upstream ``return_idxs=True``).

    Defaults are the values the reference's calls reach: conf 0.25 / iou 0.7 / max_det 300
    (``detect.py:541``; ``runs/rank_classifier/args.yaml:39-42``)."""
    pred = np.asarray(pred, dtype=np.float32)
    bsz, no, na = pred.shape
    nc = no - 4
    conf32 = np.float32(conf_thres)      # tensor > python-scalar compares in fp32
    outs, idxs = [], []
    for b in range(bsz):
        p = pred[b].T                                   # (A, 4+nc)
        xc = p[:, 4:].max(1) > conf32
        aidx = np.nonzero(xc)[0]
        x = p[xc]
        # xywh2xyxy: xy - wh/2, xy + wh/2
        half = x[:, 2:4] / np.float32(2)
        box = np.concatenate((x[:, 0:2] - half, x[:, 0:2] + half), 1)
        cls = x[:, 4:]
        if x.shape[0] == 0:
            outs.append(np.zeros((0, 6), np.float32)); idxs.append(np.zeros((0,), np.int64)); continue
        j = cls.argmax(1)                               # first maximal index, as torch.max
        conf = cls[np.arange(cls.shape[0]), j]
        filt = conf > conf32
        x6 = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32)), 1)[filt]
        aidx = aidx[filt]
        if classes is not None:
            filt = np.isin(x6[:, 5], np.asarray(classes, dtype=np.float32))
            x6, aidx = x6[filt], aidx[filt]
        n = x6.shape[0]
        if n == 0:
            outs.append(np.zeros((0, 6), np.float32)); idxs.append(np.zeros((0,), np.int64)); continue
        if n > max_nms:
            o = np.argsort(-x6[:, 4], kind="stable")[:max_nms]
            x6, aidx = x6[o], aidx[o]
        c = x6[:, 5:6] * np.float32(0 if agnostic else max_wh)
        boxes = x6[:, :4] + c
        keep = nms_torchvision(boxes, x6[:, 4], iou_thres)[:max_det]
        outs.append(x6[keep]); idxs.append(aidx[keep])
    return outs, idxs


def scale_boxes(img1_shape: Tuple[int, int], boxes: np.ndarray, img0_shape: Tuple[int, int]) -> np.ndarray:
    """[3P] ultralytics.utils.ops.scale_boxes(padding=True, xywh=False) + clip_boxes:
    undo the letterbox (img1 = network input HxW, img0 = original HxW)."""
    boxes = np.array(boxes, dtype=np.float32, copy=True)
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    boxes[:, 0] -= np.float32(pad_x); boxes[:, 1] -= np.float32(pad_y)
    boxes[:, 2] -= np.float32(pad_x); boxes[:, 3] -= np.float32(pad_y)
    boxes[:, :4] /= np.float32(gain)
    boxes[:, 0] = boxes[:, 0].clip(0, img0_shape[1]); boxes[:, 1] = boxes[:, 1].clip(0, img0_shape[0])
    boxes[:, 2] = boxes[:, 2].clip(0, img0_shape[1]); boxes[:, 3] = boxes[:, 3].clip(0, img0_shape[0])
    return boxes
