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


# ----------------------------------------------------------------------------------------------- sahi GREEDYNMM
def greedy_nmm(boxes: np.ndarray, scores: np.ndarray, match_metric: str = "IOS", match_threshold: float = 0.5):
    """[3P] sahi.postprocess.combine.greedy_nmm (sahi @ 6455e84, reference requirements.txt:76; not vendored) - the
    matching half of the "GREEDYNMM" post-process that ``get_sliced_prediction`` applies by default, i.e. what the
    reference's ``run_sahi`` reaches (pipe.py:186-188 passes no postprocess_* argument: GREEDYNMM, IOS, 0.5, class-aware).

    Candidates of ONE class: boxes [n,4] xyxy fp32, scores [n].  Repeatedly: take the highest-scoring candidate S still
    in the pool, compare it with every other candidate T still in the pool - ``inter / min(area_T, area_S)`` (IOS) or
    ``inter / (area_T - inter + area_S)`` (IOU), all in fp32 as torch computes it - and remove from the pool every T whose
    metric is NOT ``< match_threshold`` (so equality and NaN match); they are listed under S in descending score order.
    Returns ``{keep_index: [matched indices...]}`` in keep order (descending score).  Score ties: sahi sorts with
    ``torch.argsort`` (order of equal scores unspecified); here equal scores keep the LOWER index first."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    areas = (x2 - x1) * (y2 - y1)
    order = list(np.argsort(-scores, kind="stable"))          # descending; sahi pops from the end of an ascending sort
    thr = np.float32(match_threshold)
    zero = np.float32(0)
    out = {}
    with np.errstate(invalid="ignore", divide="ignore"):
        while order:
            i = order.pop(0)
            if not order:
                out[int(i)] = []
                break
            rest = np.asarray(order)
            w = np.maximum(np.minimum(x2[rest], x2[i]) - np.maximum(x1[rest], x1[i]), zero)
            h = np.maximum(np.minimum(y2[rest], y2[i]) - np.maximum(y1[rest], y1[i]), zero)
            inter = w * h
            if match_metric == "IOU":
                val = inter / ((areas[rest] - inter) + areas[i])
            elif match_metric == "IOS":
                val = inter / np.minimum(areas[rest], areas[i])
            else:
                raise ValueError(match_metric)
            unmatched = val < thr
            out[int(i)] = [int(t) for t in rest[~unmatched]]
            order = [int(t) for t in rest[unmatched]]
    return out


def _match_value64(a, b, match_metric: str) -> float:
    """[3P] sahi.postprocess.utils.calculate_bbox_iou / calculate_bbox_ios on two xyxy boxes, in float64 as numpy
    computes them from the predictions' python floats."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    area_a = (a[2] - a[0]) * (a[3] - a[1]); area_b = (b[2] - b[0]) * (b[3] - b[1])
    wh = (np.minimum(a[2:], b[2:]) - np.maximum(a[:2], b[:2])).clip(min=0)
    inter = wh[0] * wh[1]
    with np.errstate(invalid="ignore", divide="ignore"):
        if match_metric == "IOU":
            return float(inter / (area_a + area_b - inter))
        return float(inter / np.minimum(area_a, area_b))


def greedy_nmm_merge(dets: np.ndarray, match_metric: str = "IOS", match_threshold: float = 0.5,
                     class_agnostic: bool = False):
    """[3P] sahi.postprocess.combine.GreedyNMMPostprocess.__call__ (+ batched_greedy_nmm, has_match,
    merge_object_prediction_pair): dets [n,6] rows ``x1,y1,x2,y2,score,cls`` (the shifted per-slice predictions followed by
    the full-frame ones, in sahi's list order).  Per class (ascending class id; all together if ``class_agnostic``) run
    ``greedy_nmm``; then every keep absorbs its matched candidates one by one in descending score order, each only if it
    STILL matches the keep's current (already grown) box - ``metric > threshold`` strictly, in float64 - and absorbing
    means: box = the smallest box containing both, score = the larger score, class = that of the higher-scoring one.  A
    matched candidate that fails this second test is dropped without contributing.
    Returns (merged [m,6] fp32 in sahi's output order: classes ascending, keeps by descending score; keep_index [m];
    members: list of the absorbed candidate indices per output row)."""
    dets = np.asarray(dets, dtype=np.float32).reshape(-1, 6)
    n = dets.shape[0]
    groups = [np.arange(n)] if class_agnostic else [np.nonzero(dets[:, 5] == c)[0] for c in np.unique(dets[:, 5])]
    rows, keeps, members = [], [], []
    for g in groups:
        k2m = greedy_nmm(dets[g, :4], dets[g, 4], match_metric, match_threshold)
        for ki, mlist in k2m.items():
            keep = int(g[ki])
            box = [float(v) for v in dets[keep, :4]]
            score, cls = float(dets[keep, 4]), float(dets[keep, 5])
            took = []
            for mi in mlist:
                m = int(g[mi])
                if _match_value64(box, dets[m, :4], match_metric) > match_threshold:
                    mb = dets[m, :4]
                    box = [min(box[0], float(mb[0])), min(box[1], float(mb[1])), max(box[2], float(mb[2])), max(box[3], float(mb[3]))]
                    if not (score > float(dets[m, 4])):         # get_merged_category: pred1 if pred1.score > pred2.score else pred2
                        cls = float(dets[m, 5])
                    score = max(score, float(dets[m, 4]))
                    took.append(m)
            rows.append(box + [score, cls]); keeps.append(keep); members.append(took)
    merged = np.asarray(rows, dtype=np.float32).reshape(-1, 6)
    return merged, np.asarray(keeps, dtype=np.int64), members
