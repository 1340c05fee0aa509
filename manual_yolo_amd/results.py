"""Result containers with the surface the reference reads from Ultralytics' ``Results``.

Exercised by the reference at:
  * ``detect.py:122-125``  - ``results.probs.top1``, ``float(results.probs.top1conf)``
  * ``detect.py:542``      - ``sv.Detections.from_ultralytics(results)`` which reads
    ``results.boxes.xyxy/.conf/.cls(.cpu().numpy())``, ``results.names``, ``boxes.id``
  * ``yolo.py:368-373``    - ``for box in results[0].boxes: box.xyxy[0], int(box.cls), float(box.conf)``
  * ``pipe.py:100-135``    - ``res.names``, ``b.xyxy[0]``, ``b.conf``, ``b.cls`` with ``.cpu().numpy()``

Rows of ``Boxes.data`` are ``x1, y1, x2, y2, conf, cls`` in original-image pixels, in NMS
keep order (descending confidence).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch


class _TensorBox:
    def __init__(self, data, orig_shape):
        if isinstance(data, np.ndarray):
            data = torch.from_numpy(data)
        self.data = data
        self.orig_shape = orig_shape

    @property
    def shape(self):
        return self.data.shape

    def cpu(self):
        return self.__class__(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return self.__class__(self.data.cpu().numpy(), self.orig_shape) if isinstance(self.data, torch.Tensor) else self

    def cuda(self):
        return self.__class__(self.data.cuda(), self.orig_shape)

    def to(self, *a, **k):
        return self.__class__(self.data.to(*a, **k), self.orig_shape)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.__class__(self.data[idx], self.orig_shape)


class Boxes(_TensorBox):
    """(n, 6) detections: xyxy, conf, cls."""

    def __init__(self, boxes, orig_shape):
        if isinstance(boxes, np.ndarray):
            boxes = torch.from_numpy(boxes)
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        assert boxes.shape[-1] in (6, 7), f"expected 6 or 7 values per box, got {boxes.shape[-1]}"
        super().__init__(boxes, orig_shape)
        self.is_track = boxes.shape[-1] == 7

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def id(self):
        return self.data[:, -3] if self.is_track else None

    @property
    def xywh(self):
        b = self.xyxy
        out = torch.empty_like(b)
        out[:, 0] = (b[:, 0] + b[:, 2]) / 2
        out[:, 1] = (b[:, 1] + b[:, 3]) / 2
        out[:, 2] = b[:, 2] - b[:, 0]
        out[:, 3] = b[:, 3] - b[:, 1]
        return out

    @property
    def xyxyn(self):
        b = self.xyxy.clone()
        b[:, [0, 2]] /= self.orig_shape[1]
        b[:, [1, 3]] /= self.orig_shape[0]
        return b

    @property
    def xywhn(self):
        b = self.xywh
        b[:, [0, 2]] /= self.orig_shape[1]
        b[:, [1, 3]] /= self.orig_shape[0]
        return b

    def __iter__(self):
        for i in range(len(self.data)):
            yield Boxes(self.data[i:i + 1], self.orig_shape)


class Probs(_TensorBox):
    """(nc,) class probabilities (softmax)."""

    def __init__(self, probs, orig_shape=None):
        super().__init__(probs, orig_shape)

    @property
    def top1(self) -> int:
        return int(self.data.argmax())

    @property
    def top5(self):
        k = min(5, self.data.shape[0])
        return (-self.data).argsort(0)[:k].tolist()

    @property
    def top1conf(self):
        return self.data[self.top1]

    @property
    def top5conf(self):
        return self.data[self.top5]


class Detections:
    """Plain carrier with the fields of ``supervision.Detections`` that the reference reads (``detect.py:542-557,580-584``)."""

    def __init__(self, xyxy, confidence, class_id, tracker_id=None, data=None):
        self.xyxy, self.confidence, self.class_id, self.tracker_id, self.data = xyxy, confidence, class_id, tracker_id, data or {}
        self.mask = None

    def __len__(self):
        return int(self.xyxy.shape[0])


class Results:
    def __init__(self, orig_img, path: str, names: Dict[int, str], boxes=None, probs=None,
                 speed: Optional[dict] = None, anchor_idx=None):
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[:2]) if orig_img is not None else None
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.probs = Probs(probs) if probs is not None else None
        self.masks = None
        self.keypoints = None
        self.obb = None
        self.names = names
        self.path = path
        self.speed = speed or {"preprocess": None, "inference": None, "postprocess": None}
        # not in Ultralytics: index of the anchor each kept box came from (parity checks)
        self.anchor_idx = anchor_idx

    def __len__(self):
        if self.boxes is not None:
            return len(self.boxes)
        if self.probs is not None:
            return len(self.probs)
        return 0

    def cpu(self):
        r = Results(self.orig_img, self.path, self.names, speed=self.speed)
        r.boxes = self.boxes.cpu() if self.boxes is not None else None
        r.probs = self.probs.cpu() if self.probs is not None else None
        r.anchor_idx = self.anchor_idx
        return r

    def to_detections(self) -> "Detections":
        """What ``sv.Detections.from_ultralytics(results)`` extracts (reference ``detect.py:542``):
        numpy ``xyxy [n,4] f32``, ``confidence [n]``, ``class_id [n] int``, ``tracker_id`` (None), ``data['class_name']``,
        in NMS keep order (descending confidence) - the tuple ByteTrack / the reference's annotators consume.
        ``supervision`` itself reads the same attributes (``boxes.xyxy/.conf/.cls/.id``, ``names``, ``masks``, ``obb``),
        so ``sv.Detections.from_ultralytics`` also works on this object unchanged."""
        if self.boxes is None:
            raise AttributeError("classification results carry no boxes")
        b = self.boxes.cpu()
        class_id = b.cls.numpy().astype(int)
        return Detections(xyxy=b.xyxy.numpy(), confidence=b.conf.numpy(), class_id=class_id,
                          tracker_id=b.id.int().numpy() if b.id is not None else None,
                          data={"class_name": np.array([self.names[int(i)] for i in class_id])})

    def summary(self):
        out = []
        if self.boxes is not None:
            for row in self.boxes.data.tolist():
                out.append({"name": self.names.get(int(row[5]), str(int(row[5]))), "class": int(row[5]),
                            "confidence": row[4],
                            "box": {"x1": row[0], "y1": row[1], "x2": row[2], "y2": row[3]}})
        elif self.probs is not None:
            t = self.probs.top1
            out.append({"name": self.names.get(t, str(t)), "class": t, "confidence": float(self.probs.top1conf)})
        return out
