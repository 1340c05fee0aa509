"""``YOLO``: the drop-in for ``ultralytics.YOLO`` on the reference's inference path.

Surface mirrored:
  ``YOLO(path)``                                  reference detect.py:20-21, pipe.py:147, yolo.py:354
  ``model(frame, **kw) -> list[Results]``         detect.py:541 (detect), detect.py:121 (classify)
  ``model.predict(source=, imgsz=, conf=, verbose=)``   pipe.py:179
  ``model.names`` (dict with ``.get``)            detect.py:125,587
  ``model.model.device.type``                     pipe.py:151
``model.train(...)`` (reference class.py:22) is training and out of scope: it raises.

Keyword defaults are Ultralytics' predict defaults (conf 0.25, iou 0.7, max_det 300,
agnostic_nms False, half False; ``runs/rank_classifier/args.yaml:39-42``).  ``half=False``
runs the fp32 kernels (exact-fp32 MFMA: the parity mode), ``half=True`` the fp16 ones.
"""
from __future__ import annotations

import time
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .ckpt import load_weights
from .engine import Engine, engine_from_weights
from .preprocess import classify_transform_bgr, letterbox_batch, letterbox_batch_gpu, scale_params
from .results import Results


class YOLO:
    def __init__(self, model: Union[str, tuple], task: Optional[str] = None, device: Optional[int] = None,
                 verbose: bool = False):
        if isinstance(model, tuple):       # (state_dict, meta): synthetic / already-loaded weights
            self.sd, self.meta = model
            self.ckpt_path = "<memory>"
        else:
            self.sd, self.meta = load_weights(model)
            self.ckpt_path = model
        self.task = task or self.meta["task"]
        if self.task != self.meta["task"]:
            raise ValueError(f"weights are a {self.meta['task']} model, not {self.task}")
        self.names: Dict[int, str] = dict(self.meta["names"])
        self._device_index = device
        self._engines: Dict[str, Engine] = {}
        dev = torch.device("cuda", device if device is not None else 0) if torch.cuda.is_available() else torch.device("cpu")
        # pipe.py:151 probes `self.model.model.device.type`
        self.model = SimpleNamespace(device=dev, names=self.names, stride=32, task=self.task)
        self.overrides = {"imgsz": self.meta["imgsz"]}

    # ------------------------------------------------------------------ engines
    def engine(self, dtype: str = "f32") -> Engine:
        if dtype not in self._engines:
            self._engines[dtype] = engine_from_weights(self.sd, self.meta, dtype, self._device_index, bgr_input=True)
        return self._engines[dtype]

    @property
    def device(self):
        return self.model.device

    def to(self, device):
        return self

    def fuse(self):
        return self

    def train(self, *a, **k):
        raise NotImplementedError("training (reference class.py:22) is outside the MI355X inference path")

    # ------------------------------------------------------------------ predict
    def __call__(self, source=None, **kwargs) -> List[Results]:
        return self.predict(source, **kwargs)

    @staticmethod
    def _as_frames(source) -> List[np.ndarray]:
        if isinstance(source, np.ndarray):
            if source.ndim == 3:
                return [source]
            if source.ndim == 4:
                return list(source)
        if isinstance(source, (list, tuple)) and all(isinstance(s, np.ndarray) for s in source):
            return list(source)
        raise TypeError("source must be an HxWx3 BGR uint8 numpy array or a list of them")

    def predict_sliced(self, source, slice_height: int = 640, slice_width: int = 640, overlap_height_ratio: float = 0.2,
                       overlap_width_ratio: float = 0.2, conf: Optional[float] = None, iou: float = 0.7, max_det: int = 300,
                       half: bool = False, agnostic_nms: bool = False, classes=None, perform_standard_pred: bool = True,
                       imgsz=None, postprocess_type: str = "GREEDYNMM", postprocess_match_metric: str = "IOS",
                       postprocess_match_threshold: float = 0.5, postprocess_class_agnostic: bool = False) -> List[Results]:
        """Sliced inference as the reference's ``run_sahi`` does it (``pipe.py:183-194``: 640 x 640 slices, 20 % overlap,
        every other ``get_sliced_prediction`` argument at sahi's default), as ONE batched call: device-side slicing, one
        ``miyolo_detect`` over all slices, device-side merge (sahi.py).  ``perform_standard_pred`` (sahi's default): the
        full-frame prediction joins the candidates, last.  ``postprocess_*``: sahi's names and defaults - GREEDYNMM with
        IOS 0.5, class-aware; ``postprocess_type="NMS"`` merges with the model's class-aware NMS at ``iou`` instead.
        ``conf`` is the per-slice score threshold (sahi's ``AutoDetectionModel`` default is 0.3, Ultralytics' 0.25 is ours)."""
        from .sahi import slice_boxes
        if self.task != "detect":
            raise ValueError("sliced inference is a detection feature")
        frames = self._as_frames(source)
        eng = self.engine("f16" if half else "f32")
        conf = 0.25 if conf is None else conf
        out = []
        for i, f in enumerate(frames):
            H, W = f.shape[:2]
            boxes = slice_boxes(H, W, slice_height, slice_width, overlap_height_ratio, overlap_width_ratio)
            sh = (min(slice_height, H) + 31) // 32 * 32
            sw = (min(slice_width, W) + 31) // 32 * 32
            extra = None
            eng.set_classes(classes)
            if perform_standard_pred:
                x = letterbox_batch_gpu([f], tuple((imgsz, imgsz) if isinstance(imgsz, int) else (imgsz or (self.meta["imgsz"],) * 2)), 32, auto=True, device=eng.device)
                scale = torch.tensor([scale_params(tuple(x.shape[1:3]), (H, W))], dtype=torch.float32, device=eng.device)
                d0, c0, _ = eng.detect(x, conf, iou, agnostic_nms, max_det, scale, want_anchor=False)
                extra = (d0, c0)
            d, c, idx = eng.detect_sliced(torch.from_numpy(np.ascontiguousarray(f)), boxes, (sh, sw), conf, iou, agnostic_nms, max_det, extra,
                                          merge=postprocess_type, match_metric=postprocess_match_metric, match_threshold=postprocess_match_threshold,
                                          merge_agnostic=postprocess_class_agnostic if postprocess_type.upper() == "GREEDYNMM" else None)
            n = int(c)
            out.append(Results(f, f"image{i}.jpg", self.names, boxes=d[:n].cpu().clone(), anchor_idx=idx[:n].cpu().clone()))
        return out

    def predict(self, source=None, stream: bool = False, imgsz=None, conf: Optional[float] = None, iou: float = 0.7,
                max_det: int = 300, half: bool = False, agnostic_nms: bool = False, classes=None,
                verbose: bool = False, device=None, **unused) -> List[Results]:
        frames = self._as_frames(source)
        for f in frames:
            if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
                raise TypeError("frames must be HxWx3 uint8 (BGR, as cv2/mss deliver them)")
        eng = self.engine("f16" if half else "f32")
        conf = 0.25 if conf is None else conf
        imgsz = imgsz or self.meta["imgsz"]
        if isinstance(imgsz, int):
            imgsz = (imgsz, imgsz)
        t0 = time.perf_counter()
        if self.task == "detect":
            if len({f.shape for f in frames}) == 1:
                # one shape (a video stream): raw frames go up once, resize + pad run on the GPU (miyolo_letterbox)
                x = letterbox_batch_gpu(frames, tuple(imgsz), 32, auto=True, device=eng.device)
                net_hw = tuple(x.shape[1:3])
            else:                                   # mixed shapes: the reference pads to the square on the host
                batch = letterbox_batch(frames, tuple(imgsz), 32)
                x = torch.from_numpy(batch).to(eng.device, non_blocking=True)
                net_hw = batch.shape[1:3]
            scale = torch.tensor([scale_params(net_hw, f.shape[:2]) for f in frames], dtype=torch.float32,
                                 device=eng.device)
            torch.cuda.synchronize(eng.device)
            t1 = time.perf_counter()
            eng.set_classes(classes)          # filter on the device, before NMS, as [3P] non_max_suppression(classes=...)
            dets, counts, anchor = eng.detect(x, conf, iou, agnostic_nms, max_det, scale)
            torch.cuda.synchronize(eng.device)
            t2 = time.perf_counter()
            dets_c, counts_c, anchor_c = dets.cpu(), counts.cpu().tolist(), anchor.cpu()
            out = []
            for i, f in enumerate(frames):
                n = counts_c[i]
                d = dets_c[i, :n].clone()
                a = anchor_c[i, :n].clone()
                out.append(Results(f, f"image{i}.jpg", self.names, boxes=d, anchor_idx=a))
            t3 = time.perf_counter()
        else:
            size = int(imgsz[0])
            batch = np.stack([classify_transform_bgr(f, size) for f in frames])
            x = torch.from_numpy(batch).to(eng.device, non_blocking=True)
            torch.cuda.synchronize(eng.device)
            t1 = time.perf_counter()
            logits, probs = eng.classify(x)
            torch.cuda.synchronize(eng.device)
            t2 = time.perf_counter()
            probs_c = probs.cpu()
            out = [Results(f, f"image{i}.jpg", self.names, probs=probs_c[i].clone()) for i, f in enumerate(frames)]
            t3 = time.perf_counter()
        n = max(len(frames), 1)
        speed = {"preprocess": (t1 - t0) * 1e3 / n, "inference": (t2 - t1) * 1e3 / n, "postprocess": (t3 - t2) * 1e3 / n}
        for r in out:
            r.speed = speed
        if verbose:
            print(f"Speed: {speed['preprocess']:.1f}ms preprocess, {speed['inference']:.1f}ms inference, "
                  f"{speed['postprocess']:.1f}ms postprocess per image at shape {tuple(x.shape)}")
        return out
