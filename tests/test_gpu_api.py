"""-m gpu: the drop-in YOLO object (reference call sites detect.py:121,541; pipe.py:179; yolo.py:361)."""
import os

import numpy as np
import pytest
import torch

from manual_yolo_amd import YOLO
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict

pytestmark = pytest.mark.gpu


def _crops(rank_valid):
    flat, shapes = rank_valid["raw_rgb_flat"], rank_valid["raw_shapes"]
    off, out = 0, []
    for h, w in shapes:
        out.append(np.ascontiguousarray(flat[off:off + h * w * 3].reshape(h, w, 3)[..., ::-1]))   # RGB -> BGR as cv2 gives
        off += h * w * 3
    return out


def test_rank_classifier_drop_in(golden_dir, rank_valid):
    """rank_model = YOLO(RANK_MODEL_PATH); results = rank_model(crop)[0]; results.probs.top1 ...
    (detect.py:21,121-125) on the 67 reference crops -> the reference's 63/67."""
    rank_model = YOLO(os.path.join(golden_dir, "rank_best.safetensors"))
    assert rank_model.names.get(9) == "A" and rank_model.model.device.type == "cuda"
    labels = rank_valid["labels"]
    hits = 0
    for crop, lab in zip(_crops(rank_valid), labels):
        results = rank_model(crop)[0]
        assert results.boxes is None and results.probs is not None
        top = results.probs.top1
        conf = float(results.probs.top1conf)
        assert 0 < conf <= 1 and rank_model.names.get(top, "") != ""
        hits += int(top == lab)
    assert hits == 63
    # batched call: list in, list out, same answers
    rs = rank_model(_crops(rank_valid)[:9])
    assert len(rs) == 9 and all(r.probs.data.shape == (13,) for r in rs)


def test_detector_drop_in_surface():
    sd, meta = synth_state_dict("detect", 64, "n", 0), synth_meta("detect", 64, "n")
    model = YOLO((sd, meta))
    frame = synth_frames(1, 930, 1130, seed=6, kind="blocks")[0][:, :, ::-1].copy()   # detect.py:18-sized frame, BGR
    results = model(frame)[0]                                                # detect.py:541
    assert results.probs is None and results.boxes is not None and results.names is model.names
    n = len(results.boxes)
    assert n > 0
    assert results.boxes.xyxy.shape == (n, 4) and results.boxes.id is None
    conf = results.boxes.conf.cpu().numpy()
    assert np.all(np.diff(conf) <= 0)                                        # keep order = descending confidence
    xy = results.boxes.xyxy.cpu().numpy()
    assert xy.min() >= 0 and xy[:, [0, 2]].max() <= 1130 and xy[:, [1, 3]].max() <= 930
    for b in results.boxes:                                                  # yolo.py:368-373
        x1, y1, x2, y2 = b.xyxy[0].tolist()
        assert isinstance(int(b.cls), int) and 0.25 < float(b.conf) <= 1.0
        break
    r2 = model.predict(source=frame, imgsz=1280, conf=0.35, verbose=False)   # pipe.py:179
    assert len(r2) == 1 and (len(r2[0].boxes) == 0 or float(r2[0].boxes.conf.min()) > 0.35)
    r3 = model(frame, conf=0.5, half=True)                                   # yolo.py:361 + fp16 kernels
    assert len(r3[0].boxes) <= n + 5
    with pytest.raises(NotImplementedError):
        model.train(data="x")
    with pytest.raises(TypeError):
        model("not an image")
