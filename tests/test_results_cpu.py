"""Host logic of the result containers (no GPU): the attribute surface the reference's callers and
``sv.Detections.from_ultralytics`` read (detect.py:542, yolo.py:368-373, pipe.py:100-135)."""
import numpy as np
import torch

from manual_yolo_amd.results import Results


def _from_ultralytics_like(r):
    """The attribute reads of supervision.Detections.from_ultralytics (supervision is not installed here)."""
    if hasattr(r, "obb") and r.obb is not None:
        raise AssertionError
    class_id = r.boxes.cls.cpu().numpy().astype(int)
    assert not r.masks
    return dict(xyxy=r.boxes.xyxy.cpu().numpy(), confidence=r.boxes.conf.cpu().numpy(), class_id=class_id,
                tracker_id=r.boxes.id.int().cpu().numpy() if r.boxes.id is not None else None,
                class_name=np.array([r.names[i] for i in class_id]))


def test_detections_adapter_and_box_iteration():
    rows = torch.tensor([[10., 20., 110., 220., 0.91, 3.], [5., 6., 50., 60., 0.55, 0.], [1., 2., 3., 4., 0.30, 3.]])
    names = {0: "As", 3: "Kd_rank"}
    r = Results(np.zeros((480, 640, 3), np.uint8), "frame0", names, boxes=rows)
    d = r.to_detections()
    ref = _from_ultralytics_like(r)
    assert len(d) == 3 and d.tracker_id is None and ref["tracker_id"] is None
    assert np.array_equal(d.xyxy, ref["xyxy"]) and d.xyxy.dtype == np.float32
    assert np.array_equal(d.confidence, ref["confidence"]) and np.all(np.diff(d.confidence) <= 0)     # keep order = descending
    assert np.array_equal(d.class_id, ref["class_id"]) and d.class_id.dtype.kind == "i"
    assert list(d.data["class_name"]) == ["Kd_rank", "As", "Kd_rank"]
    # yolo.py:368-373 / pipe.py:112-129: iterate boxes, index xyxy[0], int(cls), float(conf)
    got = [(tuple(float(v) for v in b.xyxy[0]), int(b.cls), float(b.conf)) for b in r.boxes]
    assert got[0] == ((10.0, 20.0, 110.0, 220.0), 3, float(np.float32(0.91)))
    assert r.probs is None and r.boxes.id is None


def test_classification_results_have_no_boxes():
    r = Results(np.zeros((64, 64, 3), np.uint8), "crop", {0: "A", 1: "K"}, probs=torch.tensor([0.2, 0.8]))
    assert r.boxes is None and r.probs.top1 == 1 and abs(float(r.probs.top1conf) - 0.8) < 1e-6
    try:
        r.to_detections()
        raise AssertionError("expected AttributeError")
    except AttributeError:
        pass
