"""-m gpu: sliced ("SAHI-style") inference (reference pipe.py:183-194) as one batched call vs the same pipeline on the
CPU oracle: slice on the host -> RefYolo per slice -> oracle NMS per slice -> shift -> oracle NMS over all candidates."""
import numpy as np
import pytest
import torch

from manual_yolo_amd.model import YOLO
from manual_yolo_amd.sahi import slice_boxes
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
from oracle.post_ref import non_max_suppression
from oracle.yolo_ref import RefYolo

pytestmark = pytest.mark.gpu
NC = 64


def _oracle_sliced(sd, frame_bgr, boxes, sh, sw, conf, iou, max_det):
    ref = RefYolo(sd, "detect", NC, "n", 1e-3, nc_quirk=False)
    cands = []
    for (x1, y1, x2, y2) in boxes:
        canvas = np.full((sh, sw, 3), 114, np.uint8)
        canvas[: y2 - y1, : x2 - x1] = frame_bgr[y1:y2, x1:x2]
        x = torch.from_numpy(canvas[..., ::-1].copy()).permute(2, 0, 1)[None].float() / 255      # BGR -> RGB as the predictor does
        y = ref.forward(x)[0].numpy()
        out, _ = non_max_suppression(y, conf, iou, max_det=max_det)
        d = out[0].copy()
        d[:, [0, 2]] += np.float32(x1); d[:, [1, 3]] += np.float32(y1)
        cands.append(d)
    return cands


def _merge_oracle(cands, iou, max_det):
    allc = np.concatenate(cands) if cands else np.zeros((0, 6), np.float32)
    n = len(allc)
    y = np.zeros((1, 4 + NC, max(n, 1)), np.float32)
    if n:
        y[0, 0] = (allc[:, 0] + allc[:, 2]) / np.float32(2); y[0, 1] = (allc[:, 1] + allc[:, 3]) / np.float32(2)
        y[0, 2] = allc[:, 2] - allc[:, 0]; y[0, 3] = allc[:, 3] - allc[:, 1]
        y[0, 4 + allc[:, 5].astype(int), np.arange(n)] = allc[:, 4]
    out, idx = non_max_suppression(y, 0.0, iou, max_det=max_det)
    return out[0], idx[0], allc


@pytest.fixture(scope="module")
def model():
    sd, meta = synth_state_dict("detect", NC, "n", 0), synth_meta("detect", NC, "n")
    return sd, YOLO((sd, meta))


@pytest.mark.parametrize("H,W,sh,sw,conf", [(448, 704, 256, 256, 0.25), (300, 500, 640, 640, 0.25), (320, 320, 160, 192, 0.01)])
def test_predict_sliced_matches_oracle_pipeline(model, H, W, sh, sw, conf):
    sd, m = model
    frame = synth_frames(1, H, W, seed=21, kind="blocks")[0]
    iou, max_det = 0.7, 100
    res = m.predict_sliced(frame, slice_height=sh, slice_width=sw, overlap_height_ratio=0.2, overlap_width_ratio=0.2,
                           conf=conf, iou=iou, max_det=max_det, perform_standard_pred=False)[0]
    boxes = slice_boxes(H, W, sh, sw, 0.2, 0.2)
    ch, cw = (min(sh, H) + 31) // 32 * 32, (min(sw, W) + 31) // 32 * 32
    cands = _oracle_sliced(sd, frame, boxes, ch, cw, conf, iou, max_det)
    want, widx, allc = _merge_oracle(cands, iou, max_det)
    got = res.boxes.data.cpu().numpy()
    print(f"{len(boxes)} slices, {len(allc)} candidates, kept {len(got)} (oracle {len(want)})")
    assert len(allc) > 5 and len(got) == len(want)
    # candidate slots: slice * max_det + row on the GPU side; the oracle's candidates are packed densely per slice
    offs = np.cumsum([0] + [len(c) for c in cands])
    gslot = res.anchor_idx.cpu().numpy()
    gidx = np.array([offs[s // max_det] + s % max_det for s in gslot])
    assert np.array_equal(gidx, widx), "kept candidates differ"
    assert np.abs(got[:, :4] - want[:, :4]).max() < 2e-2 and np.abs(got[:, 4] - want[:, 4]).max() < 1e-4
    assert np.array_equal(got[:, 5], want[:, 5])


def test_predict_sliced_with_standard_pred_adds_fullframe_candidates(model):
    sd, m = model
    frame = synth_frames(1, 448, 704, seed=22, kind="blocks")[0]
    r0 = m.predict_sliced(frame, 256, 256, perform_standard_pred=False, max_det=100)[0]
    r1 = m.predict_sliced(frame, 256, 256, perform_standard_pred=True, max_det=100)[0]
    rf = m.predict(frame, max_det=100)[0]
    assert len(r1.boxes) >= 1 and len(r0.boxes) >= 1
    full = set(int(s) for s in r1.anchor_idx.tolist() if s >= len(slice_boxes(448, 704, 256, 256)) * 100)
    assert len(full) <= len(rf.boxes)          # boxes taken from the full-frame pass are among its detections
    d = r1.boxes.data
    assert bool((d[:-1, 4] >= d[1:, 4]).all())
