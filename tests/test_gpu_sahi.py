"""-m gpu: sliced ("SAHI-style") inference (reference pipe.py:183-194) as one batched call vs the same pipeline on the
CPU oracle: slice on the host -> RefYolo per slice -> oracle NMS per slice -> shift -> merge over all candidates, with
sahi's default GREEDYNMM (IOS 0.5, class-aware: what the reference's call reaches) and with the NMS merge."""
import numpy as np
import pytest
import torch

from manual_yolo_amd.model import YOLO
from manual_yolo_amd.sahi import slice_boxes
from manual_yolo_amd.synth import synth_frames, synth_meta, synth_state_dict
from oracle.post_ref import greedy_nmm_merge, non_max_suppression
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
                           conf=conf, iou=iou, max_det=max_det, perform_standard_pred=False, postprocess_type="NMS")[0]
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
    r0 = m.predict_sliced(frame, 256, 256, perform_standard_pred=False, max_det=100, postprocess_type="NMS")[0]
    r1 = m.predict_sliced(frame, 256, 256, perform_standard_pred=True, max_det=100, postprocess_type="NMS")[0]
    rf = m.predict(frame, max_det=100)[0]
    assert len(r1.boxes) >= 1 and len(r0.boxes) >= 1
    full = set(int(s) for s in r1.anchor_idx.tolist() if s >= len(slice_boxes(448, 704, 256, 256)) * 100)
    assert len(full) <= len(rf.boxes)          # boxes taken from the full-frame pass are among its detections
    d = r1.boxes.data
    assert bool((d[:-1, 4] >= d[1:, 4]).all())


def _sahi_order(merged, keeps):
    """oracle rows (sahi's output order) -> descending score, ties in sahi's order: the order the device writes."""
    o = np.argsort(-merged[:, 4], kind="stable")
    return merged[o], keeps[o]


@pytest.mark.parametrize("H,W,sh,sw,conf,metric,thr,agn", [
    (448, 704, 256, 256, 0.25, "IOS", 0.5, False),        # sahi's defaults (pipe.py:186-188)
    (448, 704, 256, 256, 0.05, "IOS", 0.5, False),        # many candidates, long absorb chains
    (300, 500, 640, 640, 0.25, "IOS", 0.5, False),        # frame smaller than a slice: one slice
    (320, 320, 160, 192, 0.01, "IOU", 0.3, False),
    (448, 704, 256, 256, 0.10, "IOS", 0.5, True),         # class-agnostic merge
])
def test_predict_sliced_greedynmm_matches_oracle(model, H, W, sh, sw, conf, metric, thr, agn):
    """Default merge = sahi's GREEDYNMM: per-slice detections (HIP) merged on the device == oracle greedy_nmm_merge on the
    oracle's per-slice detections: the same keeps, the same absorbed hull boxes (bit-exact given the same candidates;
    2e-2 px / 1e-4 allowed for the fp32 detector in front), the same order."""
    sd, m = model
    frame = synth_frames(1, H, W, seed=21, kind="blocks")[0]
    iou, max_det = 0.7, 100
    res = m.predict_sliced(frame, slice_height=sh, slice_width=sw, conf=conf, iou=iou, max_det=max_det, perform_standard_pred=False,
                           postprocess_match_metric=metric, postprocess_match_threshold=thr, postprocess_class_agnostic=agn)[0]
    boxes = slice_boxes(H, W, sh, sw, 0.2, 0.2)
    ch, cw = (min(sh, H) + 31) // 32 * 32, (min(sw, W) + 31) // 32 * 32
    cands = _oracle_sliced(sd, frame, boxes, ch, cw, conf, iou, max_det)
    allc = np.concatenate(cands)
    allc[:, [0, 2]] = allc[:, [0, 2]].clip(0, W); allc[:, [1, 3]] = allc[:, [1, 3]].clip(0, H)      # sahi clips shifted boxes to the frame
    ok = (allc[:, 0] < allc[:, 2]) & (allc[:, 1] < allc[:, 3])
    merged, keeps, members = greedy_nmm_merge(allc[ok], metric, thr, agn)
    want, wkeep = _sahi_order(merged, np.nonzero(ok)[0][keeps])
    want, wkeep = want[:max_det], wkeep[:max_det]
    got = res.boxes.data.cpu().numpy()
    n_abs = sum(len(t) for t in members)
    print(f"{len(boxes)} slices, {len(allc)} candidates -> {len(merged)} merged boxes ({n_abs} absorbed); device {len(got)}")
    assert len(allc) > 5 and len(got) == len(want)
    offs = np.cumsum([0] + [len(c) for c in cands])
    gslot = res.anchor_idx.cpu().numpy()
    gidx = np.array([offs[s // max_det] + s % max_det for s in gslot])
    assert np.array_equal(gidx, wkeep), "keeps differ"
    assert np.abs(got[:, :4] - want[:, :4]).max() < 2e-2 and np.abs(got[:, 4] - want[:, 4]).max() < 1e-4
    assert np.array_equal(got[:, 5], want[:, 5])


def test_greedynmm_device_merge_is_bit_exact_on_given_candidates():
    """The merge kernel alone on synthetic candidate lists (clusters of near-duplicates, exact ties of the metric at the
    threshold, nested boxes, boxes leaving the frame, empty slices): keeps, hull boxes, scores, classes and order equal
    the oracle's bit for bit, for both metrics and both class modes."""
    from manual_yolo_amd.engine import load_library
    lib = load_library()
    rng = np.random.default_rng(5)
    for trial in range(12):
        ns, md = int(rng.integers(1, 9)), 40
        H, W = 500, 700
        metric, agn = ("IOS", "IOU")[trial % 2], bool(trial // 2 % 2)
        thr = [0.5, 0.3, 0.7][trial % 3]
        dets = np.zeros((ns, md, 6), np.float32); counts = np.zeros(ns, np.int32); org = np.zeros((ns, 4), np.int32)
        for s in range(ns):
            org[s, :2] = rng.integers(0, 300, 2)
            k = int(rng.integers(0, md + 1))
            counts[s] = k
            ctr = rng.uniform(0, 300, (k, 2)).round(); wh = rng.choice([8.0, 16.0, 24.0, 32.0, 64.0], (k, 2))
            if k > 4:                                           # near-duplicates and exact nesting (IOS == 1, == thr cases)
                ctr[1] = ctr[0]; wh[1] = wh[0]
                ctr[2] = ctr[0]; wh[2] = wh[0] / 2
                ctr[3] = ctr[0] + np.array([wh[0, 0] / 2, 0]); wh[3] = wh[0]          # overlap exactly one half: IOS = 0.5
            sc = np.sort(rng.uniform(0.05, 0.99, k).astype(np.float32))[::-1]
            dets[s, :k, 0:2] = ctr - wh / 2; dets[s, :k, 2:4] = ctr + wh / 2
            dets[s, :k, 4] = sc; dets[s, :k, 5] = rng.integers(0, 3, k)
        d, c, o = (torch.from_numpy(a).cuda() for a in (dets, counts, org))
        max_out = 64
        od = torch.empty((max_out, 6), dtype=torch.float32, device="cuda"); oc = torch.empty(2, dtype=torch.int32, device="cuda")
        oi = torch.empty(max_out, dtype=torch.int32, device="cuda")
        rc = lib.miyolo_merge_slices_nmm(d.data_ptr(), c.data_ptr(), o.data_ptr(), ns, md, H, W, {"IOS": 0, "IOU": 1}[metric], thr, int(agn), max_out,
                                         od.data_ptr(), oc.data_ptr(), oi.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        rows, slots = [], []
        for s in range(ns):
            for r in range(counts[s]):
                b = dets[s, r].copy()
                b[[0, 2]] += np.float32(org[s, 0]); b[[1, 3]] += np.float32(org[s, 1])
                b[[0, 2]] = b[[0, 2]].clip(0, W); b[[1, 3]] = b[[1, 3]].clip(0, H)
                if b[0] < b[2] and b[1] < b[3]:
                    rows.append(b); slots.append(s * md + r)
        allc = np.asarray(rows, np.float32).reshape(-1, 6)
        merged, keeps, members = greedy_nmm_merge(allc, metric, thr, agn)
        want, wkeep = _sahi_order(merged, np.asarray(slots)[keeps]) if len(merged) else (merged, keeps)
        n_out, total = int(oc[0]), int(oc[1])
        assert total == len(want) and n_out == min(total, max_out), (trial, total, len(want))
        got = od.cpu().numpy()[:n_out]
        assert np.array_equal(oi.cpu().numpy()[:n_out], wkeep[:n_out]), trial
        assert np.array_equal(got, want[:n_out]), trial
        assert bool((od[n_out:] == 0).all())
