"""Pin the oracle against the reference's own known answers (CPU only).

Known answers (reference artefacts):
  runs/rank_classifier/results.csv:21   best.pt  top-1 0.9403 (63/67), top-5 0.98507 (66/67)
  runs/rank_classifier/results.csv:22   last.pt  top-1 0.91045 (61/67)
  runs/rank_classifier/confusion_matrix.png      diagonal + the four off-diagonal cells
"""
import numpy as np
import torch

from oracle.pre_ref import classify_transform
from oracle.yolo_ref import RefYolo, YOLOV8_DET_SPEC, count_params, parse_spec

CONF_DIAG = {"10": 4, "2": 6, "3": 5, "4": 4, "5": 4, "6": 4, "7": 6, "8": 5, "9": 4, "A": 7, "J": 4, "K": 5, "Q": 5}
CONF_OFF = {("10", "Q"), ("8", "3"), ("9", "6"), ("Q", "6")}  # (true, pred)


def _run(bundle, pre_u8, fuse=True):
    sd, meta = bundle
    ref = RefYolo(sd, "classify", meta["nc"], meta["scale"], meta["bn_eps"], fuse=fuse)
    x = torch.from_numpy(pre_u8).permute(0, 3, 1, 2).float() / 255.0
    return ref.forward(x), meta


def test_best_pt_known_answers(rank_bundles, rank_valid):
    (probs, logits), meta = _run(rank_bundles["best"], rank_valid["pre_u8"])
    names = meta["names"]
    labels = rank_valid["labels"]
    top1 = probs.argmax(1).numpy()
    assert (top1 == labels).sum() == 63 and len(labels) == 67           # results.csv:21
    top5 = (-probs).argsort(1)[:, :5].numpy()
    assert sum(int(l in t) for l, t in zip(labels, top5)) == 66          # 0.98507 * 67
    diag = {}
    off = set()
    for l, p in zip(labels, top1):
        if l == p:
            diag[names[int(l)]] = diag.get(names[int(l)], 0) + 1
        else:
            off.add((names[int(l)], names[int(p)]))
    assert diag == CONF_DIAG and off == CONF_OFF                         # confusion_matrix.png
    # val loss 0.2352 (results.csv:21, mean CE over the val set computed in batches of 64)
    ce = torch.nn.functional.cross_entropy(logits, torch.from_numpy(labels).long()).item()
    assert abs(ce - 0.2352) < 5e-3


def test_last_pt_known_answer(rank_bundles, rank_valid):
    (probs, _), _ = _run(rank_bundles["last"], rank_valid["pre_u8"])
    assert (probs.argmax(1).numpy() == rank_valid["labels"]).sum() == 61  # results.csv:22


def test_fused_equals_unfused(rank_bundles, rank_valid):
    (p1, l1), _ = _run(rank_bundles["best"], rank_valid["pre_u8"], fuse=True)
    (p2, l2), _ = _run(rank_bundles["best"], rank_valid["pre_u8"], fuse=False)
    assert (l1 - l2).abs().max().item() < 2e-4


def test_golden_logits_reproduce(rank_bundles, rank_valid):
    for tag in ("best", "last"):
        (probs, logits), _ = _run(rank_bundles[tag], rank_valid["pre_u8"])
        assert np.abs(logits.numpy() - rank_valid[f"logits_{tag}"]).max() < 1e-5
        assert np.abs(probs.numpy() - rank_valid[f"probs_{tag}"]).max() < 1e-6


def test_preprocess_reproduces_fixture(rank_valid):
    flat, shapes = rank_valid["raw_rgb_flat"], rank_valid["raw_shapes"]
    off = 0
    for i, (h, w) in enumerate(shapes):
        im = flat[off:off + h * w * 3].reshape(h, w, 3)
        off += h * w * 3
        assert np.array_equal(classify_transform(im, 64), rank_valid["pre_u8"][i])


def test_yolov8m_param_count_matches_model_card():
    # Ultralytics model card: "YOLOv8m summary: 25,902,640 parameters" at nc=80
    from manual_yolo_amd.synth import synth_state_dict
    sd = synth_state_dict("detect", nc=80, scale="m", seed=0, calibrate=False)
    assert count_params(sd) == 25_902_640
    layers = parse_spec(YOLOV8_DET_SPEC, 80, "m")
    assert [l["c2"] for l in layers[:10]] == [48, 96, 96, 192, 192, 384, 384, 576, 576, 576]


def test_nc_quirk_changes_stem_width():
    # upstream parse_model leaves a channel count equal to nc unscaled (see parse_spec docstring)
    assert parse_spec(YOLOV8_DET_SPEC, 64, "m", nc_quirk=True)[0]["c2"] == 64
    assert parse_spec(YOLOV8_DET_SPEC, 64, "m", nc_quirk=False)[0]["c2"] == 48


def test_pillow_resample_restatement_is_pinned_against_pil():
    """oracle/pre_ref.py restates Pillow's 8-bit bilinear resample (the classifier's Resize transform) so that the
    device kernel can be checked without PIL in the loop; here the restatement itself is checked against PIL."""
    from PIL import Image
    from oracle.pre_ref import classify_transform_restated, pil_resize_bilinear_u8
    rng = np.random.default_rng(0)
    for (h, w), (ow, oh) in [((37, 52), (64, 90)), ((120, 47), (64, 163)), ((200, 300), (96, 64)), ((64, 64), (64, 64)),
                             ((31, 33), (64, 68)), ((500, 260), (64, 123)), ((17, 400), (1505, 64))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(pil_resize_bilinear_u8(img, (ow, oh)), ref), ((h, w), (ow, oh))
    for h, w in [(37, 52), (52, 37), (48, 48), (130, 61), (64, 80), (80, 64), (64, 64), (29, 300), (640, 700)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(classify_transform_restated(img, 64), classify_transform(img, 64)), (h, w)


def test_safe_crop_box_matches_reference_arithmetic():
    """manual_yolo_amd/chain.py safe_crop_box = the index arithmetic of detect.py:100-113."""
    from manual_yolo_amd.chain import safe_crop_box
    assert safe_crop_box((930, 1130), 10.7, 20.2, 50.9, 70.1, 6) == (4, 14, 56, 76)
    assert safe_crop_box((930, 1130), -5, -5, 3, 3, 6) == (0, 0, 9, 9)
    assert safe_crop_box((930, 1130), 1125, 925, 1140, 940, 6) == (1119, 919, 1130, 930)
    assert safe_crop_box((100, 100), 200, 200, 210, 210, 6) == (99, 99, 100, 100)   # off-frame box: the reference keeps a 1x1 crop
    assert safe_crop_box((100, 100), 50, 10, 30, 20, 6) is None                      # inverted beyond the pad
