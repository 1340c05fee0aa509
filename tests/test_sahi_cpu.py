"""Slice grid of the sliced-inference path (manual_yolo_amd/sahi.py): [3P] sahi.slicing.get_slice_bboxes restated
(reference pipe.py:43-45: 640 x 640 slices, 20 % overlap).  No GPU needed."""
import numpy as np

from manual_yolo_amd.sahi import slice_boxes


def test_reference_frame_sizes():
    # the reference's screenshots: 1600 x 900 and 1920 x 1200 (roadmap1.v3i.yolov8), 640 slices, 0.2 overlap -> step 512
    b = slice_boxes(900, 1600)
    assert b[0] == (0, 0, 640, 640) and b[1] == (512, 0, 1152, 640) and b[2] == (960, 0, 1600, 640)
    assert b[3] == (0, 260, 640, 900) and len(b) == 6
    b = slice_boxes(1200, 1920)
    assert len(b) == 12 and b[3] == (1280, 0, 1920, 640) and b[-1] == (1280, 560, 1920, 1200)


def test_every_pixel_covered_and_slices_inside():
    rng = np.random.default_rng(0)
    for _ in range(40):
        H, W = int(rng.integers(40, 2200)), int(rng.integers(40, 2600))
        sh, sw = int(rng.integers(32, 900)), int(rng.integers(32, 900))
        oh, ow = float(rng.uniform(0, 0.6)), float(rng.uniform(0, 0.6))
        cover = np.zeros((H, W), bool)
        for (x1, y1, x2, y2) in slice_boxes(H, W, sh, sw, oh, ow):
            assert 0 <= x1 < x2 <= W and 0 <= y1 < y2 <= H
            assert x2 - x1 == min(sw, W) and y2 - y1 == min(sh, H)
            cover[y1:y2, x1:x2] = True
        assert cover.all()


def test_small_frame_is_one_slice():
    assert slice_boxes(300, 500) == [(0, 0, 500, 300)]
