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


def test_greedy_nmm_oracle_known_cases():
    """oracle/post_ref.py greedy_nmm_merge ([3P] sahi GREEDYNMM restated) on cases worked by hand."""
    from oracle.post_ref import greedy_nmm_merge
    d = np.array([
        [0, 0, 10, 10, 0.9, 1],       # keep A
        [2, 2, 8, 8, 0.8, 1],         # inside A: IOS 1 -> absorbed, hull unchanged
        [5, 0, 15, 10, 0.7, 1],       # half of it inside A: IOS exactly 0.5 -> matched (not < 0.5) but NOT absorbed (not > 0.5): dropped
        [9, 0, 19, 10, 0.6, 1],       # IOS 0.1 with A -> its own keep
        [0, 0, 10, 10, 0.95, 2],      # other class: untouched by class 1
        [100, 100, 110, 110, 0.5, 1],
    ], np.float32)
    m, k, mem = greedy_nmm_merge(d)
    assert k.tolist() == [0, 3, 5, 4] and mem == [[1], [], [], []]
    assert np.array_equal(m[0], d[0]) and np.array_equal(m[3], d[4])
    # a chain: B overlaps A enough, C overlaps only the hull of A and B enough... but C must have MATCHED A in the first pass
    d2 = np.array([[0, 0, 10, 10, 0.9, 0], [4, 0, 14, 10, 0.8, 0], [8, 0, 12, 10, 0.7, 0]], np.float32)
    m2, k2, mem2 = greedy_nmm_merge(d2)          # IOS(A,B) = 0.6 absorbed -> hull [0,14]; IOS(A,C) = 20/40 = 0.5 matched; vs hull: 40/40 = 1 > 0.5 absorbed
    assert k2.tolist() == [0] and mem2 == [[1, 2]] and m2[0, :4].tolist() == [0, 0, 14, 10] and m2[0, 4] == np.float32(0.9)
    m3, k3, _ = greedy_nmm_merge(d, class_agnostic=True)
    assert k3[0] == 4 and m3[0, 5] == 2           # agnostic: the 0.95 box of class 2 leads and absorbs class-1 boxes, keeping ITS class
