"""Sliced ("SAHI-style") inference - SURVEY.md 8f rank 4.

The reference re-runs its detector on 640 x 640 slices with 20 % overlap when the full-frame pass found few or small
objects (``pipe.py:43-45`` constants, ``pipe.py:183-194`` ``run_sahi`` -> [3P] ``sahi.get_sliced_prediction``,
``pipe.py:288-302`` the trigger).  ``sahi`` is not installed here and not vendored; this module restates the part of its
published algorithm that shapes the work - the slice grid - and replaces its host loop of per-slice predictions by ONE
batched ``miyolo_detect`` over all slices, with the merge done on the device by the same class-aware NMS as the per-frame
batched ``miyolo_detect`` over all slices, with the merge done on the device.  The merge the reference's call reaches is
[3P] sahi's default, "GREEDYNMM" with IOS 0.5, class-aware (``pipe.py:186-188`` passes no ``postprocess_*`` argument):
``miyolo_merge_slices_nmm`` (csrc/nmm.h; restated in oracle/post_ref.py ``greedy_nmm_merge``), the default of
``YOLO.predict_sliced`` since round 3.  ``postprocess_type="NMS"`` keeps round 2's merge by the model's class-aware NMS
(``miyolo_merge_slices``).
"""
from __future__ import annotations

from typing import List, Tuple


def slice_boxes(height: int, width: int, slice_h: int = 640, slice_w: int = 640, overlap_h: float = 0.2,
                overlap_w: float = 0.2) -> List[Tuple[int, int, int, int]]:
    """[3P] sahi.slicing.get_slice_bboxes: rows of slices from the top-left, each step ``slice - int(overlap * slice)``;
    a slice that would cross the right / bottom edge is moved back inside (so the last slices overlap more).  Returns
    (x1, y1, x2, y2) boxes; every pixel of the frame is covered."""
    y_overlap, x_overlap = int(overlap_h * slice_h), int(overlap_w * slice_w)
    out = []
    y_max = y_min = 0
    while y_max < height:
        x_min = x_max = 0
        y_max = y_min + slice_h
        while x_max < width:
            x_max = x_min + slice_w
            if y_max > height or x_max > width:
                xmax, ymax = min(width, x_max), min(height, y_max)
                out.append((max(0, xmax - slice_w), max(0, ymax - slice_h), xmax, ymax))
            else:
                out.append((x_min, y_min, x_max, y_max))
            x_min = x_max - x_overlap
        y_min = y_max - y_overlap
    return out
