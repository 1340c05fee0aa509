"""bench.py's parity metric ("mAP delta vs CPU ref"): mAP@[.5:.95] with the oracle's detections as ground truth."""
import numpy as np

import bench


def _dets(rng, n):
    xy = rng.uniform(0, 500, (n, 2)); wh = rng.uniform(20, 100, (n, 2))
    return np.concatenate([xy, xy + wh, rng.uniform(0.3, 1, (n, 1)), rng.integers(0, 5, (n, 1)).astype(float)], 1).astype(np.float32)


def test_identical_detections_give_map_one():
    rng = np.random.default_rng(0)
    g = [_dets(rng, 40), _dets(rng, 3), np.zeros((0, 6), np.float32)]
    assert bench.map50_95([x.copy() for x in g], g) == 1.0


def test_map_drops_with_missing_shifted_and_extra_boxes():
    rng = np.random.default_rng(1)
    g = [_dets(rng, 60), _dets(rng, 60)]
    miss = [x[::2].copy() for x in g]                       # half the boxes missing: recall 0.5
    m1 = bench.map50_95(miss, g)
    assert 0.35 < m1 < 0.65
    shift = [x.copy() for x in g]
    for x in shift:
        x[:, [0, 2]] += 0.12 * (x[:, 2] - x[:, 0])[:, None]  # IoU ~0.79: passes the thresholds up to 0.75 only
    m2 = bench.map50_95(shift, g)
    assert 0.5 < m2 < 0.7
    extra = [np.concatenate([x, _dets(rng, 30) * np.array([1, 1, 1, 1, 0.1, 1], np.float32)]) for x in g]   # low-score false positives
    assert bench.map50_95(extra, g) > 0.99
    assert bench.map50_95([np.zeros((0, 6), np.float32)] * 2, g) == 0.0


def test_roofline_traffic_lookup_finds_the_committed_pmc_passes():
    """bench.py's roofline.traffic comes from profiles/r03_traffic*.json, matched by workload; the headline workload (f16) and
    config 5 (fp8, 1280 x 1280 x 16) must both resolve for their dominant halo-slab kernels."""
    import bench
    t, src = bench.lookup_traffic("conv_h2<f16,k3,wc4,tc6>", 64, 640, 640, "f16")
    assert t is not None and 80e6 < t < 300e6 and "r03_traffic" in src
    t8, src8 = bench.lookup_traffic("conv_h2<f8,k3,wc4,tc3>", 16, 1280, 1280, "f8")
    assert t8 is not None and t8 > 10e6 and "f8" in src8
    assert bench.lookup_traffic("conv_h2<f16,k3,wc4,tc6>", 32, 640, 640, "f16") == (None, None)      # no pass of that workload
