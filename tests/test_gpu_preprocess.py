"""-m gpu: device-side LetterBox (miyolo_letterbox) against the CPU restatement of cv2's 8-bit linear resize
(oracle/pre_ref.py) - byte work, so the bar is bit-exact.  Shapes: the reference's capture size (detect.py:18,
930 x 1130), common video sizes, up- and down-scaling, identity, odd sizes; rect (auto) and square padding."""
import numpy as np
import pytest
import torch

from manual_yolo_amd.preprocess import letterbox_batch, letterbox_batch_gpu, letterbox_geometry
from manual_yolo_amd.synth import synth_frames
from oracle.pre_ref import letterbox as ref_letterbox

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w", [(930, 1130), (480, 640), (1080, 1920), (640, 640), (333, 517), (200, 150), (37, 1201), (1280, 720)])
@pytest.mark.parametrize("auto", [True, False])
@pytest.mark.parametrize("imgsz", [(640, 640), (1280, 1280), (320, 416)])
def test_letterbox_bit_exact(h, w, auto, imgsz):
    rng = np.random.default_rng(h * 7 + w)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2)]
    frames[1][::7, ::5] = 255                       # hard edges
    y = letterbox_batch_gpu(frames, imgsz, 32, auto=auto).cpu().numpy()
    for i, f in enumerate(frames):
        ref, hw = ref_letterbox(f, imgsz, auto=auto, stride=32)
        assert y[i].shape == ref.shape
        assert np.array_equal(y[i], ref), int(np.abs(y[i].astype(int) - ref.astype(int)).max())


def test_letterbox_matches_host_path_and_geometry():
    frames = list(synth_frames(3, 930, 1130, seed=3, kind="blocks"))
    host = letterbox_batch(frames, (640, 640), 32)
    dev = letterbox_batch_gpu(frames, (640, 640), 32, auto=True).cpu().numpy()
    assert np.array_equal(host, dev)
    nh, nw, top, left, oh, ow = letterbox_geometry((930, 1130), (640, 640), True)
    assert (oh, ow) == host.shape[1:3] and oh % 32 == 0 and ow % 32 == 0


def test_letterbox_device_tensor_input_and_errors():
    x = torch.randint(0, 256, (4, 96, 128, 3), dtype=torch.uint8, device="cuda:0")
    y = letterbox_batch_gpu(x, (64, 64), 32, auto=False)
    assert y.shape == (4, 64, 64, 3) and y.is_cuda
    ref, _ = ref_letterbox(x[2].cpu().numpy(), (64, 64), auto=False)
    assert np.array_equal(y[2].cpu().numpy(), ref)
    with pytest.raises(ValueError):
        letterbox_batch_gpu([np.zeros((8, 8, 3), np.uint8), np.zeros((9, 8, 3), np.uint8)], (64, 64))
    with pytest.raises(ValueError):
        letterbox_batch_gpu(torch.zeros((1, 8, 8, 3)), (64, 64))


# --------------------------------------------------------------------------- crop + classifier transform on the device
def _frame(h, w, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    f[::9, ::4] = 255
    f[5::11] //= 3
    return f


@pytest.mark.parametrize("seed", [0, 1])
def test_crop_resize_byte_exact_against_pil(seed):
    """miyolo_crop_resize vs PIL itself (classify_transform_bgr): small crops (upscaled, as the rank boxes are),
    large ones (antialiased shrink up to 10x), extreme aspect ratios, crops that need no resize on one axis."""
    from manual_yolo_amd.chain import crops_to_classifier_input
    from manual_yolo_amd.preprocess import classify_transform_bgr
    frame = _frame(930, 1130, seed)
    boxes = [(10, 20, 47, 72), (100, 100, 152, 137), (0, 0, 48, 48), (300, 40, 364, 120), (300, 40, 380, 104), (500, 500, 564, 564),
             (5, 5, 34, 305), (600, 10, 1130, 330), (0, 200, 640, 930), (700, 700, 707, 709), (1129, 929, 1130, 930),
             (200, 300, 861, 930), (40, 60, 171, 121)]
    y = crops_to_classifier_input(frame, boxes, 64).cpu().numpy()
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        ref = classify_transform_bgr(frame[y1:y2, x1:x2], 64)
        assert np.array_equal(y[i], ref), (boxes[i], int(np.abs(y[i].astype(int) - ref.astype(int)).max()))


def test_crop_resize_argument_checks():
    from manual_yolo_amd.chain import crops_to_classifier_input
    from manual_yolo_amd.engine import MiyoloError
    frame = _frame(900, 900, 2)
    assert crops_to_classifier_input(frame, [], 64).shape == (0, 64, 64, 3)
    with pytest.raises(ValueError):
        crops_to_classifier_input(frame, [(10, 10, 5, 20)], 64)
    with pytest.raises(MiyoloError):
        crops_to_classifier_input(frame, [(0, 0, 700, 800)], 64)          # short side 700 > 640


def test_classify_boxes_equals_per_crop_calls(golden_dir):
    """chain.classify_boxes (one crop kernel + one classifier batch) == rank_model(safe_crop(...)) per box, the
    loop of detect.py:580-588 / 121-125: same top1, identical probabilities."""
    import os
    from manual_yolo_amd import YOLO
    from manual_yolo_amd.chain import classify_boxes, safe_crop_box
    rank_model = YOLO(os.path.join(golden_dir, "rank_best.safetensors"))
    frame = _frame(930, 1130, 5)
    xyxy = np.array([[100.4, 200.9, 140.2, 262.0], [5, 5, 30, 44], [900, 700, 960, 790], [50, 10, 30, 20], [1100, 900, 1140, 940]], dtype=np.float32)
    got = classify_boxes(rank_model, frame, xyxy, pad=6)
    assert got[3] is None
    for i, b in enumerate(xyxy):
        sb = safe_crop_box(frame.shape[:2], *b, 6)
        if sb is None:
            continue
        r = rank_model(frame[sb[1]:sb[3], sb[0]:sb[2]])[0]
        assert got[i][0] == r.probs.top1
        assert np.allclose(got[i][2], r.probs.data.cpu().numpy(), rtol=0, atol=1e-6)
