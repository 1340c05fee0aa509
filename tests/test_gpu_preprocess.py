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
