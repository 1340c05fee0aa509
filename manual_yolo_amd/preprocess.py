"""Host-side pre-processing in front of the GPU path (numpy/PIL; not the timed hot path).

detect  : [3P] ultralytics LetterBox (reference ``detect.py:541`` -> predictor.preprocess):
          aspect-preserving resize (cv2.INTER_LINEAR), pad value 114, to a multiple of the
          stride (``auto`` / rect) or to the square ``imgsz``.  Frames stay uint8 BGR HWC:
          the BGR->RGB flip, HWC->CHW and ``/255`` of the reference's preprocess are folded
          into the stem kernel / its weight layout.
classify: the torchvision transforms pickled in ``rank_classifier.pt`` (reference
          ``detect.py:121``): PIL bilinear (antialiased) short-side resize to ``imgsz`` +
          centre crop; ToTensor's ``/255`` again happens in the stem kernel.

SURVEY.md section 8f ranks moving both onto the GPU as the next row after the hot path.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def bilinear_resize_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR) semantics for uint8 HxWxC:
    half-pixel centres, no antialiasing, 11-bit fixed-point weights, two-pass rounding."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[:2]

    def taps(dn: int, sn: int):
        # OpenCV order: the coordinate is rounded to float first; floor and fraction both come from that float
        pos = ((np.arange(dn, dtype=np.float64) + 0.5) * (1.0 / (dn / sn)) - 0.5).astype(np.float32)
        i0 = np.floor(pos).astype(np.int64)
        frac = (pos - i0.astype(np.float32)).astype(np.float32)
        under, over = i0 < 0, i0 >= sn - 1
        frac[under] = 0.0
        i0[under] = 0
        frac[over] = 0.0
        i0[over] = sn - 1
        w1 = np.rint(frac * np.float32(2048.0)).astype(np.int32)
        w0 = np.rint((np.float32(1.0) - frac) * np.float32(2048.0)).astype(np.int32)
        return i0, np.minimum(i0 + 1, sn - 1), w0, w1

    x0, x1, ax0, ax1 = taps(dw, sw)
    y0, y1, by0, by1 = taps(dh, sh)
    s = src.astype(np.int32)
    horiz = s[:, x0] * ax0[None, :, None] + s[:, x1] * ax1[None, :, None]
    top, bot = horiz[y0], horiz[y1]
    out = (((by0[:, None, None] * (top >> 4)) >> 16) + ((by1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img: np.ndarray, new_shape: Tuple[int, int], auto: bool, stride: int = 32,
              pad_value: int = 114) -> np.ndarray:
    h0, w0 = img.shape[:2]
    r = min(new_shape[0] / h0, new_shape[1] / w0)
    new_unpad = (int(round(w0 * r)), int(round(h0 * r)))          # (w, h)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    if (w0, h0) != new_unpad:
        img = bilinear_resize_u8(img, new_unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), pad_value, dtype=np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def letterbox_geometry(shape_hw: Tuple[int, int], new_shape: Tuple[int, int], auto: bool, stride: int = 32):
    """(new_h, new_w, top, left, out_h, out_w) exactly as [3P] LetterBox derives them (same arithmetic as
    ``letterbox`` above)."""
    h0, w0 = shape_hw
    r = min(new_shape[0] / h0, new_shape[1] / w0)
    new_w, new_h = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = new_shape[1] - new_w, new_shape[0] - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_h, new_w, top, left, new_h + top + bottom, new_w + left + right


def letterbox_batch_gpu(frames, imgsz: Tuple[int, int], stride: int = 32, auto: bool = True, device=None, pad_value: int = 114):
    """Same-shape BGR uint8 frames -> letterboxed device tensor [B, H, W, 3] uint8, resized and padded by the HIP
    kernel behind ``miyolo_letterbox`` (bit-exact against oracle/pre_ref.py).  ``frames``: a list of HxWx3 numpy
    arrays of one shape, or a uint8 tensor [B, h, w, 3] (host or device).  Raises if the extension is missing."""
    import torch
    from .engine import MiyoloError, load_library
    lib = load_library()
    if isinstance(frames, torch.Tensor):
        src = frames
    else:
        if len({f.shape for f in frames}) != 1:
            raise ValueError("letterbox_batch_gpu needs frames of one shape (the reference pads mixed batches to the square on the host)")
        src = torch.from_numpy(np.ascontiguousarray(np.stack(frames)))
    if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3:
        raise ValueError("frames must be uint8 [B, h, w, 3]")
    dev = torch.device(device if device is not None else (src.device if src.is_cuda else "cuda:0"))
    src = src.to(dev).contiguous()
    B, h0, w0 = int(src.shape[0]), int(src.shape[1]), int(src.shape[2])
    nh, nw, top, left, oh, ow = letterbox_geometry((h0, w0), tuple(imgsz), auto, stride)
    dst = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.miyolo_letterbox(src.data_ptr(), B, h0, w0, dst.data_ptr(), oh, ow, top, left, nh, nw, pad_value,
                                  torch.cuda.current_stream(dev).cuda_stream)
    if rc:
        raise MiyoloError(f"miyolo_letterbox failed ({rc}): {lib.miyolo_last_error(None).decode()}")
    return dst


def letterbox_batch(frames: Sequence[np.ndarray], imgsz: Tuple[int, int], stride: int = 32) -> np.ndarray:
    """[3P] DetectionPredictor.pre_transform: rect (``auto``) padding only when every frame of
    the batch has the same shape, otherwise pad to the full square."""
    same = len({f.shape for f in frames}) == 1
    return np.stack([letterbox(f, imgsz, auto=same, stride=stride) for f in frames])


def scale_params(net_hw: Tuple[int, int], orig_hw: Tuple[int, int]) -> List[float]:
    """gain, pad_x, pad_y, orig_w, orig_h as [3P] scale_boxes derives them."""
    gain = min(net_hw[0] / orig_hw[0], net_hw[1] / orig_hw[1])
    pad_x = round((net_hw[1] - orig_hw[1] * gain) / 2 - 0.1)
    pad_y = round((net_hw[0] - orig_hw[0] * gain) / 2 - 0.1)
    return [gain, float(pad_x), float(pad_y), float(orig_hw[1]), float(orig_hw[0])]


def classify_transform_bgr(img_bgr: np.ndarray, size: int) -> np.ndarray:
    """BGR uint8 crop -> BGR uint8 size x size (resize/crop act per channel, so doing them in
    BGR and letting the stem read BGR equals the reference's BGR->RGB -> PIL -> transforms)."""
    from PIL import Image
    im = Image.fromarray(np.ascontiguousarray(img_bgr))
    w, h = im.size
    if not ((w <= h and w == size) or (h <= w and h == size)):
        if w < h:
            im = im.resize((size, int(size * h / w)), Image.BILINEAR)
        else:
            im = im.resize((int(size * w / h), size), Image.BILINEAR)
    w, h = im.size
    if w < size or h < size:
        pl = (size - w) // 2 if w < size else 0
        pt = (size - h) // 2 if h < size else 0
        canvas = Image.new("RGB", (max(w, size), max(h, size)))
        canvas.paste(im, (pl, pt))
        im = canvas
        w, h = im.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return np.asarray(im.crop((left, top, left + size, top + size)), dtype=np.uint8)
