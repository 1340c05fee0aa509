"""CPU restatement of the pre-process steps in front of the network.
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

* classify: the transforms pickled inside ``rank_classifier.pt`` (``model.transforms`` =
  torchvision ``Compose[Resize(64, bilinear, antialias=True), CenterCrop((64,64)), ToTensor,
  Normalize(0,1)]``) applied to a PIL image by [3P] ClassificationPredictor.preprocess
  (reference call ``detect.py:121``).  PIL *is* installed here, and torchvision's PIL code
  path is ``Image.resize`` / ``Image.crop``, so this restatement calls PIL for the resize:
  the result is pinned by the 63/67 known answer.
* detect: [3P] ultralytics LetterBox (``auto`` rect padding or square) + cv2.resize
  INTER_LINEAR (restated from OpenCV's 8-bit fixed-point bilinear: 11-bit coefficients,
  ``(b0*(r0>>4)>>16) + (b1*(r1>>4)>>16) + 2 >> 2``).  cv2 is not installed: UNPINNED.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def classify_transform(img_rgb: np.ndarray, size: int = 64) -> np.ndarray:
    """uint8 HxWx3 RGB -> uint8 size x size x3 (the tensor the net sees is this / 255).

    torchvision Resize(int) on PIL: short side -> size, long side = int(size*long/short);
    CenterCrop: top = int(round((h-size)/2.0)), pads with 0 if smaller (not hit here)."""
    from PIL import Image
    im = Image.fromarray(img_rgb)
    w, h = im.size
    if (w <= h and w == size) or (h <= w and h == size):
        pass
    elif w < h:
        im = im.resize((size, int(size * h / w)), Image.BILINEAR)
    else:
        im = im.resize((int(size * w / h), size), Image.BILINEAR)
    w, h = im.size
    if w < size or h < size:  # torchvision center_crop pads first
        pl = (size - w) // 2 if w < size else 0
        pt = (size - h) // 2 if h < size else 0
        pr = (size - w + 1) // 2 if w < size else 0
        pb = (size - h + 1) // 2 if h < size else 0
        canvas = Image.new("RGB", (w + pl + pr, h + pt + pb))
        canvas.paste(im, (pl, pt))
        im = canvas
        w, h = im.size
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    im = im.crop((left, top, left + size, top + size))
    return np.asarray(im, dtype=np.uint8)


def _cv_round(x: np.ndarray) -> np.ndarray:
    return np.rint(x)  # cvRound = round half to even


def resize_linear_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src, dsize=(w,h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC."""
    dw, dh = dsize
    sh, sw = src.shape[:2]

    def coeffs(dn, sn):
        scale = sn / dn
        d = np.arange(dn, dtype=np.float64)
        f = (d + 0.5) * scale - 0.5
        s = np.floor(f).astype(np.int64)
        f = (f - s).astype(np.float32)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= sn - 1
        f[hi] = 0; s[hi] = sn - 1
        c1 = _cv_round(f.astype(np.float32) * np.float32(2048)).astype(np.int32)
        c0 = _cv_round((np.float32(1) - f) * np.float32(2048)).astype(np.int32)
        s1 = np.minimum(s + 1, sn - 1)
        return s, s1, c0, c1

    sx0, sx1, ax0, ax1 = coeffs(dw, sw)
    sy0, sy1, by0, by1 = coeffs(dh, sh)
    s = src.astype(np.int32)
    rows = s[:, sx0] * ax0[None, :, None] + s[:, sx1] * ax1[None, :, None]      # (sh, dw, C)
    r0 = rows[sy0]; r1 = rows[sy1]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img: np.ndarray, new_shape=(640, 640), auto: bool = True, stride: int = 32,
              pad_value: int = 114):
    """[3P] ultralytics.data.augment.LetterBox(scaleup=True, center=True).
    Returns (padded uint8 image, (net_h, net_w))."""
    shape = img.shape[:2]
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        img = resize_linear_u8(img, new_unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]),
                  pad_value, dtype=np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, out.shape[:2]
