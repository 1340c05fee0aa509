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
    """cv2.resize(src, dsize=(w,h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC - restated from OpenCV's source,
    unverified against cv2 (not installed; no reference fixture)."""
    dw, dh = dsize
    sh, sw = src.shape[:2]

    def coeffs(dn, sn):
        # OpenCV resize.cpp (INTER_LINEAR, 8-bit): inv_scale = dsize / ssize; scale = 1. / inv_scale (doubles);
        #   fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = cvFloor(fx);  fx -= sx;
        # i.e. the source coordinate is rounded to FLOAT first, and both the floor and the fraction come from that float.
        # (cv2 is not installed here and the reference holds no resize fixture: restated from the published source,
        # unverified against cv2 itself.)
        scale = 1.0 / (dn / sn)
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
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


# ----------------------------------------------------------------------------------------------------
# Pillow's 8-bit resample, restated (test infrastructure: checked against PIL itself in tests/test_oracle_kat.py,
# and the reference for the device kernel behind miyolo_crop_resize).  Follows Pillow src/libImaging/Resample.c:
# precompute_coeffs (support = filter support x max(scale, 1), window [int(c - s + .5), int(c + s + .5)),
# weights normalised in double), normalize_coeffs_8bpc (22-bit fixed point, round half away from zero),
# ImagingResampleHorizontal_8bpc then ImagingResampleVertical_8bpc with an 8-bit intermediate image.
_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_coeffs(in_size: int, out_size: int):
    scale = float(np.float32(in_size) - np.float32(0)) / out_size        # (double)(in1 - in0) / outSize, in0/in1 are C floats
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                                          # bilinear: support 1.0
    ss = 1.0 / filterscale
    bounds, coefs = [], []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        if ww != 0.0:
            k = [v / ww for v in k]
        kk = [int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS)) for v in k]
        bounds.append((xmin, xmax))
        coefs.append(kk)
    return bounds, coefs


def _pil_pass(src: np.ndarray, out_size: int) -> np.ndarray:
    """Resample axis 1 of src [rows][in][C] uint8 to out_size."""
    bounds, coefs = _pil_coeffs(src.shape[1], out_size)
    out = np.empty((src.shape[0], out_size, src.shape[2]), dtype=np.uint8)
    s = src.astype(np.int64)
    for xx, ((xmin, xmax), kk) in enumerate(zip(bounds, coefs)):
        acc = np.full((src.shape[0], src.shape[2]), 1 << (_PIL_PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += s[:, xmin + x, :] * kk[x]
        out[:, xx, :] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_bilinear_u8(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """Image.fromarray(img).resize(size_wh, Image.BILINEAR) for uint8 HxWx3 (antialiased when shrinking)."""
    ow, oh = size_wh
    h, w = img.shape[:2]
    cur = img
    if ow != w:
        cur = _pil_pass(cur, ow)                                         # horizontal first, 8-bit intermediate
    if oh != h:
        cur = _pil_pass(cur.transpose(1, 0, 2), oh).transpose(1, 0, 2)   # then vertical
    return np.ascontiguousarray(cur)


def classify_transform_restated(img: np.ndarray, size: int = 64) -> np.ndarray:
    """classify_transform without PIL: short side -> size (long = int(size * long / short)), centre crop.
    Same channel order in and out.  Only the shapes the reference path produces (no side shorter than `size`
    after the resize, i.e. no padding branch)."""
    h, w = img.shape[:2]
    if not ((w <= h and w == size) or (h <= w and h == size)):
        if w < h:
            img = pil_resize_bilinear_u8(img, (size, int(size * h / w)))
        else:
            img = pil_resize_bilinear_u8(img, (int(size * w / h), size))
    h, w = img.shape[:2]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return np.ascontiguousarray(img[top:top + size, left:left + size])
