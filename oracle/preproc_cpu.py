"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the host pre-processing either side of the two models:
  * ``normalize_det``  -- src/pipeline/pipeline2.py:312-314 (float32 /255, float64 mean/std, cast to float32)
  * ``crop_image``     -- src/det/test.py:123-130 given the box's bounding rect
  * ``resize_linear_u8`` -- cv2.resize(img, (w, h)) with the default INTER_LINEAR on uint8, restated from OpenCV's published
    algorithm (imgproc/resize.cpp: resizeGeneric_ / HResizeLinear / VResizeLinear<uchar>): coefficients rounded to 11-bit fixed
    point, a horizontal pass producing int32 rows, a vertical pass ``((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2``; and the
    special case where an exact 2x decimation is routed to the 2x2 area filter.
  * ``preprocess_for_recognition`` -- pipeline2.py:92-128.

cv2 is NOT installed in the build container and ships no fixtures in the reference, so agreement with cv2 itself is
**parity unpinned**; this file and the HIP kernel (csrc/preproc.hip) are two independent statements of the same algorithm
(here: separable two-pass over whole arrays; there: per-output-pixel evaluation) and must agree bit for bit.
"""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])


def normalize_det(img_u8: np.ndarray) -> np.ndarray:
    x = img_u8.astype(np.float32) / 255.0          # float32
    x = (x - MEAN) / STD                           # float64 (numpy promotion), as in pipeline2.py:313
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def _axis(ssize: int, dsize: int):
    scale = ssize / dsize
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= ssize - 1
    s[hi], f[hi] = ssize - 1, 0.0
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)   # cvRound: round-half-even
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, ssize - 1), a0, a1


def resize_linear_u8(img: np.ndarray, dsize_wh) -> np.ndarray:
    dw, dh = dsize_wh
    sh, sw = img.shape[:2]
    src = img.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:               # INTER_LINEAR with integer scale 2 -> INTER_AREA fast path
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _axis(sw, dw)
    y0, y1, ay0, ay1 = _axis(sh, dh)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]        # horizontal pass, int
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def crop_image(img: np.ndarray, rect) -> np.ndarray:
    h, w = img.shape[:2]
    x, y, bw, bh = rect
    x, y = max(0, x), max(0, y)
    bw, bh = min(bw, w - x), min(bh, h - y)
    return img[y:y + bh, x:x + bw]


def preprocess_for_recognition(crop: np.ndarray, img_size=(32, 256)) -> np.ndarray:
    th, tw = img_size
    if crop.size == 0:
        return np.zeros((3, th, tw), np.float32)     # pipeline2.py:154-156
    h, w = crop.shape[:2]
    new_w = int(w * (th / h))
    if new_w > tw:
        resized = resize_linear_u8(crop, (tw, th))
    else:
        new_w = max(new_w, 1)
        resized = resize_linear_u8(crop, (new_w, th))
        if tw - new_w > 0:
            resized = np.concatenate([resized, np.full((th, tw - new_w, 3), 255, np.uint8)], axis=1)
    x = resized.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return ((x - MEAN.astype(np.float32).reshape(3, 1, 1)) / STD.astype(np.float32).reshape(3, 1, 1)).astype(np.float32)
