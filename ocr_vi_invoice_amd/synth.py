"""Seeded synthetic invoices and text crops (no dataset ships with the reference and there is no network).

* ``make_invoice`` -- one RGB uint8 page (default 960x1280, BASELINE.json configs 2/4/5): near-white paper,
  ~30 text lines made of dark glyph-like strokes, plus the ground-truth line boxes.
* ``make_crops`` -- uint8 RGB text crops of random width.
* ``normalize_chw`` -- /255 then ImageNet mean/std, HWC uint8 -> CHW float32 (pipeline2.py:312-314,124-127).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float64)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float64)


def _draw_line(img: np.ndarray, rng: np.random.Generator, x0: int, y0: int, w: int, h: int) -> None:
    """Fill the box with glyph-like blobs: one random stroke pattern per character cell."""
    cw = max(int(h * 0.6), 3)
    x = x0
    while x + cw <= x0 + w:
        if rng.random() < 0.85:  # else a space
            cell = rng.random((max(h // 3, 1), max(cw // 3, 1))) < 0.55
            cell = np.kron(cell, np.ones((3, 3), dtype=bool))[:h, :cw]
            ink = rng.integers(10, 90)
            sub = img[y0:y0 + cell.shape[0], x:x + cell.shape[1]]
            sub[cell] = ink
        x += cw + 1


def make_invoice(seed: int, height: int = 960, width: int = 1280, lines: int = 30) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (image uint8 [H,W,3], boxes int32 [lines,4] as x,y,w,h)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(235, 256, size=(height, width, 1), dtype=np.uint8).repeat(3, axis=2)
    boxes = []
    pitch = (height - 40) // lines
    for i in range(lines):
        h = int(rng.integers(max(pitch // 2, 8), max(pitch - 4, 9)))
        lo = max(width // 8, 8)
        w = int(rng.integers(lo, max(width - 16, lo + 1)))
        x0 = int(rng.integers(4, max(width - w - 4, 5)))
        y0 = 20 + i * pitch
        _draw_line(img, rng, x0, y0, w, h)
        boxes.append((x0, y0, w, h))
    return img, np.asarray(boxes, dtype=np.int32)


def make_crops(seed: int, count: int, height: int = 48, max_width: int = 320) -> List[np.ndarray]:
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        w = int(rng.integers(max_width // 5, max_width + 1))
        img = rng.integers(225, 256, size=(height, w, 1), dtype=np.uint8).repeat(3, axis=2)
        _draw_line(img, rng, 2, 4, w - 4, height - 8)
        out.append(img)
    return out


def normalize_chw(img_u8: np.ndarray) -> np.ndarray:
    """float64 normalise then cast, as the reference does for detection (pipeline2.py:312-314)."""
    x = (img_u8.astype(np.float32) / 255.0 - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def pad_crop_batch(crops: List[np.ndarray], height: int = 48, width: int = 320) -> np.ndarray:
    """Right-pad with white to ``width`` and normalise -> float32 [B,3,H,W].  (Crops here are already
    ``height`` tall; the reference's resize step, pipeline2.py:100-115, lives in pipeline.py.)"""
    out = np.empty((len(crops), 3, height, width), dtype=np.float32)
    for i, c in enumerate(crops):
        canvas = np.full((height, width, 3), 255, dtype=np.uint8)
        w = min(c.shape[1], width)
        canvas[:, :w] = c[:height, :w]
        out[i] = normalize_chw(canvas)
    return out
