"""DB post-processing: the host C++ implementation behind ocrvi_db_postprocess against the independent
Python statement in oracle/dbpost_cpu.py -- bit-exact polygons, scores to 1e-6 -- plus hand-computed known answers.

Parity with cv2 / pyclipper / shapely themselves is UNPINNED: none of them is importable here and the reference holds no fixture for this
stage.  What these tests pin is that two independently written statements of the published algorithms (Suzuki-Abe border following with
CHAIN_APPROX_SIMPLE, OpenCV's approxPolyDP, fillPoly-style mask mean, Clipper 6.4.2 round-join offset) agree exactly, and the
hand-derivable cases.  No GPU needed: this stage runs on the host in the reference as well (pipeline2.py:320-321).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import dbpost_cpu as O  # noqa: E402
from ocr_vi_invoice_amd import pipeline as P  # noqa: E402


def _run_both(prob, **kw):
    pp = P.DBPostProcessor(**{k: v for k, v in kw.items() if k != "min_area"})
    if "min_area" in kw:
        pp.min_area = kw["min_area"]
    boxes, scores = pp(prob[None])
    oboxes, oscores = O.db_postprocess(prob[None], thresh=pp.thresh, box_thresh=pp.box_thresh, max_candidates=pp.max_candidates,
                                       unclip_ratio=pp.unclip_ratio, min_area=pp.min_area)
    assert len(boxes) == len(oboxes), (len(boxes), len(oboxes))
    for b, ob in zip(boxes, oboxes):
        assert b.shape == ob.shape and np.array_equal(b, ob)
    np.testing.assert_allclose(scores, oscores, rtol=0, atol=1e-6)
    return boxes, scores


def _blobs_map(seed, H, W, n_blobs, soft=True):
    """Random text-like blobs: axis-aligned and rotated bars, L shapes, rings, specks; soft probabilities."""
    rng = np.random.default_rng(seed)
    prob = rng.uniform(0.0, 0.25, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_blobs):
        kind = rng.integers(0, 5)
        cx, cy = rng.integers(0, W), rng.integers(0, H)
        w, h = rng.integers(6, max(W // 3, 8)), rng.integers(3, 14)
        val = rng.uniform(0.55, 0.98) if soft else 1.0
        if kind == 0:
            m = (abs(xx - cx) <= w // 2) & (abs(yy - cy) <= h // 2)
        elif kind == 1:
            th = rng.uniform(-0.5, 0.5)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
            v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            m = (abs(u) <= w / 2) & (abs(v) <= h / 2)
        elif kind == 2:
            m = ((abs(xx - cx) <= w // 2) & (abs(yy - cy) <= h // 2)) & ~((xx > cx) & (yy < cy))
        elif kind == 3:
            r = np.hypot(xx - cx, (yy - cy) * 2.0)
            m = (r <= w / 2) & (r >= w / 4)
        else:
            m = (abs(xx - cx) <= 1) & (abs(yy - cy) <= rng.integers(0, 2))
        prob[m] = val
    if soft:
        prob += rng.uniform(-0.04, 0.04, (H, W)).astype(np.float32)
    return np.clip(prob, 0, 1).astype(np.float32)


def test_rectangle_known_answer():
    """40x12 solid rectangle at (10,20): contour = its 4 corners in cv2's order, perimeter 2*(39+11) = 100, area 39*11 = 429,
    offset distance 429*1.6/100 = 6.864 -> the round-join offset's extent is the rectangle grown by 6.864 and rounded."""
    prob = np.full((64, 64), 0.1, np.float32)
    prob[20:32, 10:50] = 0.9
    c = O.find_contours(prob > 0.3)
    assert len(c) == 1 and c[0].tolist() == [[10, 20], [10, 31], [49, 31], [49, 20]]
    assert O.arc_length_closed(c[0]) == 100.0 and O.contour_area(c[0]) == 429.0
    boxes, scores = _run_both(prob, unclip_ratio=1.6)
    assert len(boxes) == 1 and abs(scores[0] - 0.9) < 1e-6
    b = boxes[0]
    assert (b[:, 0].min(), b[:, 0].max(), b[:, 1].min(), b[:, 1].max()) == (3, 56, 13, 38)
    # every vertex of a round-join offset lies within half a pixel (rounding) of distance d from the rectangle
    d = 6.864
    dx = np.maximum(np.maximum(10 - b[:, 0], b[:, 0] - 49), 0)
    dy = np.maximum(np.maximum(20 - b[:, 1], b[:, 1] - 31), 0)
    assert np.all(np.abs(np.hypot(dx, dy) - d) <= 0.75)


def test_filters_known_answer():
    """box_thresh, min_area and the <4-vertex rule each drop exactly the blob built to trip them."""
    prob = np.full((80, 120), 0.05, np.float32)
    prob[10:20, 10:60] = 0.95          # kept
    prob[30:40, 10:60] = 0.45          # mean 0.45 < box_thresh 0.6 -> dropped
    prob[50:53, 10:13] = 0.95          # 3x3 pixels: contour area 4 < min_area 10 -> dropped
    prob[60, 100] = 0.95               # single pixel: 1 vertex -> dropped
    prob[70, 20:40] = 0.95             # 1-pixel-high line: 2 vertices after approximation -> dropped
    boxes, scores = _run_both(prob)
    assert len(boxes) == 1 and abs(scores[0] - 0.95) < 1e-6
    b = boxes[0]
    # perimeter 2*(49+9) = 116, area 441, d = 441*1.5/116 = 5.70
    assert (b[:, 0].min(), b[:, 0].max(), b[:, 1].min(), b[:, 1].max()) == (4, 65, 4, 25)


def test_order_and_holes():
    """cv2 returns the last-found border first; a ring yields an outer and a hole border, and the hole's polygon covers mostly background
    so its score falls below box_thresh."""
    prob = np.full((64, 96), 0.1, np.float32)
    prob[5:15, 5:40] = 0.9                                   # found first -> returned last
    prob[30:55, 20:80] = 0.9
    prob[38:47, 30:70] = 0.1                                 # hole
    contours = O.find_contours(prob > 0.3)
    assert len(contours) == 3
    assert contours[-1].tolist() == [[5, 5], [5, 14], [39, 14], [39, 5]]
    boxes, scores = _run_both(prob, box_thresh=0.5)
    assert len(boxes) == 2
    assert boxes[0][:, 1].mean() > 35 and boxes[1][:, 1].mean() < 15   # ring first, top bar second
    ring_area, hole_area = 25 * 60, 9 * 40
    assert abs(scores[0] - (0.9 * (ring_area - hole_area) + 0.1 * hole_area) / ring_area) < 0.02


def test_max_candidates_counts_contours_not_boxes():
    prob = np.full((40, 200), 0.1, np.float32)
    for i in range(8):
        prob[10:20, 5 + 24 * i:5 + 24 * i + 18] = 0.9
    prob[30, 3] = 0.9                                         # found last -> candidate 0, rejected, still counts
    boxes, _ = _run_both(prob, max_candidates=4)
    assert len(boxes) == 3


@pytest.mark.parametrize("seed,H,W,n", [(0, 96, 128, 10), (1, 160, 160, 25), (2, 64, 320, 18), (3, 224, 192, 40), (4, 33, 47, 6)])
def test_random_blobs_match_oracle(seed, H, W, n):
    prob = _blobs_map(seed, H, W, n)
    boxes, _ = _run_both(prob, box_thresh=0.5)
    assert len(boxes) > 0


@pytest.mark.parametrize("seed", range(6))
def test_noise_maps_match_oracle(seed):
    """Ragged contours: thresholded white noise (nested one-pixel holes and specks) and smoothed noise (curved blobs with concavities,
    which drive the Douglas-Peucker splits and Clipper's concave-vertex branch)."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(100 + seed)
    H, W = int(rng.integers(20, 110)), int(rng.integers(20, 150))
    prob = rng.uniform(0, 1, (H, W))
    if seed % 2:
        prob = ndi.gaussian_filter(prob, 1.0 + (seed % 3))
        prob = (prob - prob.min()) / (prob.max() - prob.min())
    boxes, _ = _run_both(prob.astype(np.float32), thresh=0.5, box_thresh=0.3, min_area=2)
    assert len(boxes) > 0


def test_border_touching_and_full_frame():
    prob = np.full((48, 64), 0.9, np.float32)                # the whole frame is text
    boxes, _ = _run_both(prob)
    assert len(boxes) == 1 and boxes[0][:, 0].min() < 0 and boxes[0][:, 0].max() > 63
    prob = np.full((48, 64), 0.1, np.float32)
    prob[0:9, 0:30] = 0.9
    prob[40:48, 50:64] = 0.9
    boxes, _ = _run_both(prob)
    assert len(boxes) == 2


def test_empty_map_and_bad_arguments():
    boxes, scores = P.DBPostProcessor()(np.zeros((1, 32, 32), np.float32))
    assert boxes == [] and scores == []
    with pytest.raises(ValueError):
        P.DBPostProcessor()(np.zeros((2, 3, 4, 5), np.float32))


def test_invoice_like_page_and_crop_rects():
    """A 640x480 map with ~30 text-line blobs; boxes rescaled as pipeline2.py:324-328 and turned into crop rectangles as crop_image does
    (src/det/test.py:123-130) agree between the product helpers and the oracle's."""
    rng = np.random.default_rng(7)
    prob = rng.uniform(0, 0.2, (640, 480)).astype(np.float32)
    y = 12
    for _ in range(30):
        x0, w = int(rng.integers(8, 120)), int(rng.integers(80, 330))
        h = int(rng.integers(9, 15))
        prob[y:y + h, x0:x0 + w] = rng.uniform(0.7, 0.99, (h, w))
        y += h + int(rng.integers(5, 9))
    boxes, scores = _run_both(prob)
    assert len(boxes) == 30
    scale_h, scale_w = 640 / 1123, 480 / 794
    mine = P.rescale_boxes(boxes, scale_w, scale_h)
    for b_in, b in zip(boxes, mine):
        ob, orect = O.rescale_and_rect(b_in, scale_w, scale_h, 1123, 794)
        assert np.array_equal(b, ob) and b.dtype == np.int32
        assert P.crop_rect((1123, 794), b) == orect
