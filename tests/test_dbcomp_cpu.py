"""Host finish of the device-assisted DB post-processing (ocrvi_db_boxes_batch_sparse) without a GPU: the device half's outputs are
emulated with the oracle (oracle/dbpost_cpu.py:components -- scipy's 8-connected labelling), and the results must equal the full-map
path's (ocrvi_db_boxes_batch) exactly: same rectangles, counts and scores.  The device kernels themselves are under test in
tests/test_gpu_dbcomp.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import dbpost_cpu as O  # noqa: E402
from ocr_vi_invoice_amd import _lib as L, pipeline as P  # noqa: E402
from test_dbpost_cpu import _blobs_map  # noqa: E402


def emulate_device(prob, thresh, cap, pack_cap):
    n, H, W = prob.shape
    bits = np.zeros((n, H, W // 32), np.uint32)
    comps = np.zeros((n, cap, 8), np.int32)
    counts = np.zeros(n, np.int32)
    offsets = np.zeros((n, cap + 1), np.int64)
    packed = np.full((n, pack_cap), np.nan, np.float32)
    for pg in range(n):
        rows, bits[pg] = O.components(prob[pg], thresh)
        rng = np.random.default_rng(pg)
        rng.shuffle(rows)                                   # the device hands ids out in arbitrary order
        counts[pg] = len(rows)
        off = 0
        for i, (x0, y0, x1, y1, cnt, root, sm) in enumerate(rows[:cap]):
            comps[pg, i] = (x0, y0, x1, y1, cnt, root, sm & 0xFFFFFFFF if (sm & 0xFFFFFFFF) < 2**31 else (sm & 0xFFFFFFFF) - 2**32, sm >> 32)
            offsets[pg, i] = off
            off += (x1 - x0 + 1) * (y1 - y0 + 1)
        offsets[pg, min(len(rows), cap)] = off
        if off <= pack_cap and len(rows) <= cap:
            o = 0
            for (x0, y0, x1, y1, *_r) in rows:
                box = prob[pg, y0:y1 + 1, x0:x1 + 1]
                packed[pg, o:o + box.size] = box.ravel()
                o += box.size
    return bits, comps, counts, offsets, packed


def run_sparse(prob, pp, cap=512, pack_frac=1.0, threads=3):
    n, H, W = prob.shape
    pack_cap = int(H * W * pack_frac)
    bits, comps, ccnt, offs, packed = emulate_device(prob, pp.thresh, cap, pack_cap)
    rects, scores = np.empty((n, 1000, 5), np.int32), np.empty((n, 1000), np.float32)
    counts, skipped = np.empty(n, np.int32), np.empty(n, np.int32)
    L.check(L.load().ocrvi_db_boxes_batch_sparse(bits.ctypes.data, comps.ctypes.data, ccnt.ctypes.data, cap, offs.ctypes.data, packed.ctypes.data,
                                                 pack_cap, n, H, W, pp.box_thresh, pp.max_candidates, pp.unclip_ratio, pp.min_area, 1.0, 1.0, H, W, 0,
                                                 rects.ctypes.data, scores.ctypes.data, 1000, counts.ctypes.data, threads, skipped.ctypes.data))
    return rects, scores, counts, skipped


@pytest.mark.parametrize("seed,H,W,n", [(0, 96, 128, 10), (1, 160, 160, 25), (2, 64, 320, 18), (3, 224, 192, 40)])
def test_sparse_equals_full_map(seed, H, W, n):
    prob = np.stack([_blobs_map(seed * 10 + k, H, W, n) for k in range(3)])
    pp = P.DBPostProcessor(box_thresh=0.5, unclip_ratio=1.6)
    rf, cf, sf = P.db_boxes_batch(prob, pp, threads=2)
    rects, scores, counts, skipped = run_sparse(prob, pp)
    assert not skipped.any() and counts.tolist() == cf.tolist() and counts.sum() > 0
    assert np.array_equal(np.concatenate([rects[i, :counts[i]] for i in range(3)]), rf)
    assert np.array_equal(np.concatenate([scores[i, :counts[i]] for i in range(3)]), sf)          # bit-equal: same values, same code


def test_overflowing_pages_are_skipped_not_guessed():
    prob = np.stack([_blobs_map(5, 96, 128, 12), _blobs_map(6, 96, 128, 12)])
    pp = P.DBPostProcessor(box_thresh=0.5)
    _, _, counts, skipped = run_sparse(prob, pp, cap=2)                    # table too small
    assert skipped.tolist() == [1, 1] and counts.tolist() == [0, 0]
    _, _, counts, skipped = run_sparse(prob, pp, pack_frac=0.01)           # packed buffer too small
    assert skipped.tolist() == [1, 1]


def test_oracle_components_known_answer():
    prob = np.zeros((8, 32), np.float32)
    prob[1, 2:5] = 0.9          # bar
    prob[2, 5] = 0.5            # touches the bar diagonally -> same component (8-connectivity)
    prob[6, 30] = 0.7           # lone pixel
    rows, bits = O.components(prob, 0.3)
    assert [r[:6] for r in rows] == [(2, 1, 5, 2, 4, 1 * 32 + 2), (30, 6, 30, 6, 1, 6 * 32 + 30)]
    assert rows[0][6] == 3 * round(0.9 * 2 ** 20) + round(0.5 * 2 ** 20)
    assert bits[1, 0] == 0b11100 and bits[2, 0] == 1 << 5 and bits[6, 0] == 1 << 30
