"""GPU: the device half of DB post-processing (ocrvi_db_components: threshold -> 1-bit mask -> 8-connected components -> box / count /
probability sum -> packed box values) against the oracle (oracle/dbpost_cpu.py:components, scipy's labelling) -- integer work, so
bit-exact --, and the whole device-assisted stage (DBComponents.run -> to_host -> boxes) against the host-only stage on the same maps."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import dbpost_cpu as O  # noqa: E402
from ocr_vi_invoice_amd import pipeline as P  # noqa: E402
from test_dbpost_cpu import _blobs_map  # noqa: E402

pytestmark = pytest.mark.gpu


def _device_rows(dc, pg):
    n = int(dc.h_counts[pg])
    c = dc.h_comps[pg, :min(n, dc.cap)].numpy().astype(np.int64)
    rows = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), int((r[6] & 0xFFFFFFFF) | (r[7] << 32))) for r in c]
    order = sorted(range(len(rows)), key=lambda i: rows[i][5])
    return n, [rows[i] for i in order], order


def _check_against_oracle(prob, thresh, cap=4096):
    n, H, W = prob.shape
    dc = P.DBComponents(n, H, W, cap=cap, pack_frac=1.0)
    dprob = torch.from_numpy(prob).cuda()
    dc.run(dprob, thresh)
    dc.to_host()
    for pg in range(n):
        rows, bits = O.components(prob[pg], thresh)
        cnt, drows, order = _device_rows(dc, pg)
        assert cnt == len(rows)
        assert np.array_equal(dc.h_bits[pg].numpy().view(np.uint32), bits)
        assert drows == rows
        offs = dc.h_offsets[pg].numpy()
        fits = offs[min(cnt, dc.cap)] <= dc.pack_cap            # (a page whose boxes exceed the packed buffer is left unpacked by design)
        for k, i in enumerate(order if fits else []):          # packed values of every box = the map inside the box
            x0, y0, x1, y1 = drows[k][:4]
            box = dc.h_packed[pg, offs[i]:offs[i] + (x1 - x0 + 1) * (y1 - y0 + 1)].numpy().reshape(y1 - y0 + 1, x1 - x0 + 1)
            assert np.array_equal(box, prob[pg, y0:y1 + 1, x0:x1 + 1])
        assert offs[cnt] == sum((r[2] - r[0] + 1) * (r[3] - r[1] + 1) for r in rows)
    return dc


@pytest.mark.parametrize("H,W", [(96, 128), (64, 96), (33, 160), (120, 1248)])   # widths that are / are not multiples of 64 and of 256
def test_components_match_oracle_on_blob_maps(H, W):
    prob = np.stack([_blobs_map(7 + k, H, W, 14 + 3 * k) for k in range(3)])
    _check_against_oracle(prob, 0.3)


@pytest.mark.parametrize("seed", range(3))
def test_components_match_oracle_on_noise(seed):
    """Thresholded noise: thousands of ragged components with diagonal-only links, spirals and nested holes -- the union-find's worst case."""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    H, W = 128, 256
    prob = rng.uniform(0, 1, (2, H, W)).astype(np.float32)
    if seed:
        prob = np.stack([ndi.gaussian_filter(p, 0.8 * seed) for p in prob]).astype(np.float32)
        prob = (prob - prob.min()) / (prob.max() - prob.min())
    _check_against_oracle(prob, 0.5 if seed else 0.6, cap=8192)


def test_components_edge_cases():
    H, W = 40, 64
    prob = np.zeros((4, H, W), np.float32)
    prob[1] = 1.0                                              # one component covering the page
    prob[2, ::2, :] = 0.9                                      # stripes: H/2 components, each a full row crossing the 64-pixel segments
    prob[3, 0, 0] = prob[3, H - 1, W - 1] = prob[3, 0, W - 1] = 0.9   # corners
    yy, xx = np.mgrid[0:H, 0:W]
    prob[3][(yy == xx // 2 + 5)] = 0.8                          # a staircase: links through NE / NW neighbours only
    dc = _check_against_oracle(prob, 0.3)
    assert dc.h_counts.tolist()[:3] == [0, 1, H // 2]


def test_fullsize_page_and_device_assisted_stage_equals_host_stage():
    """960x1280 pages as the bench builds them (text kernels + low noise): the device-assisted stage returns exactly the rectangles and
    scores of the host-only stage, and moves a fraction of the bytes."""
    from ocr_vi_invoice_amd import synth
    H, W, n = 960, 1280, 3
    rng = np.random.default_rng(3)
    prob = (0.25 * rng.uniform(0, 1, (n, H, W))).astype(np.float32)
    for pg in range(n):
        _, boxes = synth.make_invoice(pg, H, W, 30)
        for (x, y, w, h) in boxes:
            prob[pg, y + 3:y + h - 3, x + 3:x + w - 3] += 0.7
    pp = P.DBPostProcessor(thresh=0.3, box_thresh=0.5, unclip_ratio=1.6)
    rf, cf, sf = P.db_boxes_batch(prob, pp, threads=4)
    dc = P.DBComponents(n, H, W)
    dprob = torch.from_numpy(prob).cuda()
    dc.run(dprob, pp.thresh)
    moved = dc.to_host()
    rs, cs, ss = dc.boxes(pp, dprob, threads=4)
    assert cs.tolist() == cf.tolist() and cs.sum() >= 60
    assert np.array_equal(rs, rf) and np.array_equal(ss, sf)
    assert moved < 0.5 * prob.nbytes
    for pg in range(n):                                        # and the table agrees with the oracle at full size
        rows, bits = O.components(prob[pg], pp.thresh)
        cnt, drows, _ = _device_rows(dc, pg)
        assert cnt == len(rows) and drows == rows and np.array_equal(dc.h_bits[pg].numpy().view(np.uint32), bits)


def test_overflow_falls_back_to_the_full_map():
    prob = np.stack([_blobs_map(21, 96, 128, 12), _blobs_map(22, 96, 128, 12)])
    pp = P.DBPostProcessor(box_thresh=0.5)
    rf, cf, sf = P.db_boxes_batch(prob, pp)
    dc = P.DBComponents(2, 96, 128, cap=2)                     # table far too small: every page overflows
    dprob = torch.from_numpy(prob).cuda()
    dc.run(dprob, pp.thresh)
    dc.to_host()
    assert int(dc.h_counts.min()) > 2
    rs, cs, ss = dc.boxes(pp, dprob)
    assert cs.tolist() == cf.tolist() and np.array_equal(rs, rf) and np.array_equal(ss, sf)
    with pytest.raises(RuntimeError):
        dc.boxes(pp, None)
