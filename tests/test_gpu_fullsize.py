"""GPU: BASELINE.json's full sizes (det N=16 x 960x1280, rec B=256 x 48x320).  Where the CPU oracle is too slow to run the whole
batch: determinism, per-sample independence (no cross-sample op exists in eval mode), agreement of the two decode entry points and
agreement of the 16-bit modes with the parity mode; and ONE full-size page of the parity mode against the CPU oracle itself
(probability maps, backbone / neck taps, crop rectangles), so that the kernel instantiations only full-size shapes select are compared
with the oracle and not only with themselves."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARITY_DTYPES = ["f32", "f16x2"]      # modes that must reproduce the CPU oracle (1e-3 maps / log-probs, identical rectangles and strings)


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_mod_fullsize", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("dt", PARITY_DTYPES)
def test_det_fullsize_parity_mode_matches_cpu_oracle(dt):
    """dbnet.py:13-17 at the size it is benchmarked at: one 960x1280 page (M = 76800 rows in layer 1, 1200 in layer 4: the tiles
    launch_mode / launch_gemm_ring / launch_offs_conv pick by shape differ from every reduced-size test) against oracle.dbnet_cpu:
    three probability maps <= 1e-3, c2..c5 + fused taps, and the crop rectangles DB post-processing derives from the (blended, as in
    bench.py) binary maps equal."""
    from ocr_vi_invoice_amd import DBNetPP, synth, weights
    from ocr_vi_invoice_amd.pipeline import DBPostProcessor, db_boxes_batch
    from oracle import dbnet_cpu
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    bench = _bench_module()
    H, W = 960, 1280
    sd = weights.make_det_state_dict(seed=1234)
    img, boxes = synth.make_invoice(3, H, W, 30)
    x = torch.from_numpy(synth.normalize_chw(img))[None]
    ref = dbnet_cpu.forward(sd, x, return_feats=True)
    m = DBNetPP(pretrained=False, state_dict=sd, dtype=dt)
    feats = m.debug_features(x.cuda())
    for k in ("c2", "c3", "c4", "c5", "fused"):
        r = ref[k]
        scale = max(float(r.abs().max()), 1.0)
        err = float((feats[k].cpu() - r).abs().max())
        assert err < 2e-4 * scale, (k, err, scale)
    out = m(x.cuda())
    for k in ("binary", "thresh", "thresh_binary"):
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert err < 1e-3, (k, err)                                   # north_star: probability maps within 1e-3
    add = np.zeros((H, W), np.float32)
    for bx, by, bw, bh in boxes:
        sx, sy, sw, sh = bench.shrink_box(int(bx), int(by), int(bw), int(bh))
        add[sy:sy + sh, sx:sx + sw] = 0.75
    pp = DBPostProcessor(0.3, 0.5, 1000, 1.6)
    ma = (add + np.float32(0.25) * out["binary"][0, 0].cpu().numpy()).astype(np.float32)
    mb = (add + np.float32(0.25) * ref["binary"][0, 0].numpy()).astype(np.float32)
    ra, ca, _ = db_boxes_batch(ma[None], pp)
    rb, cb, _ = db_boxes_batch(mb[None], pp)
    assert int(ca[0]) >= 25 and np.array_equal(ra, rb)               # north_star: polygon boxes identical
    # the raw (structureless, random-weight) map too: whatever components it has must come out the same
    r2a, _, _ = db_boxes_batch(out["binary"][0].cpu().numpy(), pp)
    r2b, _, _ = db_boxes_batch(ref["binary"][0].numpy(), pp)
    assert np.array_equal(r2a, r2b)


# budgets = 1.5 x the binary-map errors measured on MI355X in round 2 at full size (profiles/r02_precision_study.json: f16 0.0115, bf16 0.084)
# f16x2 -- the mode bench.py quotes its headline in -- is held to 1e-4 here (measured 1.9e-5 against the CPU oracle on single pages): at
# N = 16 its layers take other tiles than at N = 1 (ring grids at M = 1 228 800, conv_gemm<f16x2, 128, 128> at 9600 tiles, dcn_pipe over 16 images)
@pytest.mark.parametrize("dt,tol", [("f16x2", 1e-4), ("f16", 0.0173), ("bf16", 0.126)])
def test_det_fullsize_lowp_tracks_parity_mode(dt, tol):
    """16 pages 960x1280 (config 2's batch, the bench's det chunk): the detector in every other mode against the exact-fp32 mode on the
    GPU -- a wrong dcn_pipe patch shape, a wrong second column tile at N = 512 or a wrong ring tile at real M would show here,
    determinism alone would not."""
    from ocr_vi_invoice_amd import DBNetPP, synth
    imgs = np.stack([synth.normalize_chw(synth.make_invoice(s, 960, 1280, 30)[0]) for s in range(16)])
    x = torch.from_numpy(imgs).cuda()
    ref = DBNetPP(pretrained=False, dtype="f32", seed=1234)
    rb = torch.cat([ref(x[i:i + 4])["binary"] for i in range(0, 16, 4)])
    del ref
    out = DBNetPP(pretrained=False, dtype=dt, seed=1234)(x)["binary"]
    err = float((out - rb).abs().max())
    mean = float((out - rb).abs().mean())
    print(f"\n[{dt}] 16 full-size pages vs f32 mode: binary max-abs-err {err:.4f} mean {mean:.5f}")
    assert err < tol, err


# f16 is BASELINE.json configs[4]'s dtype and bench.py's throughput mode; f16x2 its headline mode; f32 the facades' default.  Batch
# independence is bit-exact in every mode: the tiles a layer takes depend on M (256- vs 128-row ring builds, 32- vs 64-row DCN tiles) but
# only regroup rows -- every output element sees the same products in the same order (and the 4-byte ring builds start every tile's
# accumulators at the bias whatever the tile height).
@pytest.mark.parametrize("dt", ["bf16", "f16", "f16x2", "f32"])
def test_det_fullsize_determinism_and_independence(dt):
    from ocr_vi_invoice_amd import DBNetPP, synth
    m = DBNetPP(pretrained=False, dtype=dt, seed=1234)
    imgs = np.stack([synth.normalize_chw(synth.make_invoice(s, 960, 1280, 30)[0]) for s in range(4)])
    x = torch.from_numpy(imgs).cuda().repeat(4, 1, 1, 1)              # 16 pages: 4 distinct, each repeated 4x
    a = m(x)
    b = m(x)
    for k in a:
        assert a[k].shape == (16, 1, 960, 1280)
        assert torch.equal(a[k], b[k]), f"{k}: two runs differ"       # deterministic (no atomics on the path)
        assert torch.isfinite(a[k]).all()
    assert float(a["binary"].min()) >= 0 and float(a["binary"].max()) <= 1
    for i in range(4):                                                # copies of a page anywhere in the batch agree bit for bit
        for r in range(1, 4):
            assert torch.equal(a["binary"][i], a["binary"][i + 4 * r])
    single = m(x[2:3])                                                # and a page alone == the same page inside the batch
    assert torch.equal(single["binary"][0], a["binary"][2])
    # thresh_binary is the step function of the other two maps (head.py:28-30)
    tb = torch.reciprocal(1 + torch.exp(-50.0 * (a["binary"] - a["thresh"])))
    np.testing.assert_allclose(a["thresh_binary"].cpu().numpy(), tb.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(a["binary"].cpu().numpy(), torch.sigmoid(a["bin_logits"]).cpu().numpy(), atol=2e-6)


@pytest.mark.parametrize("dt", ["bf16", "f16", "f16x2", "f32"])
def test_rec_fullsize_determinism_independence_and_decode_paths(dt):
    from ocr_vi_invoice_amd import SVTRv2, synth
    m = SVTRv2("base", dtype=dt, seed=1234)
    x = torch.from_numpy(synth.pad_crop_batch(synth.make_crops(7, 256, 48, 320), 48, 320)).cuda()
    lp = m(x)
    assert lp.shape == (80, 256, 232) and float(lp.max()) <= 0
    assert torch.equal(lp, m(x))
    np.testing.assert_allclose(torch.logsumexp(lp, -1).cpu().numpy(), 0.0, atol=1e-4)     # rows are log-probabilities
    sub = m(x[100:132])                                                                  # a sub-batch reproduces its rows exactly
    assert torch.equal(sub, lp[:, 100:132])
    t1 = m.decode_probs(lp)            # argmax + collapse of given log-probs on the device
    t2 = m.decode_greedy(x)            # fused forward + decode
    assert t1 == t2 and len(t1) == 256
    # host-side greedy decode of the same log-probs (svtrv2.py:555-566 restated inline) agrees with the device decoder
    ids = lp.argmax(-1).T.cpu().tolist()
    host = []
    for seq in ids:
        keep, prev = [], None
        for p in seq:
            if p != 0 and p != prev:
                keep.append(p)
            prev = p
        host.append(keep)
    assert m.tokenizer.decode(host) == t1


def test_rec_bf16_tracks_fp32_parity_mode_on_a_subbatch():
    from ocr_vi_invoice_amd import SVTRv2, synth
    x = torch.from_numpy(synth.pad_crop_batch(synth.make_crops(9, 32, 48, 320), 48, 320)).cuda()
    ref = SVTRv2("base", dtype="f32", seed=1234)(x)
    # budgets = 1.5 x the values measured on MI355X in round 2 (bf16 0.1409 / 0.9969, f16 0.0211 / 0.9996)
    for dt, tol, min_agree in (("bf16", 0.21, 0.993), ("f16", 0.032, 0.999)):
        lp = SVTRv2("base", dtype=dt, seed=1234)(x)
        err = float((lp - ref).abs().max())
        agree = float((lp.argmax(-1) == ref.argmax(-1)).float().mean())
        print(f"\n[{dt}] vs f32: max-abs-err {err:.4f}, argmax agreement {agree:.4f}")
        assert err < tol and agree > min_agree
