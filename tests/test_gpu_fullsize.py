"""GPU: size-independent properties at BASELINE.json's full sizes (det N=16 x 960x1280, rec B=256 x 48x320, bf16), where the CPU
oracle is too slow to run: determinism, per-sample independence (no cross-sample op exists in eval mode), agreement of the two
decode entry points, and agreement with the fp32 parity mode on a sub-batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", ["bf16", "f16"])   # f16 is BASELINE.json configs[4]'s dtype and bench.py's throughput mode
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


@pytest.mark.parametrize("dt", ["bf16", "f16"])
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
