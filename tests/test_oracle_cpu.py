"""CPU: the oracle (oracle/*.py) against the reference-generated goldens and its self-checks."""
import os

import numpy as np
import pytest
import torch

from ocr_vi_invoice_amd import weights
from ocr_vi_invoice_amd.vocab import Tokenizer
from oracle import dbnet_cpu, svtrv2_cpu

torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.mark.parametrize("name", ["rec_tiny_32x256", "rec_base_48x320"])
def test_svtrv2_oracle_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    variant = str(g["variant"])
    sd = weights.make_rec_state_dict(variant, seed=int(g["seed"]))
    lp, taps = svtrv2_cpu.forward(sd, torch.from_numpy(g["x"]), variant, return_taps=True)
    assert lp.shape == g["log_probs"].shape  # (T=W/4, B, 232)  tests/test_model.py:271-277
    assert float(lp.max()) <= 0.0
    np.testing.assert_allclose(taps["backbone_norm"].numpy(), g["backbone_norm"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(taps["frm"].numpy(), g["frm"], atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(lp.numpy(), g["log_probs"], atol=1e-4, rtol=1e-5)
    ids = svtrv2_cpu.greedy_ids(lp)
    assert Tokenizer().decode(ids) == [str(s) for s in g["strings"]]


def test_ctc_decode_kat(golden_dir):
    g = np.load(os.path.join(golden_dir, "ctc_kat.npz"))
    seqs = g["seqs"]
    B, T = seqs.shape
    lp = torch.full((T, B, 232), -10.0)
    for b in range(B):
        for t in range(T):
            lp[t, b, int(seqs[b, t])] = -0.1
    assert Tokenizer().decode(svtrv2_cpu.greedy_ids(lp)) == [str(s) for s in g["strings"]]
    # all-equal logits -> argmax picks index 0 (blank) -> empty string
    assert Tokenizer().decode(svtrv2_cpu.greedy_ids(torch.zeros(4, 1, 232))) == [""]


def test_neck_head_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "det_neckhead.npz"))
    sd = weights.make_det_state_dict(seed=int(g["seed"]))
    feats = [torch.from_numpy(g[k]) for k in ("c2", "c3", "c4", "c5")]
    fused = dbnet_cpu.neck(sd, feats)
    np.testing.assert_allclose(fused.numpy(), g["fused"], atol=1e-4, rtol=1e-5)
    maps = dbnet_cpu.head(sd, fused)
    for k in ("binary", "thresh", "thresh_binary"):
        assert maps[k].shape == (1, 1, 64, 96)
        np.testing.assert_allclose(maps[k].numpy(), g[k], atol=1e-5)
    for k in ("bin_logits", "thresh_logits"):
        np.testing.assert_allclose(maps[k].numpy(), g[k], atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("stride", [1, 2])
def test_dcn_two_formulations_agree(stride):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 14, 18, generator=g)
    Ho, Wo = (14 - 1) // stride + 1, (18 - 1) // stride + 1
    off = torch.randn(2, 18, Ho, Wo, generator=g) * 3.0
    mask = torch.rand(2, 9, Ho, Wo, generator=g)
    w = torch.randn(8, 16, 3, 3, generator=g) * 0.1
    a = dbnet_cpu.deform_conv2d_gather(x, off, mask, w, stride)
    b = dbnet_cpu.deform_conv2d_gridsample(x, off, mask, w, stride)
    assert a.shape == (2, 8, Ho, Wo)
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-5)


@pytest.mark.parametrize("stride", [1, 2])
def test_dcn_zero_offset_identity(stride):
    """Reference init (dcn.py:28-29): offsets 0, mask sigmoid(0)=0.5 -> 0.5 * conv2d."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 8, 12, 12, generator=g)
    w = torch.randn(4, 8, 3, 3, generator=g)
    Ho = (12 - 1) // stride + 1
    out = dbnet_cpu.deform_conv2d_gather(x, torch.zeros(1, 18, Ho, Ho), torch.full((1, 9, Ho, Ho), 0.5), w, stride)
    ref = 0.5 * torch.nn.functional.conv2d(x, w, None, stride, 1)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-5)


def test_dcn_integer_shift():
    """Integer offsets == conv over a shifted (zero-padded) input."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 10, 10, generator=g)
    w = torch.randn(3, 4, 3, 3, generator=g)
    off = torch.zeros(1, 18, 10, 10)
    off[:, 0::2] = 1.0   # dy = +1
    off[:, 1::2] = -2.0  # dx = -2
    out = dbnet_cpu.deform_conv2d_gather(x, off, torch.ones(1, 9, 10, 10), w, 1)
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 2:] = x[:, :, 1:, :-2]
    ref = torch.nn.functional.conv2d(xs, w, None, 1, 1)
    # borders differ by construction (the shifted copy zero-pads where DCN still samples in range)
    np.testing.assert_allclose(out.numpy()[..., 2:-2, 3:-3], ref.numpy()[..., 2:-2, 3:-3], atol=1e-5)


def test_dbnet_shapes_and_param_count():
    """tests/test_model.py:102-143,162-165 shape/range KATs + SURVEY 8c param-count KAT."""
    sd = weights.make_det_state_dict(seed=1)
    n = sum(v.numel() for k, v in sd.items() if not k.endswith("num_batches_tracked")
            and not any(s in k for s in ("running_mean", "running_var")))
    # 30 106 637 total minus torchvision's unused fc (2048*1000+1000), which the generator does not emit
    assert n == 30_106_637 - 2_049_000
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(0))
    out = dbnet_cpu.forward(sd, x, return_feats=True)
    assert [out[k].shape[1] for k in ("c2", "c3", "c4", "c5")] == [256, 512, 1024, 2048]
    assert out["fused"].shape == (1, 256, 16, 24)
    for k in ("binary", "thresh", "thresh_binary"):
        assert out[k].shape == (1, 1, 64, 96)
        assert 0.0 <= float(out[k].min()) and float(out[k].max()) <= 1.0


def test_preproc_oracle_known_answers():
    """Hand-checkable cases of the cv2-style 8-bit bilinear resize restatement (oracle/preproc_cpu.py)."""
    from oracle import preproc_cpu as P
    flat = np.full((7, 13, 3), 200, np.uint8)
    assert (P.resize_linear_u8(flat, (29, 48)) == 200).all()            # constant image stays constant
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    np.testing.assert_array_equal(P.resize_linear_u8(img, (6, 4)), img)  # identity scale
    two = P.resize_linear_u8(img, (3, 2))                                # exact 2x decimation = 2x2 box mean, round half up
    want = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    np.testing.assert_array_equal(two, want)
    up = P.resize_linear_u8(np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8), (4, 1))[0, :, 0]
    assert list(up) == [0, 64, 191, 255]                                 # taps at -0.25, 0.25, 0.75, 1.25 -> weights .25/.75
    crop = np.full((10, 40, 3), 255, np.uint8)
    t = P.preprocess_for_recognition(crop, (32, 256))                    # new_w = int(40 * 3.2) = 128, right-padded with 255
    assert t.shape == (3, 32, 256)
    np.testing.assert_allclose(t[0], (1.0 - 0.485) / 0.229, rtol=1e-6)
    assert (P.preprocess_for_recognition(np.zeros((0, 5, 3), np.uint8)) == 0).all()
