"""GPU: the exponent range of the f16x2 mode (include/ocrvi.h, csrc/common.h).

The reference is fp32 end to end (src/pipeline/pipeline2.py:312-318: 8 exponent bits); f16x2 keeps every GEMM operand as two fp16
halves, so it is only as wide as fp16's exponent.  Pinned here:

* the error law at the kernel level: activations scaled by 2^-12 .. 2^+12 through the ring GEMM and a 3x3 convolution against an fp64
  product of the same fp32 operands.  While |x| >= 2^-3 an element keeps >= 22 bits (relative error <= 2^-23); below that its lo half
  is a subnormal fp16 number and the error is absolute, 2^-25 per element, i.e. 2^-25 / rms(x) relative to the output.  Weights carry
  their own power-of-two scale per layer (ConvParams::wscale), so scaling them changes nothing;
* the supported range that follows from it: rms(x) in [2^-8, 2^+12] keeps the fp32 modes' 2e-5 budget (4e-5 at the lower edge);
* the upper edge is LOUD: a value with |x| >= 65520 raises the device's range flag wherever it is packed (input cast, GEMM epilogue,
  LayerNorm output), *_forward hands the flag to the handle without synchronising, ocrvi_{det,rec}_status return OCRVI_ERANGE and the
  facades raise OverflowError at the point where they synchronise anyway;
* the same at the model level: one layer of SVTRv2 driven to 2^-10 .. 2^+10 (LayerNorm affine scaled, the following Linear scaled back)
  stays within 1e-3 of the exact-fp32 mode's log-probs; driven past 65520 it raises.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}


def _L():
    from ocr_vi_invoice_amd import _lib as L
    return L


def _flag():
    L = _L()
    v = C.c_int(0)
    L.check(L.load().ocrvi_range_flag(0, C.byref(v)))
    return v.value


def _reset():
    L = _L()
    L.check(L.load().ocrvi_range_reset(0, None))
    torch.cuda.synchronize()


def _gemm(a, w, b, dt, out_f32=1):
    L = _L()
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), device="cuda")
    ms = C.c_float(0)
    ad = a.cuda()
    wh, bh = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(b.numpy())
    L.check(L.load().ocrvi_test_gemm(0, DT[dt], ad.data_ptr(), wh.ctypes.data, bh.ctypes.data, None, M, K, N, 0, 0, out_f32, out.data_ptr(), 0,
                                     C.byref(ms)))
    return out.cpu()


def _law(scale_log2):
    """max |error| / rms(output) allowed for activations of rms 2^scale_log2: the fp32 modes' budget while every element keeps its
    22 bits, 5 x (2^-25 / rms(x)) -- five standard deviations of a sum of uniform +-2^-25 element errors -- below that."""
    s = 2.0 ** scale_log2
    return max(2e-5, 5.0 * 2.0 ** -25 / s)


@pytest.mark.parametrize("e", [-12, -10, -8, -4, 0, 4, 8, 12])
def test_ring_gemm_error_law_over_the_activation_range(e):
    _reset()
    g = torch.Generator().manual_seed(100 + e)
    M, K, N = 9000, 384, 384
    a = torch.randn(M, K, generator=g) * 2.0 ** e
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.zeros(N)
    ref = (a.double() @ w.double().t())
    rms = float(ref.pow(2).mean().sqrt())
    got = _gemm(a, w, b, "f16x2")
    err = float((got.double() - ref).abs().max()) / rms
    exact = float((_gemm(a, w, b, "f32").double() - ref).abs().max()) / rms
    print(f"\n[2^{e}] f16x2 max err / rms {err:.2e} (law {_law(e):.2e}); exact-fp32 mode {exact:.2e}")
    assert err < _law(e), (e, err)
    if -8 <= e:
        assert err < 4e-5          # the supported range: the fp32 modes' budget (2e-5) up to the 2^-8 edge's factor of two
    assert exact < 2e-5            # (the exact mode does not care)
    assert _flag() == 0            # nothing left fp16's range: max |a| = 5 sigma * 2^12 < 65520
    # a weight scale changes nothing: the packer normalises every layer by a power of two
    got_w = _gemm(a, w * 2.0 ** 9, b, "f16x2")
    assert float((got_w.double() / 2.0 ** 9 - ref).abs().max()) / rms < _law(e)


@pytest.mark.parametrize("e", [-10, 0, 10])
def test_conv3x3_error_law_over_the_activation_range(e):
    L = _L()
    _reset()
    g = torch.Generator().manual_seed(7 + e)
    x = torch.randn(2, 128, 24, 40, generator=g) * 2.0 ** e
    w = torch.randn(128, 128, 3, 3, generator=g) / np.sqrt(9 * 128)
    b = torch.zeros(128)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    out = torch.empty(ref.shape, device="cuda")
    wh, bh = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(b.numpy())
    ms = C.c_float(0)
    xd = x.cuda()
    L.check(L.load().ocrvi_test_conv(0, DT["f16x2"], xd.data_ptr(), wh.ctypes.data, bh.ctypes.data, 2, 128, 24, 40, 128, 3, 1, 1, 1, 0, out.data_ptr(), 0,
                                     C.byref(ms)))
    err = float((out.cpu().double() - ref).abs().max() / ref.pow(2).mean().sqrt())
    print(f"\n[2^{e}] conv3x3 f16x2 max err / rms {err:.2e} (law {_law(e):.2e})")
    assert err < _law(e) and _flag() == 0


def test_out_of_range_input_and_output_raise_the_flag():
    g = torch.Generator().manual_seed(3)
    M, K, N = 4000, 256, 256
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.zeros(N)
    # (1) an input element fp16 cannot carry: caught where it is packed (the cast in front of the GEMM)
    _reset()
    a = torch.randn(M, K, generator=g)
    a[1234, 17] = 70000.0
    _gemm(a, w, b, "f16x2")
    assert _flag() == 1
    _reset()
    assert _flag() == 0
    # (2) inputs in range, OUTPUT out of range: the ring GEMM's epilogue packs f16x2 when the output is not fp32
    a = torch.randn(M, K, generator=g) * 2.0 ** 10
    out = _gemm(a, w * 2.0 ** 7, b, "f16x2", out_f32=0)          # output rms 2^17
    assert _flag() == 1
    assert not torch.isfinite(out).all()                          # what the flag is there to announce
    _reset()
    # (3) the same product with an fp32 output is representable and raises nothing; nor does anything in the other modes
    ok = _gemm(a, w * 2.0 ** 7, b, "f16x2", out_f32=1)
    assert torch.isfinite(ok).all() and _flag() == 0
    _gemm(a, w * 2.0 ** 7, b, "f32")
    assert _flag() == 0
    # (4) the largest representable magnitude passes: 65504 = fp16 max, 65519.9 still rounds to it
    a = torch.randn(M, K, generator=g)
    a[7, 7], a[8, 8] = 65504.0, -65519.0
    _gemm(a, w, b, "f16x2")
    assert _flag() == 0


def _scaled_block_sd(sd, s):
    """SVTRv2 state_dict with the MLP input of stage 0 / block 0 driven to rms ~ s: norm2's affine (svtrv2.py:95,100) times s, fc1's
    weight (svtrv2.py:30) divided by s -- the same function in exact arithmetic."""
    out = {k: v.clone() for k, v in sd.items()}
    out["stages.0.blocks.0.norm2.weight"] *= s
    out["stages.0.blocks.0.norm2.bias"] *= s
    out["stages.0.blocks.0.mlp.fc1.weight"] /= s
    return out


@pytest.mark.parametrize("e", [-10, -6, 6, 10])
def test_model_level_activation_scale_keeps_the_parity_bar(e):
    from ocr_vi_invoice_amd import SVTRv2, synth, weights
    sd = weights.make_rec_state_dict("tiny", seed=3)
    x = torch.from_numpy(synth.pad_crop_batch(synth.make_crops(5, 4, height=32, max_width=128), 32, 128)).cuda()
    base = SVTRv2("tiny", state_dict=sd, dtype="f32")(x)
    sds = _scaled_block_sd(sd, 2.0 ** e)
    ref = SVTRv2("tiny", state_dict=sds, dtype="f32")(x)
    assert float((ref - base).abs().max()) < 1e-3                                    # the rescaled model IS the same function
    m = SVTRv2("tiny", state_dict=sds, dtype="f16x2")
    m.reset_range()
    lp = m(x)
    m.check_range()                                                                  # in range: no error
    err = float((lp - ref).abs().max())
    print(f"\n[2^{e}] f16x2 vs f32 mode log-probs: max |d| {err:.2e}")
    assert err < 1e-3 and m.decode_probs(lp) == m.decode_probs(ref)


def test_model_level_overflow_is_reported_not_swallowed():
    from ocr_vi_invoice_amd import DBNetPP, SVTRv2, synth, weights
    sd = weights.make_rec_state_dict("tiny", seed=3)
    x = torch.from_numpy(synth.pad_crop_batch(synth.make_crops(5, 4, height=32, max_width=128), 32, 128)).cuda()
    bad = SVTRv2("tiny", state_dict=_scaled_block_sd(sd, 2.0 ** 17), dtype="f16x2")    # LayerNorm output ~ 2^17 > 65504
    bad.reset_range()
    lp = bad(x)                                                                      # forward itself never synchronises or raises
    with pytest.raises(OverflowError, match="65520"):
        bad.check_range()
    with pytest.raises(OverflowError):
        bad.decode_greedy(x)                                                         # the decode path checks where it waits for the ids
    # the exact-fp32 mode runs the same weights without complaint, and a healthy f16x2 model is clean again after the reset
    SVTRv2("tiny", state_dict=_scaled_block_sd(sd, 2.0 ** 17), dtype="f32")(x)
    good = SVTRv2("tiny", state_dict=sd, dtype="f16x2")
    good.reset_range()
    assert len(good.decode_greedy(x)) == 4
    good.check_range()
    del lp
    # detector: an input pixel value past fp16's range (a broken normalisation upstream) is caught by the first layout kernel
    det = DBNetPP(pretrained=False, dtype="f16x2", seed=3)
    det.reset_range()
    xi = torch.zeros(1, 3, 64, 96, device="cuda")
    det(xi)
    det.check_range()
    xi[0, 1, 5, 5] = 1e5
    det(xi)
    with pytest.raises(OverflowError):
        det.check_range()
    det.reset_range()
