"""GPU parity of the device pre-processing kernels (integer resize: bit-exact; float normalisation: 1 ulp)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _L():
    from ocr_vi_invoice_amd import _lib
    return _lib, _lib.load()


def test_normalize_u8_matches_oracle():
    from oracle import preproc_cpu as P
    L, lib = _L()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(2, 64, 96, 3), dtype=np.uint8)
    d = torch.from_numpy(img).cuda()
    out = torch.empty((2, 3, 64, 96), device="cuda")
    L.check(lib.ocrvi_normalize_u8(0, d.data_ptr(), 2, 64, 96, out.data_ptr(), None))
    ref = np.stack([P.normalize_det(im) for im in img])
    np.testing.assert_array_equal(out.cpu().numpy(), ref)     # same float32/float64 operation order -> identical bits


@pytest.mark.parametrize("oh,ow", [(48, 320), (32, 256)])
def test_crop_resize_normalize_matches_oracle(oh, ow):
    from ocr_vi_invoice_amd import synth
    from oracle import preproc_cpu as P
    L, lib = _L()
    imgs = np.stack([synth.make_invoice(s, 192, 400, lines=5)[0] for s in (3, 4)])
    rng = np.random.default_rng(1)
    boxes = []
    for i in range(40):
        w, h = int(rng.integers(1, 399)), int(rng.integers(1, 100))
        x, y = int(rng.integers(0, 400 - w)), int(rng.integers(0, 192 - h))
        boxes.append((i % 2, x, y, w, h))
    boxes += [(0, 10, 20, 2 * 100, 2 * oh),     # exact 2x decimation -> area path
              (1, 0, 0, 400, 192),              # whole image, squashed (new_w > target)
              (0, 5, 5, 1, 1), (1, 7, 9, 3, oh), (0, 0, 0, 0, 10), (1, 3, 3, 10, 0)]   # tiny / empty crops
    b = np.asarray(boxes, np.int32)
    out = torch.empty((len(b), 3, oh, ow), device="cuda")
    di, db = torch.from_numpy(imgs).cuda(), torch.from_numpy(b).cuda()
    L.check(lib.ocrvi_crop_resize_normalize(0, di.data_ptr(), 2, 192, 400, db.data_ptr(), len(b), oh, ow, out.data_ptr(), None))
    got = out.cpu().numpy()
    for k, (i, x, y, w, h) in enumerate(boxes):
        ref = P.preprocess_for_recognition(imgs[i][y:y + h, x:x + w], (oh, ow))
        # integer pixel values are bit-exact; the float32 normalisation may differ by 1 ulp between numpy and the device
        np.testing.assert_allclose(got[k], ref, rtol=0, atol=3e-7, err_msg=f"box {k} {boxes[k]}")
