"""CPU: host logic -- alphabet/tokenizer, weight folding + blob round trip, C-ABI library loads and exports every symbol
include/ocrvi.h declares (no compute calls: there is no GPU here)."""
import os
import re

import numpy as np
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
