"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own modules.

Run only in the build container (needs /root/reference; nothing here travels to the GPU box except the
.npz outputs):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference package cannot be imported normally (model/__init__.py pulls in torchvision, which is
absent), so the torch-only files are loaded by path under stub parent packages:
model/rec2/{vocab,tokenizer,svtrv2}.py and model/det/{layers,neck,head}.py.  The ResNet-50/DCN backbone
(model/det/backbone.py, dcn.py) needs torchvision and is therefore not covered by reference-run goldens.

Fixtures (inputs and reference outputs only -- no reference source):
  rec_tiny_32x256.npz, rec_base_48x320.npz : x, log_probs, stage taps, decoded ids/strings
  det_neckhead.npz                          : c2..c5 for a 64x96 image -> fused neck feature + 5 head maps
  ctc_kat.npz                               : id sequences -> reference decode_probs strings
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from ocr_vi_invoice_amd import synth, weights  # noqa: E402
from ocr_vi_invoice_amd.vocab import VOCAB  # noqa: E402


def _load_ref(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for pkg in ("model", "model.rec2", "model.det"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    _load_ref("model.rec2.vocab", "model/rec2/vocab.py")
    _load_ref("model.rec2.tokenizer", "model/rec2/tokenizer.py")
    sv = _load_ref("model.rec2.svtrv2", "model/rec2/svtrv2.py")
    _load_ref("model.det.layers", "model/det/layers.py")
    nk = _load_ref("model.det.neck", "model/det/neck.py")
    hd = _load_ref("model.det.head", "model/det/head.py")
    return sv, nk, hd


def rec_case(sv, variant, H, W, seed, out):
    model = sv.SVTRv2(variant=variant).eval()
    sd = weights.make_rec_state_dict(variant, seed=seed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("sgm.") for k in missing), [k for k in missing if not k.startswith("sgm.")]
    ref_keys = {k for k in model.state_dict() if not k.startswith("sgm.")}
    assert ref_keys == set(sd), ref_keys ^ set(sd)
    crops = synth.make_crops(seed + 1, 2, height=H, max_width=W)
    x = torch.from_numpy(synth.pad_crop_batch(crops, H, W))
    taps = {}
    with torch.no_grad():
        feats, fh, fw = model.extract_features(x)
        taps["backbone_norm"] = feats.numpy()
        taps["frm"] = model.frm(feats, fh, fw).numpy()
        lp = model(x)
        strings = model.decode_probs(lp)
    top2 = lp.topk(2, dim=-1).values
    margin = float((top2[..., 0] - top2[..., 1]).min())
    ids = lp.argmax(-1).permute(1, 0).numpy().astype(np.int32)
    print(f"{out}: log_probs {tuple(lp.shape)} range [{float(lp.min()):.3f},{float(lp.max()):.3f}] "
          f"min top-2 margin {margin:.4f} strings {strings}")
    np.savez_compressed(os.path.join(REPO, "tests/golden", out), x=x.numpy(), log_probs=lp.numpy(),
                        argmax_ids=ids, strings=np.array(strings), seed=seed, variant=variant, **taps)


def det_neckhead_case(nk, hd, seed, out):
    sd = weights.make_det_state_dict(seed=seed)
    neck = nk.FPN_ASF([256, 512, 1024, 2048], inner_channels=256).eval()
    head = hd.DBHead(256, k=50).eval()
    neck.load_state_dict({k[len("neck."):]: v for k, v in sd.items() if k.startswith("neck.")}, strict=True)
    head.load_state_dict({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}, strict=True)
    g = torch.Generator().manual_seed(seed + 5)
    H, W = 64, 96
    feats = [torch.relu(torch.randn(1, c, H // s, W // s, generator=g)) for c, s in
             ((256, 4), (512, 8), (1024, 16), (2048, 32))]
    with torch.no_grad():
        fused = neck(feats)
        maps = head(fused)
    print(f"{out}: fused {tuple(fused.shape)} binary range [{float(maps['binary'].min()):.3f},"
          f"{float(maps['binary'].max()):.3f}]")
    np.savez_compressed(os.path.join(REPO, "tests/golden", out), seed=seed,
                        c2=feats[0].numpy(), c3=feats[1].numpy(), c4=feats[2].numpy(), c5=feats[3].numpy(),
                        fused=fused.numpy().astype(np.float32), **{k: v.numpy() for k, v in maps.items()})


def ctc_kat(sv, out):
    """Known-answer decode cases through the reference's decode_probs (svtrv2.py:545-569)."""
    model = sv.SVTRv2(variant="tiny").eval()
    tok = model.tokenizer
    assert "".join(tok.charset) == VOCAB and tok.num_classes == 232
    a, b = tok.token_to_id["a"], tok.token_to_id["b"]
    seqs = [[a, a, 0, a, b, b], [a, 1, a, 0, 0, 0], [0, 0, 0, 0, 0, 0], [b, b, b, b, b, b],
            [231, 230, 2, 2, 0, 2], [1, 1, a, a, 1, b]]
    T, B = len(seqs[0]), len(seqs)
    lp = torch.full((T, B, 232), -10.0)
    for bi, s in enumerate(seqs):
        for t, c in enumerate(s):
            lp[t, bi, c] = -0.1
    strings = model.decode_probs(lp)
    flat = model.decode_probs(torch.zeros(4, 1, 232))  # all-equal -> argmax 0 -> ""
    assert flat == [""]
    print(f"{out}: {strings}")
    np.savez_compressed(os.path.join(REPO, "tests/golden", out), seqs=np.array(seqs, dtype=np.int32),
                        strings=np.array(strings))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sv, nk, hd = load_reference()
    rec_case(sv, "tiny", 32, 256, 7, "rec_tiny_32x256.npz")
    rec_case(sv, "base", 48, 320, 1234, "rec_base_48x320.npz")
    det_neckhead_case(nk, hd, 1234, "det_neckhead.npz")
    ctc_kat(sv, "ctc_kat.npz")
