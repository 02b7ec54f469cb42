"""Weights: reference state_dict schema -> BN-folded tensors -> one flat blob for libocrvi.

* ``make_rec_state_dict`` / ``make_det_state_dict`` build *seeded synthetic* state_dicts keyed
  exactly like the reference modules' ``state_dict()`` (SURVEY.md Appendix A;
  model/rec2/svtrv2.py:410-470, model/det/{backbone,dcn,neck,head}.py).  No trained checkpoint
  ships with the reference, so these are what parity fixtures and the benchmark run on.  BN
  running stats / affines are deliberately non-trivial so folding bugs show.
* ``fold_rec`` / ``fold_det`` accept such a state_dict (or a real checkpoint in the same schema,
  dict-wrapped or ``module.``-prefixed as pipeline2.py:46-52,75-80 tolerates), fold eval-mode
  BatchNorm into the preceding conv, precompute input-independent terms, and return an ordered
  ``{name: float32 ndarray}`` in standard OIHW / (out,in) layouts.
* ``pack_blob`` serialises that dict to the byte layout ``ocrvi_*_create`` parses
  (include/ocrvi.h).  Kernel-specific layouts and the compute dtype are chosen on the C side.
"""
from __future__ import annotations

import math
import struct
from collections import OrderedDict
from typing import Dict, Mapping, Tuple

import numpy as np
import torch

BN_EPS = 1e-5
LN_EPS = 1e-5

# model/rec2/svtrv2.py:391-407
REC_VARIANTS = {
    "tiny": dict(dims=[64, 128, 256], num_blocks=[3, 6, 3], num_local=[3, 3, 0]),
    "small": dict(dims=[96, 192, 256], num_blocks=[3, 6, 6], num_local=[3, 3, 0]),
    "base": dict(dims=[128, 256, 384], num_blocks=[3, 6, 6], num_local=[3, 2, 0]),
}
NUM_CLASSES = 232  # 230 chars + blank + pad (tokenizer.py:21)

# torchvision resnet50: Bottleneck counts and widths per layer
R50_BLOCKS = [3, 4, 6, 3]
R50_WIDTH = [64, 128, 256, 512]


# ----------------------------------------------------------------------------- generators
class _Gen:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)

    def normal(self, shape, std=1.0, mean=0.0):
        return torch.randn(shape, generator=self.g) * std + mean

    def uniform(self, shape, lo, hi):
        return torch.rand(shape, generator=self.g) * (hi - lo) + lo


def _bn(sd, g: _Gen, prefix, c, gamma=(0.5, 1.5)):
    sd[prefix + ".weight"] = g.uniform((c,), *gamma)
    sd[prefix + ".bias"] = g.normal((c,), 0.1)
    sd[prefix + ".running_mean"] = g.normal((c,), 0.1)
    sd[prefix + ".running_var"] = g.uniform((c,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _ln(sd, g: _Gen, prefix, c):
    sd[prefix + ".weight"] = g.uniform((c,), 0.8, 1.2)
    sd[prefix + ".bias"] = g.normal((c,), 0.05)


def _conv(sd, g: _Gen, prefix, cout, cin_g, kh, kw, bias, gain=2.0):
    fan_in = cin_g * kh * kw
    sd[prefix + ".weight"] = g.normal((cout, cin_g, kh, kw), math.sqrt(gain / fan_in))
    if bias:
        sd[prefix + ".bias"] = g.normal((cout,), 0.1)


def _linear(sd, g: _Gen, prefix, cout, cin, gain=1.0):
    sd[prefix + ".weight"] = g.normal((cout, cin), math.sqrt(gain / cin))
    sd[prefix + ".bias"] = g.normal((cout,), 0.02)


def _mlp(sd, g, prefix, dim):
    _linear(sd, g, prefix + ".fc1", 4 * dim, dim, gain=2.0)
    _linear(sd, g, prefix + ".fc2", dim, 4 * dim, gain=0.5)


def make_rec_state_dict(variant: str = "base", seed: int = 1234) -> "OrderedDict[str, torch.Tensor]":
    """Seeded SVTRv2 inference weights in the reference key schema (SGM keys omitted: training only,
    svtrv2.py:521)."""
    if variant not in REC_VARIANTS:
        raise ValueError(f"Unknown variant: {variant}. Choose from {list(REC_VARIANTS)}")
    cfg = REC_VARIANTS[variant]
    dims, nb, nl = cfg["dims"], cfg["num_blocks"], cfg["num_local"]
    g = _Gen(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    mid = dims[0] // 2
    _conv(sd, g, "stem.conv1", mid, 3, 3, 3, True)
    _bn(sd, g, "stem.bn1", mid)
    _conv(sd, g, "stem.conv2", dims[0], mid, 3, 3, True)
    _bn(sd, g, "stem.bn2", dims[0])
    for s in range(3):
        d = dims[s]
        for b in range(nb[s]):
            p = f"stages.{s}.blocks.{b}"
            _ln(sd, g, p + ".norm1", d)
            if b < nl[s]:
                groups = max(d // 32, 1)
                _conv(sd, g, p + ".mixer.conv1", d, d // groups, 3, 3, True)
                _bn(sd, g, p + ".mixer.bn1", d)
                _conv(sd, g, p + ".mixer.conv2", d, d // groups, 3, 3, True, gain=0.5)
                _bn(sd, g, p + ".mixer.bn2", d, gamma=(0.3, 0.7))
            else:
                _linear(sd, g, p + ".mixer.qkv", 3 * d, d, gain=2.0)
                _linear(sd, g, p + ".mixer.proj", d, d, gain=0.5)
            _ln(sd, g, p + ".norm2", d)
            _mlp(sd, g, p + ".mlp", d)
        if s < 2:
            _conv(sd, g, f"merges.{s}.conv", dims[s + 1], d, 3, 3, True, gain=1.0)
            _bn(sd, g, f"merges.{s}.norm", dims[s + 1])
    d = dims[2]
    _ln(sd, g, "backbone_norm", d)
    sd["frm.select_token"] = g.normal((1, 1, d), 0.5)
    _linear(sd, g, "frm.h_qkv", 3 * d, d, gain=2.0)
    _linear(sd, g, "frm.h_proj", d, d, gain=0.5)
    _ln(sd, g, "frm.h_norm", d)
    _mlp(sd, g, "frm.h_mlp", d)
    _ln(sd, g, "frm.h_norm2", d)
    _linear(sd, g, "frm.v_q", d, d, gain=2.0)
    _linear(sd, g, "frm.v_kv", 2 * d, d, gain=2.0)
    _linear(sd, g, "frm.v_proj", d, d, gain=1.0)
    _ln(sd, g, "frm.v_norm_q", d)
    _ln(sd, g, "frm.v_norm_kv", d)
    _mlp(sd, g, "frm.v_mlp", d)
    _ln(sd, g, "frm.v_norm2", d)
    _linear(sd, g, "head", NUM_CLASSES, d, gain=6.0)
    return sd


def make_det_state_dict(seed: int = 1234, dcn_offset_std: float = 1.5) -> "OrderedDict[str, torch.Tensor]":
    """Seeded DBNet++ (ResNet-50-DCN) weights in the reference key schema.

    Backbone keys follow torchvision's resnet50 naming under ``backbone.model.`` (backbone.py:16-37);
    the aliases ``backbone.layerN.*`` the reference's ``state_dict()`` also carries are not emitted
    (``fold_det`` accepts either).  ``dcn_offset_std`` sets the scale of the offset conv so offsets are
    a few pixels (the reference zero-inits it, dcn.py:28-29, which would hide sampling bugs).
    """
    g = _Gen(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    bb = "backbone.model."
    _conv(sd, g, bb + "conv1", 64, 3, 7, 7, False)
    _bn(sd, g, bb + "bn1", 64)
    inpl = 64
    for li, (nblk, w) in enumerate(zip(R50_BLOCKS, R50_WIDTH), start=1):
        for b in range(nblk):
            p = f"{bb}layer{li}.{b}"
            _conv(sd, g, p + ".conv1", w, inpl, 1, 1, False)
            _bn(sd, g, p + ".bn1", w)
            _conv(sd, g, p + ".conv2", w, w, 3, 3, False)
            if li >= 2:  # DeformableConv2d (backbone.py:28-31, dcn.py:17-32)
                fan_in = w * 9
                sd[p + ".conv2.offset_mask_conv.weight"] = g.normal((27, w, 3, 3), dcn_offset_std / math.sqrt(fan_in))
                sd[p + ".conv2.offset_mask_conv.bias"] = g.normal((27,), 0.5)
            _bn(sd, g, p + ".bn2", w)
            _conv(sd, g, p + ".conv3", 4 * w, w, 1, 1, False, gain=1.0)
            _bn(sd, g, p + ".bn3", 4 * w, gamma=(0.2, 0.6))
            if b == 0:
                _conv(sd, g, p + ".downsample.0", 4 * w, inpl, 1, 1, False, gain=1.0)
                _bn(sd, g, p + ".downsample.1", 4 * w, gamma=(0.4, 0.8))
            inpl = 4 * w
    for i, c in enumerate([256, 512, 1024, 2048]):
        _conv(sd, g, f"neck.lateral_convs.{i}", 256, c, 1, 1, True, gain=1.0)
        _conv(sd, g, f"neck.fpn_convs.{i}.conv", 256, 256, 3, 3, False)
        _bn(sd, g, f"neck.fpn_convs.{i}.bn", 256)
    _conv(sd, g, "neck.asf.conv_atten", 4, 1024, 1, 1, True, gain=4.0)
    for br in ("bin_conv", "thresh_conv"):
        _conv(sd, g, f"head.{br}.0.conv", 64, 256, 3, 3, False)
        _bn(sd, g, f"head.{br}.0.bn", 64)
        sd[f"head.{br}.1.weight"] = g.normal((64, 64, 2, 2), math.sqrt(2.0 / 64))
        sd[f"head.{br}.1.bias"] = g.normal((64,), 0.1)
        _bn(sd, g, f"head.{br}.2", 64)
        sd[f"head.{br}.4.weight"] = g.normal((64, 1, 2, 2), math.sqrt(4.0 / 64))
        sd[f"head.{br}.4.bias"] = g.normal((1,), 0.1)
    return sd


# ----------------------------------------------------------------------------- folding
def unwrap_checkpoint(ckpt) -> Dict[str, torch.Tensor]:
    """Accept ``{'model_state_dict': ...}`` or a bare state_dict; strip one leading ``module.``
    (pipeline2.py:46-52,75-80)."""
    sd = ckpt.get("model_state_dict", ckpt) if isinstance(ckpt, Mapping) else ckpt
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def _f32(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().to(torch.float64).cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def _fold_bn(w, b, sd, bn, out_axis=0) -> Tuple[np.ndarray, np.ndarray]:
    """conv/deconv weight ``w`` (+ optional bias ``b``) followed by eval-mode BN ``bn`` -> (w', b').
    Done in float64 then rounded once to float32."""
    gamma, beta = _f32(sd[bn + ".weight"]), _f32(sd[bn + ".bias"])
    mean, var = _f32(sd[bn + ".running_mean"]), _f32(sd[bn + ".running_var"])
    s = gamma / np.sqrt(var + BN_EPS)
    shape = [1] * w.ndim
    shape[out_axis] = -1
    w2 = _f32(w) * s.reshape(shape)
    b0 = _f32(b) if b is not None else np.zeros_like(mean)
    b2 = (b0 - mean) * s + beta
    return w2.astype(np.float32), b2.astype(np.float32)


def _plain(sd, name):
    return _f32(sd[name]).astype(np.float32)


def fold_rec(state_dict, variant: str = "base") -> "OrderedDict[str, np.ndarray]":
    sd = unwrap_checkpoint(state_dict)
    cfg = REC_VARIANTS[variant]
    dims, nb, nl = cfg["dims"], cfg["num_blocks"], cfg["num_local"]
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def put_conv_bn(dst, conv, bn):
        out[dst + ".w"], out[dst + ".b"] = _fold_bn(sd[conv + ".weight"], sd[conv + ".bias"], sd, bn)

    def put_lin(dst, src):
        out[dst + ".w"], out[dst + ".b"] = _plain(sd, src + ".weight"), _plain(sd, src + ".bias")

    put_conv_bn("stem.conv1", "stem.conv1", "stem.bn1")
    put_conv_bn("stem.conv2", "stem.conv2", "stem.bn2")
    for s in range(3):
        for b in range(nb[s]):
            p = f"stages.{s}.blocks.{b}"
            put_lin(p + ".norm1", p + ".norm1")
            if b < nl[s]:
                put_conv_bn(p + ".mixer.conv1", p + ".mixer.conv1", p + ".mixer.bn1")
                put_conv_bn(p + ".mixer.conv2", p + ".mixer.conv2", p + ".mixer.bn2")
            else:
                put_lin(p + ".mixer.qkv", p + ".mixer.qkv")
                put_lin(p + ".mixer.proj", p + ".mixer.proj")
            put_lin(p + ".norm2", p + ".norm2")
            put_lin(p + ".mlp.fc1", p + ".mlp.fc1")
            put_lin(p + ".mlp.fc2", p + ".mlp.fc2")
        if s < 2:
            put_conv_bn(f"merges.{s}", f"merges.{s}.conv", f"merges.{s}.norm")
    put_lin("backbone_norm", "backbone_norm")
    for n in ("h_qkv", "h_proj", "h_norm", "h_norm2", "v_kv", "v_proj", "v_norm_kv", "v_norm2"):
        put_lin("frm." + n, "frm." + n)
    for n in ("h_mlp", "v_mlp"):
        put_lin(f"frm.{n}.fc1", f"frm.{n}.fc1")
        put_lin(f"frm.{n}.fc2", f"frm.{n}.fc2")
    # Input-independent query of the vertical cross-attention: v_q(v_norm_q(select_token))
    # (svtrv2.py:228,233,236) -- the token is the same for every column, so fold it at load.
    tok = _f32(sd["frm.select_token"]).reshape(-1)
    mu, var = tok.mean(), tok.var()
    tn = (tok - mu) / np.sqrt(var + LN_EPS) * _f32(sd["frm.v_norm_q.weight"]) + _f32(sd["frm.v_norm_q.bias"])
    q = _f32(sd["frm.v_q.weight"]) @ tn + _f32(sd["frm.v_q.bias"])
    out["frm.select_token"] = tok.astype(np.float32)
    out["frm.vq"] = q.astype(np.float32)
    put_lin("head", "head")
    assert out["head.w"].shape == (NUM_CLASSES, dims[2])
    return out


def _bb_key(sd, li, b, rest):
    """Backbone tensors appear as backbone.model.layerN.* and, in real reference checkpoints, also under
    the alias backbone.layerN.* (layer1 = backbone.layer1.4.*)."""
    k = f"backbone.model.layer{li}.{b}.{rest}"
    if k in sd:
        return k
    alias = f"backbone.layer{li}.{'4.' if li == 1 else ''}{b}.{rest}"
    if alias in sd:
        return alias
    raise KeyError(k)


def fold_det(state_dict) -> "OrderedDict[str, np.ndarray]":
    sd = unwrap_checkpoint(state_dict)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    c1 = "backbone.model.conv1.weight" if "backbone.model.conv1.weight" in sd else "backbone.layer1.0.weight"
    b1 = "backbone.model.bn1" if "backbone.model.bn1.weight" in sd else "backbone.layer1.1"
    out["stem.w"], out["stem.b"] = _fold_bn(sd[c1], None, sd, b1)
    for li, nblk in enumerate(R50_BLOCKS, start=1):
        for b in range(nblk):
            p = f"layer{li}.{b}"

            def bnp(name):
                return _bb_key(sd, li, b, name + ".weight")[: -len(".weight")]

            out[p + ".conv1.w"], out[p + ".conv1.b"] = _fold_bn(sd[_bb_key(sd, li, b, "conv1.weight")], None, sd, bnp("bn1"))
            out[p + ".conv2.w"], out[p + ".conv2.b"] = _fold_bn(sd[_bb_key(sd, li, b, "conv2.weight")], None, sd, bnp("bn2"))
            if li >= 2:
                out[p + ".conv2.off.w"] = _plain(sd, _bb_key(sd, li, b, "conv2.offset_mask_conv.weight"))
                out[p + ".conv2.off.b"] = _plain(sd, _bb_key(sd, li, b, "conv2.offset_mask_conv.bias"))
            out[p + ".conv3.w"], out[p + ".conv3.b"] = _fold_bn(sd[_bb_key(sd, li, b, "conv3.weight")], None, sd, bnp("bn3"))
            if b == 0:
                out[p + ".down.w"], out[p + ".down.b"] = _fold_bn(
                    sd[_bb_key(sd, li, b, "downsample.0.weight")], None, sd, bnp("downsample.1"))
    for i in range(4):
        out[f"neck.lat{i}.w"] = _plain(sd, f"neck.lateral_convs.{i}.weight")
        out[f"neck.lat{i}.b"] = _plain(sd, f"neck.lateral_convs.{i}.bias")
        out[f"neck.fpn{i}.w"], out[f"neck.fpn{i}.b"] = _fold_bn(
            sd[f"neck.fpn_convs.{i}.conv.weight"], None, sd, f"neck.fpn_convs.{i}.bn")
    out["neck.asf.w"] = _plain(sd, "neck.asf.conv_atten.weight").reshape(4, 1024)
    out["neck.asf.b"] = _plain(sd, "neck.asf.conv_atten.bias")
    hw, hb = [], []
    for br, short in (("bin_conv", "bin"), ("thresh_conv", "thr")):
        w, b = _fold_bn(sd[f"head.{br}.0.conv.weight"], None, sd, f"head.{br}.0.bn")
        hw.append(w)
        hb.append(b)
        # ConvTranspose2d weight is (C_in, C_out, 2, 2) (head.py:13): BN scales along axis 1
        out[f"head.{short}.dc1.w"], out[f"head.{short}.dc1.b"] = _fold_bn(
            sd[f"head.{br}.1.weight"], sd[f"head.{br}.1.bias"], sd, f"head.{br}.2", out_axis=1)
        out[f"head.{short}.dc2.w"] = _plain(sd, f"head.{br}.4.weight")
        out[f"head.{short}.dc2.b"] = _plain(sd, f"head.{br}.4.bias")
    # both branches read the same neck feature (head.py:34-35): one 256 -> 128 conv
    out["head.conv.w"] = np.concatenate(hw, 0)
    out["head.conv.b"] = np.concatenate(hb, 0)
    return out


# ----------------------------------------------------------------------------- blob
BLOB_MAGIC = b"OCRVIW1\0"
_ENTRY = struct.Struct("<64sI4IQQ")  # name, ndim, dims[4], byte offset, element count


def pack_blob(folded: Mapping[str, np.ndarray]) -> bytes:
    """[magic 8][n u32][pad u32][entries n*104][float32 data, each tensor 16-byte aligned]."""
    names = list(folded)
    header = len(BLOB_MAGIC) + 8 + _ENTRY.size * len(names)
    off = (header + 15) // 16 * 16
    entries, chunks = [], []
    pos = off
    for n in names:
        a = np.ascontiguousarray(folded[n], dtype=np.float32)
        if a.ndim > 4 or len(n.encode()) > 63:
            raise ValueError(f"tensor {n}: unsupported rank/name")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        entries.append(_ENTRY.pack(n.encode(), a.ndim, *dims, pos, a.size))
        raw = a.tobytes()
        pad = (-len(raw)) % 16
        chunks.append(raw + b"\0" * pad)
        pos += len(raw) + pad
    head = BLOB_MAGIC + struct.pack("<II", len(names), 0) + b"".join(entries)
    head += b"\0" * (off - len(head))
    return head + b"".join(chunks)


def unpack_blob(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    assert blob[:8] == BLOB_MAGIC
    (n, _pad) = struct.unpack_from("<II", blob, 8)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off, cnt = _ENTRY.unpack_from(blob, 16 + i * _ENTRY.size)
        shape = (d0, d1, d2, d3)[:ndim]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, np.float32, cnt, off).reshape(shape)
    return out
