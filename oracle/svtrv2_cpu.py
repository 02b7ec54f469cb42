"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain functional PyTorch fp32, of the reference's SVTRv2 inference forward
(model/rec2/svtrv2.py) operating directly on a reference-schema ``state_dict`` (BatchNorm is applied
as BatchNorm, not folded, so the product's folding is checked independently).

Pinned: tests/golden/rec_*.npz hold outputs of the reference's own ``SVTRv2`` module (imported by file
path in the build container, script tests/golden/make_golden.py) for the same seeded weights; this file
agrees with them to fp32 round-off (tests/test_oracle_cpu.py).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

VARIANTS = {  # svtrv2.py:391-407
    "tiny": ([64, 128, 256], [3, 6, 3], [3, 3, 0]),
    "small": ([96, 192, 256], [3, 6, 6], [3, 3, 0]),
    "base": ([128, 256, 384], [3, 6, 6], [3, 2, 0]),
}
HEAD_DIM = 32  # heads = dim // 32 (svtrv2.py:70-72,169-171)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _mlp(sd, p, x):  # svtrv2.py:38-39 (GELU = exact erf form)
    return _lin(sd, p + ".fc2", F.gelu(_lin(sd, p + ".fc1", x)))


def _mhsa(q, k, v):
    """q,k,v: (B, heads, N, hd) -> softmax(q k^T / sqrt(hd)) v  (svtrv2.py:82-85)."""
    a = (q @ k.transpose(-2, -1)) * (q.shape[-1] ** -0.5)
    return a.softmax(dim=-1) @ v


def local_mixing(sd, p, x, H, W):  # svtrv2.py:57-63
    B, N, D = x.shape
    g = max(D // 32, 1)
    y = x.transpose(1, 2).reshape(B, D, H, W)
    y = F.gelu(_bn(sd, p + ".bn1", F.conv2d(y, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], 1, 1, 1, g)))
    y = F.gelu(_bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], 1, 1, 1, g)))
    return y.flatten(2).transpose(1, 2)


def global_mixing(sd, p, x):  # svtrv2.py:77-86
    B, N, D = x.shape
    h = max(D // HEAD_DIM, 1)
    qkv = _lin(sd, p + ".qkv", x).reshape(B, N, 3, h, D // h).permute(2, 0, 3, 1, 4)
    o = _mhsa(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, N, D)
    return _lin(sd, p + ".proj", o)


def mixing_block(sd, p, x, H, W, is_local):  # svtrv2.py:98-101
    xn = _ln(sd, p + ".norm1", x)
    x = x + (local_mixing(sd, p + ".mixer", xn, H, W) if is_local else global_mixing(sd, p + ".mixer", xn))
    return x + _mlp(sd, p + ".mlp", _ln(sd, p + ".norm2", x))


def stem(sd, x):  # svtrv2.py:118-122
    x = F.gelu(_bn(sd, "stem.bn1", F.conv2d(x, sd["stem.conv1.weight"], sd["stem.conv1.bias"], 2, 1)))
    return F.gelu(_bn(sd, "stem.bn2", F.conv2d(x, sd["stem.conv2.weight"], sd["stem.conv2.bias"], 2, 1)))


def patch_merge(sd, i, x, H, W):  # svtrv2.py:131-138: conv 3x3 stride (2,1) + BN, no activation
    B, _, D = x.shape
    y = x.transpose(1, 2).reshape(B, D, H, W)
    y = _bn(sd, f"merges.{i}.norm", F.conv2d(y, sd[f"merges.{i}.conv.weight"], sd[f"merges.{i}.conv.bias"], (2, 1), 1))
    return y.flatten(2).transpose(1, 2), y.shape[2], y.shape[3]


def extract_features(sd, x, variant) -> Tuple[torch.Tensor, int, int, Dict[str, torch.Tensor]]:
    dims, nb, nl = VARIANTS[variant]
    taps: Dict[str, torch.Tensor] = {}
    x = stem(sd, x)
    B, D, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    taps["stem"] = x
    for s in range(3):
        for b in range(nb[s]):
            x = mixing_block(sd, f"stages.{s}.blocks.{b}", x, H, W, b < nl[s])
        taps[f"stage{s}"] = x
        if s < 2:
            x, H, W = patch_merge(sd, s, x, H, W)
            taps[f"merge{s}"] = x
    x = _ln(sd, "backbone_norm", x)
    taps["backbone_norm"] = x
    return x, H, W, taps


def frm(sd, x, H, W):  # svtrv2.py:192-247
    B, N, D = x.shape
    h = max(D // HEAD_DIM, 1)
    hd = D // h
    rows = x.reshape(B * H, W, D)
    qkv = _lin(sd, "frm.h_qkv", _ln(sd, "frm.h_norm", rows)).reshape(B * H, W, 3, h, hd).permute(2, 0, 3, 1, 4)
    o = _mhsa(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B * H, W, D)
    rows = rows + _lin(sd, "frm.h_proj", o)
    rows = rows + _mlp(sd, "frm.h_mlp", _ln(sd, "frm.h_norm2", rows))
    xh = rows.reshape(B, H, W, D)
    cols = xh.permute(0, 2, 1, 3).reshape(B * W, H, D)
    tq = sd["frm.select_token"].expand(B, W, -1).reshape(B * W, 1, D)
    q = _lin(sd, "frm.v_q", _ln(sd, "frm.v_norm_q", tq)).reshape(B * W, 1, h, hd).permute(0, 2, 1, 3)
    kv = _lin(sd, "frm.v_kv", _ln(sd, "frm.v_norm_kv", cols)).reshape(B * W, H, 2, h, hd).permute(2, 0, 3, 1, 4)
    o = _mhsa(q, kv[0], kv[1]).transpose(1, 2).reshape(B * W, 1, D)
    tq = tq + _lin(sd, "frm.v_proj", o)
    tq = tq + _mlp(sd, "frm.v_mlp", _ln(sd, "frm.v_norm2", tq))
    return tq.reshape(B, W, D)


@torch.no_grad()
def forward(sd, x: torch.Tensor, variant: str = "base", return_taps: bool = False):
    """(B,3,H,W) fp32 -> log_probs (T=W/4, B, 232) fp32   (svtrv2.py:503-536, targets=None)."""
    feats, H, W, taps = extract_features(sd, x.float(), variant)
    cf = frm(sd, feats, H, W)
    taps["frm"] = cf
    logits = _lin(sd, "head", cf).permute(1, 0, 2)
    lp = F.log_softmax(logits, dim=-1)
    return (lp, taps) if return_taps else lp


def greedy_ids(log_probs: torch.Tensor, blank_id: int = 0) -> List[List[int]]:
    """argmax -> collapse repeats -> drop blank (svtrv2.py:555-566).  Pad id 1 survives here (it breaks a
    repeat run) and is dropped later by the tokenizer (tokenizer.py:73)."""
    preds = log_probs.argmax(dim=-1).permute(1, 0).tolist()
    out = []
    for seq in preds:
        keep, prev = [], None
        for p in seq:
            if p != blank_id and p != prev:
                keep.append(p)
            prev = p
        out.append(keep)
    return out
