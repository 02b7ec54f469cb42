"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in plain functional PyTorch fp32, of the reference's DBNet++ inference forward
(model/det/{dbnet,backbone,dcn,neck,head,layers}.py) on a reference-schema ``state_dict``.

What pins it:
* neck + head (model/det/neck.py, head.py, layers.py): tests/golden/det_neckhead.npz holds outputs of the
  reference's own ``FPN_ASF`` / ``DBHead`` modules (imported by file path; tests/golden/make_golden.py).
* ResNet-50 backbone + modulated deformable conv: the reference delegates both to **torchvision**
  (unpinned version; absent from the build container, so backbone.py / dcn.py cannot be imported).  This
  file restates torchvision's published semantics -- Bottleneck v1.5 (stride on the 3x3), stem 7x7/2 +
  maxpool 3x3/2, and ``ops.deform_conv2d`` (DCNv2: offset channel 2k = dy, 2k+1 = dx of tap k = 3i+j,
  zero-padded bilinear sampling, per-tap mask) -- **parity unpinned** by any reference-run output.  It is
  self-checked by two independent DCN formulations (explicit corner gather vs ``F.grid_sample``) and the
  zero-offset / mask=0.5 identity implied by dcn.py:28-29 (tests/test_oracle_cpu.py).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

R50_BLOCKS = [3, 4, 6, 3]


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


# ------------------------------------------------------------------ deformable conv (torchvision semantics)
def _sample_positions(offset, H, W, Ho, Wo, stride, pad=1, dil=1):
    """offset: (N,18,Ho,Wo) -> py, px: (N,9,Ho,Wo) absolute sampling coordinates."""
    dev = offset.device
    ho = torch.arange(Ho, device=dev, dtype=torch.float32).view(1, 1, Ho, 1) * stride - pad
    wo = torch.arange(Wo, device=dev, dtype=torch.float32).view(1, 1, 1, Wo) * stride - pad
    ki = (torch.arange(9, device=dev) // 3).float().view(1, 9, 1, 1) * dil
    kj = (torch.arange(9, device=dev) % 3).float().view(1, 9, 1, 1) * dil
    py = ho + ki + offset[:, 0::2]
    px = wo + kj + offset[:, 1::2]
    return py, px


def deform_conv2d_gather(x, offset, mask, weight, stride):
    """Explicit 4-corner gather.  x (N,C,H,W); offset (N,18,Ho,Wo); mask (N,9,Ho,Wo); weight (Co,C,3,3)."""
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[-2:]
    py, px = _sample_positions(offset, H, W, Ho, Wo, stride)
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    y0, x0 = torch.floor(py), torch.floor(px)
    ly, lx = py - y0, px - x0
    hy, hx = 1 - ly, 1 - lx
    y0, x0 = y0.long(), x0.long()
    y1, x1 = y0 + 1, x0 + 1
    xf = x.reshape(N, C, H * W)
    cols = torch.zeros(N, C, 9 * Ho * Wo, dtype=x.dtype)
    for yy, xx, wgt in ((y0, x0, hy * hx), (y0, x1, hy * lx), (y1, x0, ly * hx), (y1, x1, ly * lx)):
        ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(N, 1, -1).expand(-1, C, -1)
        v = torch.gather(xf, 2, idx)
        cols += v * (wgt * ok * mask).reshape(N, 1, -1)
    cols = cols.reshape(N, C * 9, Ho * Wo)                      # k index = c*9 + tap
    out = weight.reshape(weight.shape[0], -1) @ cols            # (N,Co,Ho*Wo)
    return out.reshape(N, -1, Ho, Wo)


def deform_conv2d_gridsample(x, offset, mask, weight, stride):
    """Independent formulation: one ``F.grid_sample`` (bilinear, zeros padding, align_corners=True) per tap."""
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[-2:]
    py, px = _sample_positions(offset, H, W, Ho, Wo, stride)
    out = 0
    for k in range(9):
        gx = 2 * px[:, k] / max(W - 1, 1) - 1
        gy = 2 * py[:, k] / max(H - 1, 1) - 1
        s = F.grid_sample(x, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="zeros", align_corners=True)
        s = s * mask[:, k:k + 1]
        out = out + F.conv2d(s, weight[:, :, k // 3, k % 3].unsqueeze(-1).unsqueeze(-1))
    return out


def dcn_module(sd, p, x, stride, impl=deform_conv2d_gather):
    """DeformableConv2d.forward (dcn.py:41-59): 27-channel conv -> offsets = ch 0..17 (chunk/cat is the
    identity on them), mask = sigmoid(ch 18..26)."""
    om = F.conv2d(x, sd[p + ".offset_mask_conv.weight"], sd[p + ".offset_mask_conv.bias"], stride, 1)
    return impl(x, om[:, :18], torch.sigmoid(om[:, 18:]), sd[p + ".weight"], stride)


# ------------------------------------------------------------------ backbone
def bottleneck(sd, p, x, stride, has_down, dcn):
    idn = x
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    if dcn:
        y = dcn_module(sd, p + ".conv2", y, stride)
    else:
        y = F.conv2d(y, sd[p + ".conv2.weight"], None, stride, 1)
    y = F.relu(_bn(sd, p + ".bn2", y))
    y = _bn(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]))
    if has_down:
        idn = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    return F.relu(y + idn)


def backbone(sd, x, dcn=True) -> List[torch.Tensor]:
    """ResNet.forward (backbone.py:55-60) -> [c2, c3, c4, c5]."""
    bb = "backbone.model."
    y = F.relu(_bn(sd, bb + "bn1", F.conv2d(x, sd[bb + "conv1.weight"], None, 2, 3)))
    y = F.max_pool2d(y, 3, 2, 1)
    feats = []
    for li, nblk in enumerate(R50_BLOCKS, start=1):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            y = bottleneck(sd, f"{bb}layer{li}.{b}", y, stride, b == 0, dcn and li >= 2)
        feats.append(y)
    return feats


# ------------------------------------------------------------------ neck + head
def _cbr(sd, p, x):  # ConvBnRelu (layers.py:13-18)
    return F.relu(_bn(sd, p + ".bn", F.conv2d(x, sd[p + ".conv.weight"], None, 1, 1)))


def neck(sd, feats, return_levels=False):
    """FPN_ASF.forward + ScaleFeatureSelection.forward (neck.py:26-46,57-79)."""
    def lat(i, t):
        return F.conv2d(t, sd[f"neck.lateral_convs.{i}.weight"], sd[f"neck.lateral_convs.{i}.bias"])

    last = lat(3, feats[3])
    ps = [_cbr(sd, "neck.fpn_convs.3", last)]
    for i in (2, 1, 0):
        last = lat(i, feats[i]) + F.interpolate(last, size=feats[i].shape[-2:], mode="nearest")
        ps.insert(0, _cbr(sd, f"neck.fpn_convs.{i}", last))
    size = ps[0].shape[-2:]
    ups = [ps[0]] + [F.interpolate(p, size=size, mode="bilinear", align_corners=True) for p in ps[1:]]
    score = F.softmax(F.conv2d(torch.cat(ups, 1), sd["neck.asf.conv_atten.weight"], sd["neck.asf.conv_atten.bias"]), 1)
    out = sum(u * score[:, i:i + 1] for i, u in enumerate(ups))
    return (out, ps) if return_levels else out


def head(sd, x, k=50.0) -> Dict[str, torch.Tensor]:
    """DBHead.forward (head.py:32-48)."""
    def branch(p):
        y = _cbr(sd, p + ".0", x)
        y = F.conv_transpose2d(y, sd[p + ".1.weight"], sd[p + ".1.bias"], 2)
        y = F.relu(_bn(sd, p + ".2", y))
        return F.conv_transpose2d(y, sd[p + ".4.weight"], sd[p + ".4.bias"], 2)

    bl, tl = branch("head.bin_conv"), branch("head.thresh_conv")
    b, t = torch.sigmoid(bl), torch.sigmoid(tl)
    return {"binary": b, "thresh": t, "thresh_binary": torch.reciprocal(1 + torch.exp(-k * (b - t))),
            "bin_logits": bl, "thresh_logits": tl}


@torch.no_grad()
def forward(sd, x, return_feats=False):
    """DBNetPP.forward (dbnet.py:13-17): (N,3,H,W) fp32, H,W % 32 == 0 -> dict of five (N,1,H,W) maps."""
    feats = backbone(sd, x.float())
    fused = neck(sd, feats)
    out = head(sd, fused)
    if return_feats:
        out = dict(out, c2=feats[0], c3=feats[1], c4=feats[2], c5=feats[3], fused=fused)
    return out
