"""Cycle breakdown of every ring-GEMM launch of one rec (B=256, 48x320) and one det (N=16, 960x1280) forward.
Run with OCRVI_RING_PROF=1 (development build aid; prints one line per launch on stderr)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import DBNetPP, SVTRv2
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
which = sys.argv[2] if len(sys.argv) > 2 else "rec,det"
if "rec" in which:
    rec = SVTRv2("base", dtype=dt)
    c = torch.randn(256, 3, 48, 320, device="cuda")
    print("== rec", file=sys.stderr, flush=True)
    rec.decode_greedy(c); torch.cuda.synchronize()
if "det" in which:
    det = DBNetPP(pretrained=False, dtype=dt)
    x = torch.randn(16, 3, 960, 1280, device="cuda")
    print("== det", file=sys.stderr, flush=True)
    det(x); torch.cuda.synchronize()
