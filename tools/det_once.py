import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import DBNetPP
det = DBNetPP(pretrained=False, dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16")
x = torch.randn(16, 3, 960, 1280, device="cuda")
for _ in range(3): det(x)
torch.cuda.synchronize()
