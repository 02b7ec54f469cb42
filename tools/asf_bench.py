#!/usr/bin/env python3
"""Times the ASF kernels inside one detector forward (16 pages 960x1280) through the library's profiler."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import DBNetPP, _lib
lib = _lib.load()
x = torch.randn(16, 3, 960, 1280, device="cuda")
for dt in sys.argv[1:] or ["f16"]:
    det = DBNetPP(pretrained=False, dtype=dt)
    det(x); torch.cuda.synchronize()
    lib.ocrvi_prof_reset(); lib.ocrvi_prof_enable(1)
    for _ in range(3): det(x)
    torch.cuda.synchronize(); lib.ocrvi_prof_enable(0)
    rep = _lib.prof_report()
    v = rep["asf_fused"]
    print(f"{dt}: asf_fused {v['ms']/v['launches']*1e3:8.1f} us/launch  (det forward {sum(u['ms'] for u in rep.values())/3:.2f} ms)", flush=True)
    del det
