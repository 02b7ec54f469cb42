#!/usr/bin/env python3
"""Times the fused MLP kernel (ocrvi_test_mlp) at the recogniser's shapes (256 crops 48x320, SVTRv2-base)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
for dt, name in ((3, "f16x2"), (2, "f16")):
    for M, D in ((245760, 128), (122880, 256), (61440, 384)):
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(M, D, generator=g)).cuda()
        h = lambda t: np.ascontiguousarray(t.numpy(), dtype=np.float32)
        a = [h(torch.ones(D)), h(torch.zeros(D)), h(torch.randn(4 * D, D, generator=g) * (2.0 / D) ** 0.5), h(torch.zeros(4 * D)),
             h(torch.randn(D, 4 * D, generator=g) * (0.5 / (4 * D)) ** 0.5), h(torch.zeros(D))]
        xn = torch.zeros((M, D), device="cuda")
        ms = C.c_float(0)
        L.check(lib.ocrvi_test_mlp(0, dt, x.data_ptr(), *[v.ctypes.data for v in a], a[0].ctypes.data, a[1].ctypes.data, 1, M, D, xn.data_ptr(), 20, C.byref(ms)))
        fl = 2.0 * M * 8 * D * D
        print(f"{name} M={M} D={D}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:8.1f} TFLOP/s ({fl/ms.value/1e9/(2500 if dt != 3 else 2500/3)*100:.1f}% of the mode's MFMA roof)", flush=True)
