#!/usr/bin/env python3
"""Times the detector's stem (conv 7x7 / 2 + ReLU + max-pool 3x3 / 2) at 16 pages 960x1280: the fused f16x2 kernel against the two-kernel form."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
g = torch.Generator().manual_seed(1)
N, H, W = 16, 960, 1280
x = torch.rand(N, 3, H, W, generator=g).cuda()
w = np.ascontiguousarray((torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).numpy())
b = np.zeros(64, np.float32)
out = torch.empty(N, 64, H // 4, W // 4, device="cuda")
for name, dt, fused in (("f16x2 fused", 3, 1), ("f16x2 two kernels", 3, 0), ("f16 two kernels", 2, 0)):
    ms = C.c_float(0)
    L.check(lib.ocrvi_test_stem_pool(0, dt, x.data_ptr(), w.ctypes.data, b.ctypes.data, N, H, W, fused, out.data_ptr(), 10, C.byref(ms)))
    fl = 2.0 * N * (H // 2) * (W // 2) * 64 * 147
    print(f"{name}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TFLOP/s", flush=True)
