#!/usr/bin/env python3
"""One deformable-conv launch set for profiling: C, H, W, stride from argv (default the layer-3 shape)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
Cc, H, W, st = [int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (256, 60, 80, 1))]
dt = {"f32": 0, "bf16": 1, "f16": 2}[sys.argv[5] if len(sys.argv) > 5 else "f16"]
N = 16
g = torch.Generator().manual_seed(1)
Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
x = torch.randn(N, Cc, H, W, generator=g).cuda()
off = (torch.randn(N, 18, Ho, Wo, generator=g) * 1.5).cuda()
mask = torch.rand(N, 9, Ho, Wo, generator=g).cuda()
w = np.ascontiguousarray((torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5).numpy())
b = np.zeros(Cc, np.float32)
out = torch.empty(N, Cc, Ho, Wo, device="cuda")
ms = C.c_float(0)
L.check(lib.ocrvi_test_deform_conv(0, dt, x.data_ptr(), off.data_ptr(), mask.data_ptr(), w.ctypes.data, b.ctypes.data, N, Cc, H, W, Cc, st, 1, out.data_ptr(), 3, C.byref(ms)))
print(ms.value * 1e3, "us")
