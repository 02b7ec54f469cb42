#!/usr/bin/env python3
"""Times the deformable-conv kernel (ocrvi_test_deform_conv) at the detector's layer shapes (16 pages 960x1280)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
dts = [(2, "f16")] if len(sys.argv) < 2 else [({"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}[a], a) for a in sys.argv[1:]]
for dt, name in dts:
    for N, Cc, H, W, st in ((16, 128, 120, 160, 1), (16, 256, 60, 80, 1), (16, 512, 30, 40, 1), (16, 128, 240, 320, 2), (16, 256, 120, 160, 2)):
        g = torch.Generator().manual_seed(1)
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        x = torch.randn(N, Cc, H, W, generator=g).cuda()
        off = (torch.randn(N, 18, Ho, Wo, generator=g) * 1.5).cuda()
        mask = torch.rand(N, 9, Ho, Wo, generator=g).cuda()
        w = np.ascontiguousarray((torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5).numpy())
        b = np.zeros(Cc, np.float32)
        out = torch.empty(N, Cc, Ho, Wo, device="cuda")
        ms = C.c_float(0)
        L.check(lib.ocrvi_test_deform_conv(0, dt, x.data_ptr(), off.data_ptr(), mask.data_ptr(), w.ctypes.data, b.ctypes.data, N, Cc, H, W, Cc, st, 1,
                                           out.data_ptr(), 10, C.byref(ms)))
        fl = 2.0 * N * Ho * Wo * Cc * Cc * 9
        peak = 157.3 if dt == 0 else (2500.0 / 3 if dt == 3 else 2500.0)
        print(f"{name} C={Cc} {H}x{W} s{st}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:8.1f} TFLOP/s ({fl/ms.value/1e9/peak*100:.1f}% of peak)", flush=True)
