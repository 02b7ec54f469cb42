#!/usr/bin/env python3
"""Times the ring GEMM (ocrvi_test_gemm) on a few Linear shapes: usage gemm_bench.py [dtype] [M,K,N ...]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
DT = {"f32": 0, "bf16": 1, "f16": 2}[dt]
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(8192, 384, 1536), (61440, 384, 1536), (61440, 1536, 384), (122880, 256, 1024), (61440, 384, 384)]
peak = 157.3 if dt == "f32" else 2500.0
for M, K, N in shapes:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).cuda()
    w = np.ascontiguousarray((torch.randn(N, K, generator=g) / K ** 0.5).numpy())
    b = np.zeros(N, np.float32)
    out = torch.empty(M, N, device="cuda")
    ms = C.c_float(0)
    L.check(lib.ocrvi_test_gemm(0, DT, a.data_ptr(), w.ctypes.data, b.ctypes.data, None, M, K, N, 0, 0, 1 if dt == "f32" else 0, out.data_ptr(), 10, C.byref(ms)))
    fl = 2.0 * M * K * N
    print(f"{dt} M={M} K={K} N={N}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:8.1f} TFLOP/s ({fl/ms.value/1e9/peak*100:.1f}% of peak)", flush=True)
