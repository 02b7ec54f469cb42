#!/usr/bin/env python3
"""Times the ring GEMM (ocrvi_test_gemm) on a few Linear shapes and reports its error against an fp64 product of the same fp32 operands:
usage  gemm_bench.py [dtype[,dtype...]] [M,K,N ...]   (dtypes: f32 f16x2 f16 bf16; default f32,f16x2,f16)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
dts = (sys.argv[1] if len(sys.argv) > 1 else "f32,f16x2,f16").split(",")
CODE = {"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}
PEAK = {"f32": 157.3, "f16x2": 157.3, "f16": 2500.0, "bf16": 2500.0}
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(8192, 384, 1536), (61440, 384, 1536), (61440, 1536, 384), (122880, 256, 1024), (61440, 384, 384),
                                                                          (1228800, 64, 256), (1228800, 256, 64), (307200, 512, 128), (76800, 1024, 256)]
print("| M | K | N | dtype | us | TFLOP/s | of the fp32 MFMA peak (157.3) | max err / rms(C) | rms err / rms(C) |\n|---|---|---|---|---|---|---|---|---|")
for M, K, N in shapes:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).cuda()
    wt = torch.randn(N, K, generator=g) / K ** 0.5
    w = np.ascontiguousarray(wt.numpy())
    b = np.zeros(N, np.float32)
    rows = min(M, 8192)                                      # fp64 reference on a slab of rows (the whole product for the short shapes)
    ref = a[:rows].double() @ wt.cuda().double().t()
    rms = float(ref.pow(2).mean().sqrt())
    for dt in dts:
        out = torch.empty(M, N, device="cuda")
        ms = C.c_float(0)
        L.check(lib.ocrvi_test_gemm(0, CODE[dt], a.data_ptr(), w.ctypes.data, b.ctypes.data, None, M, K, N, 0, 0, 1, out.data_ptr(), 10, C.byref(ms)))
        fl = 2.0 * M * K * N
        e = out[:rows].double() - ref
        print(f"| {M} | {K} | {N} | {dt} | {ms.value*1e3:.1f} | {fl/ms.value/1e9:.1f} | {fl/ms.value/1e9/157.3*100:.0f} % | {float(e.abs().max())/rms:.2e} | {float(e.pow(2).mean().sqrt())/rms:.2e} |", flush=True)
