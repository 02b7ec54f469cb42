"""Grouped 3x3 (LocalMixing) micro-benchmark: direct halo-tile kernel vs conv_gemm (OCRVI_GCONV32=0), with and without GELU."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import _lib
lib = _lib.load()
for name, N, Cc, H, W, Co, g, act in [("d128 gelu",256,128,12,80,128,4,2),("d128 none",256,128,12,80,128,4,0),("d256 gelu",256,256,6,80,256,8,2),("d256 none",256,256,6,80,256,8,0)]:
    x = torch.randn(N, Cc, H, W, device="cuda")
    w = (np.random.randn(Co, Cc // g, 3, 3) / 17).astype(np.float32); b = np.zeros(Co, np.float32)
    out = torch.empty(N, Co, H, W, device="cuda"); ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_conv(0, 1, x.data_ptr(), w.ctypes.data, b.ctypes.data, N, Cc, H, W, Co, 3, 1, 1, g, act, out.data_ptr(), 20, C.byref(ms)))
    print(name, os.environ.get("OCRVI_GCONV32","1"), f"{ms.value*1e3:.1f} us", flush=True)
