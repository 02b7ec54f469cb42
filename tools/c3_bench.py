import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from ocr_vi_invoice_amd import _lib
lib = _lib.load()
for name, N, Cc, H, W, Co, act in [("fpn p2 K2304 N256", 16, 256, 240, 320, 256, 1), ("head K2304 N128", 16, 256, 240, 320, 128, 1), ("fpn p3 N256", 16, 256, 120, 160, 256, 1)]:
    x = torch.randn(N, Cc, H, W, device="cuda")
    w = (np.random.randn(Co, Cc, 3, 3) / np.sqrt(Cc * 9)).astype(np.float32)
    b = np.zeros(Co, np.float32)
    out = torch.empty(N, Co, H, W, device="cuda")
    ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_conv(0, 3, x.data_ptr(), w.ctypes.data, b.ctypes.data, N, Cc, H, W, Co, 3, 1, 1, 1, act, out.data_ptr(), 5, C.byref(ms)))
    fl = 2.0 * N * H * W * Co * Cc * 9
    print(f"{name:22s} {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s", flush=True)
