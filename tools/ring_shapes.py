"""Ring-GEMM shapes of the two models through ocrvi_test_conv (no residual), HIP-event timed; OCRVI_TEST_PADC / OCRVI_RING_PROF apply."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DT = {"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}[dt]
lib = _lib.load()
# M, K, N, act
SHAPES = [(61440, 1536, 384, 0), (122880, 1024, 256, 0), (61440, 384, 1536, 2), (61440, 384, 1536, 0), (61440, 384, 1152, 0), (122880, 256, 1024, 2),
          (245760, 512, 128, 0), (76800, 1024, 256, 1), (76800, 256, 1024, 1), (19200, 2048, 512, 1), (307200, 128, 512, 1), (1228800, 64, 256, 1)]
iters = int(os.environ.get("ITERS", "10"))
for M, K, N, act in SHAPES:
    x = torch.randn(1, K, M // 64, 64, device="cuda")
    w = (np.random.randn(N, K, 1, 1) / np.sqrt(K)).astype(np.float32)
    b = np.zeros(N, np.float32)
    out = torch.empty(1, N, M // 64, 64, device="cuda")
    ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_conv(0, DT, x.data_ptr(), w.ctypes.data, b.ctypes.data, 1, K, M // 64, 64, N, 1, 1, 1, 1, act, out.data_ptr(), iters, C.byref(ms)))
    fl = 2.0 * M * N * K; by = (M * K + M * N) * 2
    t = max(ms.value, 1e-9) * 1e-3
    print(f"M{M} K{K} N{N} act{act}: {ms.value*1e3:8.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s", flush=True)
