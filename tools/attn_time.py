"""Per-shape timing of the attention kernels through ocrvi_test_attention (HIP-event ms): usage  python tools/attn_time.py f16x2"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import _lib
lib = _lib.load()
dt = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
DT = {"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}[dt]
for B, N, heads in [(256, 480, 8), (256, 240, 12), (256 * 12, 80, 12), (256 * 6, 80, 8)]:
    D = heads * 32
    qkv = torch.randn(B, N, 3 * D, device="cuda")
    out = torch.empty((B, N, D), device="cuda")
    ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_attention(0, DT, qkv.data_ptr(), B, N, heads, out.data_ptr(), 10, C.byref(ms)))
    fl = 4.0 * B * heads * N * N * 32
    print(f"{dt} B{B} N{N} heads{heads}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s", flush=True)
