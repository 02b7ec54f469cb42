"""Micro-benchmark of the attention kernel through the C ABI hook: python tools/attn_bench.py [dtype]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DT = {"f32": 0, "bf16": 1, "f16": 2}[dt]
lib = _lib.load()
for B, N, heads in [(256, 480, 8), (256, 240, 12), (768, 80, 12), (256, 256, 8), (256, 300, 8)]:
    qkv = torch.randn(B, N, 3 * heads * 32, device="cuda")
    out = torch.empty(B, N, heads * 32, device="cuda")
    ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_attention(0, DT, qkv.data_ptr(), B, N, heads, out.data_ptr(), 20, C.byref(ms)))
    fl = 4.0 * B * heads * N * N * 32
    print(f"B={B} N={N} heads={heads}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s ({fl/ms.value/1e9/25:.1f}% of bf16 MFMA peak)")
