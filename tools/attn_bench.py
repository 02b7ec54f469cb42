#!/usr/bin/env python3
"""Times the attention kernel (ocrvi_test_attention) at the recogniser's shapes: B sequences x N tokens x heads (head_dim 32)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocr_vi_invoice_amd import _lib as L
lib = L.load()
NAMES = {"f32": 0, "bf16": 1, "f16": 2}
for dt, name in [(NAMES[a], a) for a in sys.argv[1:]] or ((1, "bf16"), (2, "f16")):
    for B, N, heads in ((256, 480, 8), (256, 240, 12), (768, 80, 12)):
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(B, N, 3 * heads * 32, generator=g).cuda()
        out = torch.empty(B, N, heads * 32, device="cuda")
        ms = C.c_float(0)
        L.check(lib.ocrvi_test_attention(0, dt, qkv.data_ptr(), B, N, heads, out.data_ptr(), 20, C.byref(ms)))
        fl = 4.0 * B * heads * N * N * 32
        by = B * N * heads * 32 * 4 * 2
        print(f"{name} B={B} N={N} heads={heads}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:8.1f} TFLOP/s ({fl/ms.value/1e9/(157.3 if dt == 0 else 2500)*100:.1f}% of peak)  {by/ms.value/1e6:7.0f} GB/s algorithmic", flush=True)
