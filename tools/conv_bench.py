"""Micro-benchmark of conv_gemm shapes through the C ABI hook (ocrvi_test_conv / ocrvi_test_deform_conv), HIP-event timed.
usage: python tools/conv_bench.py [dtype]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
DT = {"f32": 0, "bf16": 1, "f16": 2}[dt]
lib = _lib.load()
peak = 2500e12 if dt != "f32" else 157e12
esz = 4 if dt == "f32" else 2
# name, N, C, H, W, Co, ks, stride, groups, act
SHAPES = [
    ("L1 c1x1 K64 N256",   16, 64, 240, 320, 256, 1, 1, 1, 1),
    ("L1 c1x1 K256 N64",   16, 256, 240, 320, 64, 1, 1, 1, 1),
    ("L2 c1x1 K128 N512",  16, 128, 120, 160, 512, 1, 1, 1, 1),
    ("L3 c1x1 K256 N1024", 16, 256, 60, 80, 1024, 1, 1, 1, 1),
    ("L3 c1x1 K1024 N256", 16, 1024, 60, 80, 256, 1, 1, 1, 1),
    ("fpn c3x3 K2304 N256",16, 256, 240, 320, 256, 3, 1, 1, 1),
    ("L1 c3x3 K576 N64",   16, 64, 240, 320, 64, 3, 1, 1, 1),
    ("rec fc1 K384 N1536", 256, 384, 240, 1, 1536, 1, 1, 1, 2),
    ("rec fc1 noact",      256, 384, 240, 1, 1536, 1, 1, 1, 0),
    ("rec fc2 K1536 N384", 256, 1536, 240, 1, 384, 1, 1, 1, 0),
    ("rec qkv K384 N1152", 256, 384, 240, 1, 1152, 1, 1, 1, 0),
    ("rec fc1 K256 N1024", 256, 256, 480, 1, 1024, 1, 1, 1, 2),
    ("rec fc1 K128 N512",  256, 128, 960, 1, 512, 1, 1, 1, 2),
    ("rec gconv d128",     256, 128, 12, 80, 128, 3, 1, 4, 2),
]
for name, N, Cc, H, W, Co, ks, st, g, act in SHAPES:
    x = torch.randn(N, Cc, H, W, device="cuda")
    w = (np.random.randn(Co, Cc // g, ks, ks) / np.sqrt(Cc // g * ks * ks)).astype(np.float32)
    b = np.zeros(Co, np.float32)
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
    out = torch.empty(N, Co, Ho, Wo, device="cuda")
    ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_conv(0, DT, x.data_ptr(), w.ctypes.data, b.ctypes.data, N, Cc, H, W, Co, ks, st, st, g, act, out.data_ptr(), 20, C.byref(ms)))
    M = N * Ho * Wo
    fl = 2.0 * M * Co * (Cc // g) * ks * ks
    by = (N * H * W * Cc + M * Co) * esz
    t = ms.value * 1e-3
    roof = max(fl / peak, by / 6.3e12)
    print(f"{name:22s} {ms.value*1e3:8.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s   roof {roof*1e6:7.1f} us ({roof/t*100:5.1f}%)")
# DCN shapes
for name, N, Cc, H, W, st in [("dcn L2 C128 s1", 16, 128, 120, 160, 1), ("dcn L3 C256 s1", 16, 256, 60, 80, 1), ("dcn L4 C512 s1", 16, 512, 30, 40, 1), ("dcn L3 C256 s2", 16, 256, 120, 160, 2)]:
    x = torch.randn(N, Cc, H, W, device="cuda")
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    off = torch.randn(N, 18, Ho, Wo, device="cuda") * 2
    mask = torch.rand(N, 9, Ho, Wo, device="cuda")
    w = (np.random.randn(Cc, Cc, 3, 3) / np.sqrt(Cc * 9)).astype(np.float32)
    out = torch.empty(N, Cc, Ho, Wo, device="cuda")
    ms = C.c_float(0)
    _lib.check(lib.ocrvi_test_deform_conv(0, DT, x.data_ptr(), off.data_ptr(), mask.data_ptr(), w.ctypes.data, None, N, Cc, H, W, Cc, st, 1, out.data_ptr(), 20, C.byref(ms)))
    M = N * Ho * Wo
    fl = 2.0 * M * Cc * Cc * 9
    t = ms.value * 1e-3
    print(f"{name:22s} {ms.value*1e3:8.1f} us  {fl/t/1e12:7.1f} TF/s   ({fl/t/peak*100:5.1f}% of MFMA peak)")
