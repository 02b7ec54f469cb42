"""Per-shape kernel timing of one det forward (N=16, 960x1280) and one rec forward (B=256, 48x320): achieved TFLOP/s and GB/s per
conv shape against its own roofline (max of MFMA-bound and HBM-bound time).  Run with OCRVI_PROF_DETAIL=1."""
import json, sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ocr_vi_invoice_amd import DBNetPP, SVTRv2, _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
lib = _lib.load()
det = DBNetPP(pretrained=False, dtype=dt); rec = SVTRv2("base", dtype=dt)
x = torch.randn(16, 3, 960, 1280, device="cuda"); c = torch.randn(256, 3, 48, 320, device="cuda")
for which, fn in (("det", lambda: det(x)), ("rec", lambda: rec.decode_greedy(c))):
    fn(); torch.cuda.synchronize()
    lib.ocrvi_prof_reset(); lib.ocrvi_prof_enable(1)
    for _ in range(3): fn()
    torch.cuda.synchronize(); lib.ocrvi_prof_enable(0)
    rep = _lib.prof_report()
    tot = sum(v["ms"] for v in rep.values()) / 3
    print(f"== {which} {dt}: {tot:.2f} ms per forward")
    peak = {"f32": 157.3e12, "f16x2": 2500e12 / 3}.get(dt, 2500e12)   # f16x2: three 16-bit partial products per product
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:45]:
        ms = v["ms"] / v["launches"]; n = v["launches"] // 3
        tf = v["flops"] / v["launches"] / ms / 1e9; gb = v["bytes"] / v["launches"] / ms / 1e6
        ideal = max(v["flops"] / v["launches"] / peak, v["bytes"] / v["launches"] / 6.3e12) * 1e3
        print(f"{v['ms']/3:8.3f} ms  x{n:3d}  {ms*1e3:8.1f} us/launch  {tf:7.1f} TF/s  {gb:7.0f} GB/s  roof {ideal*1e3:7.1f} us ({ideal/ms*100:5.1f}%)  {k}")
