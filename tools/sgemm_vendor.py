#!/usr/bin/env python3
"""Reference point for the fp32 ring GEMM: the vendor library's fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS) at the recogniser's and
detector's Linear / 1x1 shapes.  Not part of the product path."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(61440, 1536, 384), (61440, 384, 1536), (122880, 1024, 256), (122880, 256, 1024), (61440, 1152, 384), (76800, 1024, 256),
          (76800, 256, 1024), (1228800, 256, 256), (307200, 256, 512)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda")
    for scale, tag in ((1.0, "random"), (0.0, "zeros")):
        x = a * scale
        w = torch.randn(N, K, device="cuda") * scale
        for _ in range(3):
            y = x @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = x @ w.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M{M} N{N} K{K} {tag}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s ({2.0*M*N*K/ms/1e9/157.3*100:.1f}% of 157.3)", flush=True)
