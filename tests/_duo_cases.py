"""Helper of tests/test_gpu_duo.py: run in a child process with OCRVI_GEMM_DUO=1 (the switch is read once per process), so that the
opt-in duo ring GEMM (csrc/gemm_duo.h) is the kernel that runs; every case against an fp64 product of the same fp32 operands, twice,
bit-identical.  Prints one line per case and exits non-zero on the first failure."""
import ctypes as C
import sys

import numpy as np
import torch
import torch.nn.functional as F

from ocr_vi_invoice_amd import _lib as L

# M, K, N, act (0 none, 1 ReLU), residual, res_post, out_f32 -- all shapes gemm_duo_eligible admits: N tiles by 256 or 192, K >= 128
CASES = [
    (70000, 256, 256, 1, True, 0, 0),      # Bottleneck conv3: relu(conv + identity), f16x2 residual and output; ragged M, 64 columns per wave
    (40000, 1536, 384, 0, True, 0, 1),     # fc2 + fp32 residual stream (fp32 output): 48 columns per wave, 48 K-steps
    (50000, 256, 256, 0, True, 0, 1),      # proj + fp32 residual
    (3000, 384, 232, 0, False, 0, 1),      # CTC head: N 232 inside one 256-column tile, fp32 logits, 24 row tiles over 24 workgroups
    (33000, 128, 512, 1, True, 0, 0),      # four K-steps: the shortest the kernel takes; two column tiles
    (255, 256, 768, 0, False, 0, 0),       # a single partial row tile per workgroup: group 1 never starts
    (129, 384, 1152, 1, False, 0, 0),      # two row tiles: one per group, six 192-column tiles
    (61440, 384, 1152, 0, False, 0, 0),    # qkv at the bench's size: several tiles per group
    (20000, 512, 2048, 1, True, 0, 0),     # layer-4 conv3 shape
]


def main():
    lib = L.load()
    bad = 0
    for M, K, N, act, with_res, res_post, out_f32 in CASES:
        g = torch.Generator().manual_seed(M + K + N)
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / np.sqrt(K)
        b = torch.randn(N, generator=g) * 0.2
        res = torch.randn(M, N, generator=g) if with_res else None
        y = a.double() @ w.double().t() + b.double()
        if res is not None and not res_post:
            y = y + res.double()
        y = F.relu(y) if act == 1 else y
        if res is not None and res_post:
            y = y + res.double()
        ref = y.float()
        ad, rd = a.cuda(), (res.cuda() if res is not None else None)
        wh, bh = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(b.numpy())
        outs = []
        for _ in range(2):
            out = torch.empty((M, N), device="cuda")
            ms = C.c_float(0)
            L.check(lib.ocrvi_test_gemm(0, 3, ad.data_ptr(), wh.ctypes.data, bh.ctypes.data, rd.data_ptr() if rd is not None else None, M, K, N, act,
                                        res_post, out_f32, out.data_ptr(), 0, C.byref(ms)))
            outs.append(out.cpu())
        err = float((outs[0] - ref).abs().max() / (ref.pow(2).mean().sqrt() + 1e-12))
        ok = err < 2e-5 and torch.equal(outs[0], outs[1])
        print(f"duo case M{M} K{K} N{N} act{act} res{int(with_res)} f32o{out_f32}: rel err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        bad += not ok
    # the profiler's tag says which kernel ran
    L.check(lib.ocrvi_prof_reset())
    L.check(lib.ocrvi_prof_enable(1))
    a = torch.randn(4096, 256).cuda()
    w = np.ascontiguousarray((torch.randn(256, 256) / 16).numpy())
    out = torch.empty((4096, 256), device="cuda")
    ms = C.c_float(0)
    L.check(lib.ocrvi_test_gemm(0, 3, a.data_ptr(), w.ctypes.data, np.zeros(256, np.float32).ctypes.data, None, 4096, 256, 256, 0, 0, 1, out.data_ptr(), 0,
                                C.byref(ms)))
    L.check(lib.ocrvi_prof_enable(0))
    tags = list(L.prof_report())
    print("tags", tags, flush=True)
    if not any(t.startswith("gemm_duo") for t in tags):
        print("FAIL: the duo kernel did not run")
        bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
