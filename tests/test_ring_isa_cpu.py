"""The ring GEMM (csrc/gemm_ring.h) synchronises its LDS-DMA ring and its inline-asm residual loads with hand-counted
`s_waitcnt vmcnt(N)`.  That protocol holds only if the compiler adds no vector-memory operation of its own (spills) and no vmcnt
wait of its own inside the pipeline, and leaves the asm loads' destination registers alone until the counted wait.  A run that passes
says nothing about this (a DMA that happens to land early hides a missing wait), so the generated ISA is checked instead.
Cross-compiles for gfx950; no GPU needed."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("src,nmin", [("conv_bf16.hip", 18), ("conv_f16.hip", 18), ("conv_f32.hip", 3), ("conv_f16x2.hip", 6)])
def test_ring_gemm_isa_keeps_the_vmcnt_protocol(src, nmin):
    import check_ring_isa
    rep = check_ring_isa.check(src)
    assert len(rep) >= nmin      # x3 activations; 16-bit: {256, 128}-row tiles x {16-bit, fp32} output, 3x3 mode, 256x64; fp32: 128-row; f16x2: 256 / 128
    for name, r in rep.items():
        assert r["mfma"] > 0 and r["asm_loads"] > 0, name
        assert r["scratch"] == 0, f"{name}: {r['scratch']} scratch instructions (register spills)"
        assert not r["touches"], f"{name}: asm-loaded registers touched before the counted wait: {r['touches'][:3]}"
        assert r["compiler_vmcnt_waits"] == ["s_waitcnt vmcnt(0)"], f"{name}: compiler vmcnt waits {r['compiler_vmcnt_waits']}"


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("src", ["conv_bf16.hip", "conv_f16.hip", "conv_f32.hip", "conv_f16x2.hip", "mlp_fused.hip", "attention.hip", "stem_pool.hip"])
def test_no_mfma_kernel_of_the_library_spills(src):
    """dcn_pipe, offs_conv and mlp_fused count their vmcnt by hand like the ring GEMM: a spill there is an uncounted VMEM operation.  In the
    other MFMA kernels (conv_gemm, gconv32, attention) it is a performance bug.  None may contain a scratch instruction."""
    import check_ring_isa
    rep = check_ring_isa.check_scratch(src)
    assert rep, src
    bad = {n: v for n, v in rep.items() if v[0] != 0}
    assert not bad, f"kernels with scratch instructions: {bad}"
    assert all(v[1] > 0 for v in rep.values())


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_duo_gemm_isa_keeps_its_protocol():
    """gemm_duo.h counts vmcnt by hand like the ring and sets M0 without saving it: no scratch, no compiler vmcnt wait inside the pipeline,
    no compiler use of M0."""
    import check_ring_isa
    rep = check_ring_isa.check_duo()
    assert len(rep) >= 12
    for name, r in rep.items():
        assert r["mfma"] > 0 and r["scratch"] == 0, (name, r["scratch"])
        assert len(r["compiler_vmcnt_waits"]) <= 1, (name, r["compiler_vmcnt_waits"])
        assert not r["m0_uses"], (name, r["m0_uses"][:3])


@pytest.mark.skipif(not shutil.which("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_mlp_x2_isa_keeps_its_ring_protocol():
    """mlp_x2.hip counts the vmcnt of its weight ring by hand: no scratch instruction inside the chunk loop of any build, none at all in the
    8-wave builds (D = 128 / 256); the 4-wave D = 384 build fills all 512 registers and may spill loop-invariant values outside the loop."""
    import check_ring_isa
    rep = check_ring_isa.check_mlp_x2()
    assert len(rep) >= 3
    for name, (inner, outer, mfma) in rep.items():
        assert mfma > 0 and inner == 0, (name, inner)
        if "ILi384E" not in name:
            assert outer == 0, (name, outer)
