"""Static checks of the ring-GEMM code objects that the kernel's hand-counted vmcnt protocol relies on (gemm_ring.h):

  1. no scratch (spill) instructions: a spill is a VMEM operation the protocol does not count, and hipcc drains vmcnt(0) around it;
  2. the destination registers of the inline-asm residual loads are not read or written by anything between the load and the
     inline-asm `s_waitcnt vmcnt(N)` that retires it (the compiler believes they are valid immediately);
  3. the compiler itself inserts exactly one vmcnt wait (for the bias loads before the pipeline starts): any other one would
     drain the DMA ring, which the compiler cannot see.

usage: python tools/check_ring_isa.py [conv_bf16.hip ...]   (cross-compiles with hipcc -S for gfx950; no GPU needed)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ocr_vi_invoice_amd", "csrc")


def kernel_bodies(asm_text):
    lines = asm_text.split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"(_ZN5ocrvi16gemm_ring_kernel\w+):", l)
        if not m:
            continue
        body, inasm = [], False
        for t in lines[i + 1:]:
            t = t.strip()
            if "s_endpgm" in t:
                break
            if "#ASMSTART" in t:
                inasm = True
                continue
            if "#ASMEND" in t:
                inasm = False
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            body.append((t, inasm))
        out[m.group(1)] = body
    return out


def regs_of(text):
    s = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        s |= {int(c)} if c else set(range(int(a), int(b) + 1))
    return s


def check(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", os.path.join(CSRC, src), "-o", out], check=True, capture_output=True)
        bodies = kernel_bodies(open(out).read())
    assert bodies, "no gemm_ring kernels found in " + src
    report = {}
    for name, body in bodies.items():
        scratch = sum("scratch_" in t for t, _ in body)
        touches, loads = [], 0
        for i, (t, a) in enumerate(body):
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off$", t)
            if not (m and a):
                continue
            loads += 1
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for tt, aa in body[i + 1:]:
                if aa and tt.startswith("s_waitcnt vmcnt"):
                    break
                if not tt.startswith("global_load_dwordx4") and regs_of(tt) & dst:
                    touches.append((t, tt))
        cwaits = [t for t, a in body if not a and t.startswith("s_waitcnt") and "vmcnt" in t]
        report[name] = dict(instructions=len(body), scratch=scratch, asm_loads=loads, touches=touches, compiler_vmcnt_waits=cwaits,
                            mfma=sum("v_mfma" in t for t, _ in body))
    return report


if __name__ == "__main__":
    ok = True
    for src in sys.argv[1:] or ["conv_bf16.hip"]:
        for name, r in check(src).items():
            good = r["scratch"] == 0 and not r["touches"] and len(r["compiler_vmcnt_waits"]) == 1
            ok &= good
            print(("ok   " if good else "FAIL ") + name, {k: (len(v) if isinstance(v, list) else v) for k, v in r.items()})
    sys.exit(0 if ok else 1)
