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


def kernel_bodies(asm_text, prefix="_ZN5ocrvi16gemm_ring_kernel"):
    """name -> list of (text, inside_inline_asm) for every kernel whose mangled name starts with `prefix`; labels are kept (as
    '.LBB..:' entries) so that the checks can follow the control flow"""
    lines = asm_text.split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"(" + re.escape(prefix) + r"\w*):", l)
        if not m:
            continue
        body, inasm = [], False
        for t in lines[i + 1:]:
            t = t.strip()
            if t.startswith(".Lfunc_end"):        # (not the first s_endpgm: the block layout may put an exit path ahead of the main loop)
                break
            if "s_endpgm" in t:
                body.append(("s_endpgm", False))
                continue
            if "#ASMSTART" in t:
                inasm = True
                continue
            if "#ASMEND" in t:
                inasm = False
                continue
            if re.match(r"\.LBB\w+:", t):
                body.append((t.split(":")[0] + ":", False))
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            body.append((t, inasm))
        out[m.group(1)] = body
    return out


def touches_before_wait(body, start, dst):
    """Instructions that read or write a register of `dst` between body[start] and the inline-asm `s_waitcnt vmcnt` that retires the asm
    load issued just before body[start] -- followed to the END OF THE BASIC BLOCK only (first label or branch): inside a block the text
    order is the execution order; across blocks it is not (the block layout interleaves paths that the wave-uniform flags of the kernel
    make mutually exclusive, and a path-insensitive walk reports registers that a different path legitimately reuses).  What this
    catches is the realistic failure: a register copy or spill the allocator places right behind the load."""
    hits = []
    for t, a in body[start:]:
        if (a and t.startswith("s_waitcnt vmcnt")) or t.endswith(":") or t.startswith("s_cbranch") or t.startswith("s_branch") or t == "s_endpgm":
            break
        if not t.startswith("global_load_dwordx4") and regs_of(t) & dst:
            hits.append(t)
    return hits


ALL_SOURCES = ["conv_bf16.hip", "conv_f16.hip", "conv_f32.hip", "conv_f16x2.hip", "duo_f16x2.hip", "mlp_fused.hip", "attention.hip", "stem_pool.hip", "mlp_x2.hip"]
_ASM = {}


def asm_of(src):
    """gfx950 assembly of one translation unit (hipcc -S, device only).  The first call compiles ALL_SOURCES in parallel and keeps the
    text for the life of the process, so that the checks of one test session share one compilation of each file."""
    if src not in _ASM:
        from concurrent.futures import ThreadPoolExecutor
        todo = [f for f in dict.fromkeys([src] + ALL_SOURCES) if f not in _ASM]
        with tempfile.TemporaryDirectory() as td:
            def one(f):
                out = os.path.join(td, f + ".s")
                subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                                "--cuda-device-only", os.path.join(CSRC, f), "-o", out], check=True, capture_output=True)
                return f, open(out).read()
            with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
                for f, txt in ex.map(one, todo):
                    _ASM[f] = txt
    return _ASM[src]


def regs_of(text):
    s = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        s |= {int(c)} if c else set(range(int(a), int(b) + 1))
    return s


def check(src):
    bodies = kernel_bodies(asm_of(src))
    assert bodies, "no gemm_ring kernels found in " + src
    report = {}
    for name, body in bodies.items():
        scratch = sum("scratch_" in t for t, _ in body)
        touches, loads = [], 0
        for i, (t, a) in enumerate(body):
            # the residual loads: (64-bit address, off) or (32-bit offset, SGPR base) form
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], (?:v\[\d+:\d+\], off|v\d+, s\[\d+:\d+\])$", t)
            if not (m and a):
                continue
            loads += 1
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            touches += [(t, tt) for tt in touches_before_wait(body, i + 1, dst)]
        # compiler waits up to the last inline-asm statement (= the last counted wait / DMA / residual load of the pipeline); what follows
        # it is the tail of the kernel (the f16x2 range flag's load + store), where a drain is harmless
        last_asm = max((i for i, (t, a) in enumerate(body) if a), default=len(body))
        cwaits = [t for i, (t, a) in enumerate(body) if not a and i < last_asm and t.startswith("s_waitcnt") and "vmcnt" in t]
        report[name] = dict(instructions=sum(not t.endswith(":") for t, _ in body), scratch=scratch, asm_loads=loads, touches=touches, compiler_vmcnt_waits=cwaits,
                            mfma=sum("v_mfma" in t for t, _ in body))
    return report


# Every other MFMA kernel of the library: no scratch instruction at all (a spill inside dcn_pipe / offs_conv / mlp_fused would be an
# uncounted VMEM operation in a hand-counted vmcnt protocol; anywhere else it is a performance bug).
OTHER_KERNELS = ["_ZN5ocrvi15dcn_pipe_kernel", "_ZN5ocrvi16offs_conv_kernel", "_ZN5ocrvi16mlp_fused_kernel", "_ZN5ocrvi16conv_gemm_kernel",
                 "_ZN5ocrvi14gconv32_kernel", "_ZN5ocrvi16attention_kernel", "_ZN5ocrvi18attention16_kernel", "_ZN5ocrvi16stem_pool_kernel"]


def check_duo(src="duo_f16x2.hip"):
    """The duo ring GEMM (gemm_duo.h): per kernel -- scratch instructions (must be 0), compiler-inserted vmcnt waits before the last inline-asm
    statement (at most the one behind the bias loads), and uses of M0 outside inline asm (must be 0: a DMA piece sets M0 and leaves it)."""
    rep = {}
    for name, body in kernel_bodies(asm_of(src), "_ZN5ocrvi15gemm_duo_kernel").items():
        last_asm = max((i for i, (t, a) in enumerate(body) if a), default=len(body))
        rep[name] = dict(scratch=sum("scratch_" in t for t, _ in body), mfma=sum("v_mfma" in t for t, _ in body),
                         compiler_vmcnt_waits=[t for i, (t, a) in enumerate(body) if not a and i < last_asm and t.startswith("s_waitcnt") and "vmcnt" in t],
                         m0_uses=[t for t, a in body if not a and re.search(r"\bm0\b", t)])
    return rep


def check_mlp_x2(src="mlp_x2.hip"):
    """The f16x2 fused MLP (mlp_x2.hip) counts the vmcnt of its weight ring by hand.  name -> (scratch instructions inside the chunk loop -- loop
    depth >= 2, where a spill would sit between the counted waits --, scratch instructions elsewhere, MFMA instructions).  The 4-wave D = 384
    build uses all 512 registers and spills a few loop-invariant values OUTSIDE the chunk loop (tile prologue / epilogue: extra VMEM operations
    there only make the next counted wait stricter); the 8-wave builds must not spill at all."""
    lines = asm_of(src).split("\n")
    rep = {}
    for i, l in enumerate(lines):
        m = re.match(r"(_ZN5ocrvi13mlp_x2_kernel\w*):", l)
        if not m:
            continue
        depth, inner, outer, mfma = 0, 0, 0, 0
        for t in lines[i + 1:]:
            if t.strip().startswith(".Lfunc_end"):
                break
            if re.match(r"\s*(\.LBB\w+:|; %bb\.\d+:)", t):
                d = re.search(r"Depth=(\d+)", t)
                depth = int(d.group(1)) if d else 0
            body = t.split(";")[0]
            if "scratch_" in body:
                if depth >= 2:
                    inner += 1
                else:
                    outer += 1
            mfma += "v_mfma" in body
        rep[m.group(1)] = (inner, outer, mfma)
    return rep


def check_scratch(src):
    """name -> (scratch instructions, MFMA instructions) for every non-ring MFMA kernel in the translation unit"""
    txt = asm_of(src)
    rep = {}
    for prefix in OTHER_KERNELS:
        for name, body in kernel_bodies(txt, prefix).items():
            rep[name] = (sum("scratch_" in t for t, _ in body), sum("v_mfma" in t for t, _ in body))
    return rep


if __name__ == "__main__":
    ok = True
    for src in sys.argv[1:] or ["conv_bf16.hip"]:
        for name, r in check(src).items():
            good = r["scratch"] == 0 and not r["touches"] and len(r["compiler_vmcnt_waits"]) == 1
            ok &= good
            print(("ok   " if good else "FAIL ") + name, {k: (len(v) if isinstance(v, list) else v) for k, v in r.items()})
    sys.exit(0 if ok else 1)
