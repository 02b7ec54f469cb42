// Persistent LDS-DMA ring GEMM for the pure-GEMM members of the conv family: stride-1 1x1 convolutions and every nn.Linear
// (out[m, n] = act(sum_k A[m, k] * Wt[n, k] + bias[n] (+ res)),  A row-major [M][lda], K % (128 bytes) == 0).
//
// Why a second kernel: conv_gemm stages operands through registers, so its prefetch depth is one K-step and a tile's prologue and
// epilogue latency is only hidden by other workgroups.  Here operands go global -> LDS directly (global_load_lds_dwordx4, no staging
// VGPRs) into a 3-stage ring that keeps streaming ACROSS tile boundaries: the next tile's first two K-steps are in flight while the
// previous tile's epilogue runs.  One persistent workgroup per CU: a 256x128 tile on 4 waves (2 x 2, one per SIMD, 128x64 each: the
// wave then owns the SIMD's whole 512-entry register file, which the parked accumulators of the deferred epilogue need), or a
// 128x128 tile on 8 waves (4 x 2) for short M, or a 256x64 tile (8 x 1) for 64-channel layers.
//
// Protocol per K-step s (slot = s % 3), every wave:
//   s_waitcnt vmcnt(N)   own DMAs of stage s have landed        (N counts exactly the younger VMEM ops: stage s+1's DMAs and, right
//   s_barrier            => everybody's have; everybody is also  after an epilogue, its stores -- which are made unconditional via a
//                           done reading slot (s-1) % 3          dump page so the count is exact; VMEM ops retire in issue order)
//   issue stage s+2      into slot (s+2) % 3 == (s-1) % 3
//   ds_read + MFMA       on slot s % 3
// The DMA is issued from inline asm so the compiler's own wait-count bookkeeping never sees it (with the builtin it drains vmcnt(0)
// before every ds_read, which serialises the ring); every asm statement carries a "memory" clobber, so LDS reads cannot move across
// the wait+barrier.  LDS image per stage = the same [rows][128 B] XOR-swizzled tiles conv_gemm uses; the swizzle is applied on the
// per-lane SOURCE address because a DMA instruction writes 64 lanes x 16 B linearly (cdna_hip_programming.md rule 21).
#pragma once
#include "conv_gemm.h"

namespace ocrvi {

// 16 bytes per lane from (uniform base + per-lane 32-bit offset) to LDS at lds_dst + lane * 16
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// same, L1-bypassing (sc1: served by the XCD's L2): for operands every CU streams once per step, so they do not evict the lines a
// gather wants to find in the 32-KiB vector L1
__device__ __forceinline__ void glds16_sc1(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// same with a full per-lane 64-bit address (3x3 mode: border lanes are redirected to the zero page)
__device__ __forceinline__ void glds16v(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {  // tell the compiler the pointer is wave-uniform
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16(u32x4& dst, const void* gsrc) {  // asynchronous: dst is valid only after wait_vm_only + bind16
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(gsrc) : "memory");
}
// same from (wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset): one VALU op per address instead of a 64-bit add chain
__device__ __forceinline__ void gload16s(u32x4& dst, const char* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_only() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bind16(u32x4& r) { asm volatile("" : "+v"(r)); }  // every later use of r is ordered after this point
// (development: -DOCRVI_TIMING_RING_NOLDS replaces the fragment reads of the f16x2 ring by undefined registers -- wrong results, timing only)
__device__ __forceinline__ uint4 lds16(const char* p) {
#ifdef OCRVI_TIMING_RING_NOLDS
    u32x4 v;
    asm volatile("" : "=v"(v));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *(const uint4*)p;
#endif
}

// Epilogue layout.  fp32 output: MFMA block a is channels 16a .. 16a+15, lane (lr, g) holds 4 consecutive ones -> a 16-byte access
// per lane, 64 contiguous bytes per pixel row per instruction.  16-bit output: the weight fragment of block a reads tile row
// 32*(a>>1) + 8*(lr>>2) + 4*(a&1) + (lr&3) of the wave's 64 instead (a permutation of the output channels, conflict-free under the
// same XOR swizzle), so lane (lr, g) ends up with channels [8g, 8g+8) and [32+8g, 32+8g+8): again 16 bytes per lane and 64
// contiguous bytes per row per instruction.  A residual has the output's element type (checked by gemm_ring_eligible).
//
// Deferred epilogue.  The kernel is bound by the latency of its operand stream, not by MFMA issue, so a tile's epilogue (bias,
// activation, residual, stores) is not run at the tile boundary but in NGRP = MI / SPS slice groups of SPS 16-row slices, one group
// in each of the first NGRP K-steps of the NEXT tile -- in the shadow of DMAs that are in flight anyway.  Tiles alternate between
// two accumulator sets (the tile loop is unrolled by two and the first NGRP K-steps are peeled), so every register index is static
// and nothing is copied.  A workgroup keeps one column tile (nt) for its whole life, so its bias lives in registers.  Residual slices
// are fetched by inline-asm loads issued before the step's DMAs and awaited with a counted vmcnt after the step's MFMAs
// (compiler-visible loads would make hipcc wait vmcnt(0), draining the ring's DMAs: it cannot see them).  Host contract:
// gridDim.x = Gm * (Np / 128) with Gm <= ceil(M / BM), and NGRP <= Kp / BKE (a tile's groups fit into the next tile's K-steps).
//
// PROF (development only, build with -DOCRVI_RING_PROF_BUILD, run with OCRVI_RING_PROF=1): every wave accumulates shader-clock
// cycles spent in wait+barrier / issue / ds_read+MFMA / epilogue work and adds them into p.out2 (uint64[4]) at exit.
template <int I> struct IC { static constexpr int value = I; };

template <typename T, int BM, int NW, int SPS, bool F32O, bool C3 = false, int BN = 128, bool PROF = false, int ACT = -1>
// (second launch bound = waves per SIMD in HIP: NW / 4 of them share a SIMD's 512 registers)
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_ring_kernel(const ConvParams p) {
    constexpr int EPC = TypeInfo<T>::EPC, BKE = 8 * EPC;
    static_assert(BN == 128 || BN == 64, "column tile");
    constexpr int WN = BN / 64, WM = NW / WN, TM = BM / WM, TN = 64, MI = TM / 16, NI = 4;  // waves: WM along M x WN along N, 64 columns each
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int NSTAGE = 3, AHEAD = NSTAGE - 1;   // ring slots; stage s + AHEAD is issued during step s
    constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;  // 1-KiB DMA pieces per wave per stage (8 rows x 128 B each)
    constexpr int G = NA + NB;                 // VMEM ops per wave per stage
    constexpr int NGRP = MI / SPS;             // slice groups per tile
    static_assert(MI % SPS == 0, "slice group size must divide the wave's row blocks");
    constexpr int SG = F32O ? SPS * 4 : SPS * 2;  // VMEM stores per wave per slice group (unconditional)
    static_assert(G + SG < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int lr = lane & 15, g = lane >> 4;        // (not const: drain() recomputes them, see there)
    constexpr bool f32o = F32O;                                   // output (and residual) element type: fp32 or T
    static_assert(F32O || sizeof(T) == 2, "an fp32 GEMM has fp32 output");
    const bool has_res = p.res_mode != RES_NONE;                  // RES_SAME, or RES_UP2: nearest 2x upsample of a half-resolution tensor
    const int swa = swz128(lr);                                   // A rows: b*16 + lr
    // B rows of MFMA block a: 16-bit output: 32*(a>>1) + 4*(a&1) + brow (channel permutation, see above); fp32 output: 16*a + lr
    const int brow = f32o ? lr : 8 * (lr >> 2) + (lr & 3);
    const int swb = swz128(brow);                                 // (the per-a offset touches neither row bit 1 nor bit 3)
    const int foa0 = ((2 * g) ^ swa) << 4, foa1 = ((2 * g + 1) ^ swa) << 4;
    const int fob0 = ((2 * g) ^ swb) << 4, fob1 = ((2 * g + 1) ^ swb) << 4;
    const int ntiles = p.Np / BN, mtiles = (p.M + BM - 1) / BM;
    const int nk = p.Kp / BKE;
    const int Gd = gridDim.x, Gm = Gd / ntiles;
    const int wg = xcd_remap(blockIdx.x, Gd);
    const int nt = wg % ntiles, mt0 = wg / ntiles;               // this workgroup: column tile nt, row tiles mt0, mt0 + Gm, ...
    const int nb = nt * BN + wn * TN;                             // first output channel of this wave
    const char* const A = (const char*)p.x + (size_t)p.cin_off * sizeof(T);
    const int lda_b = p.Cin * (int)sizeof(T), ldw_b = p.Kp * (int)sizeof(T);

    // ---- DMA issue state.  A piece is 8 rows x 128 B; piece i of this wave covers tile rows (i * NW + wave) * 8 + prow.  Sources are
    // (uniform base of the tile row 0 at the stage's K offset) + (per-lane 32-bit offset): the XOR swizzle is applied on the source
    // chunk, whose row bits 1 and 3 do not depend on i, so B needs one offset register and A one per piece (rows past M are clamped
    // to the last valid row: what they produce is never stored).
    const int prow = lane >> 3;                                   // row inside an 8-row piece
    const int chunk = (lane & 7) ^ swz128(wave * 8 + prow);       // source chunk that must land in LDS chunk (lane & 7)
    const unsigned b_off = (unsigned)((wave * 8 + prow) * ldw_b + chunk * 16);
    const char* const b_tile = uniform_ptr((const char*)p.w + (size_t)(nt * BN) * ldw_b);
    unsigned a_off[C3 ? 1 : NA];
    const char* a_tile = nullptr;                                 // row 0 of the tile at the issue cursor
    // C3 (3x3, stride 1, pad 1, Cin % BKE == 0: K-step ks = tap (ks / cpb) x channel block (ks % cpb)): the A row of output pixel
    // m at tap (r, s) is input pixel m + (r-1) W + (s-1) -- a uniform displacement -- or zeros outside the image: per piece one
    // per-lane pointer to the centre pixel and a 9-bit mask of the taps that fall inside; invalid lanes read the zero page.
    const char* a_ptr[C3 ? NA : 1];
    unsigned a_mask[C3 ? NA : 1];
    const int cpb = C3 ? p.Cin_g / BKE : 1;                       // channel blocks per tap
    int i_tap = 0, i_cb = 0;
    int i_mt = mt0, i_ks = 0;                                     // issue cursor
    auto setup_issue = [&](int mt) {
        if constexpr (C3) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = mt * BM + (i * NW + wave) * 8 + prow;
                const bool ok = m < p.M;
                const int mm = ok ? m : 0;
                const int t = fastdiv(mm, p.mg_ow), ow = mm - t * p.OW, oh = t - fastdiv(t, p.mg_oh) * p.OH;
                a_ptr[i] = A + (size_t)mm * lda_b + chunk * 16;
                unsigned mk = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (ok && (unsigned)(oh + r - 1) < (unsigned)p.H && (unsigned)(ow + c - 1) < (unsigned)p.W) mk |= 1u << (r * 3 + c);
                a_mask[i] = mk;
            }
        } else if (p.SH != 1 || p.SW != 1) {  // strided 1x1 (ResNet downsample): input pixel (img, oh*SH, ow*SW); offsets from the tensor base
            a_tile = uniform_ptr(A);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = min(mt * BM + (i * NW + wave) * 8 + prow, p.M - 1);
                const int t = fastdiv(m, p.mg_ow), ow = m - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
                a_off[i] = (unsigned)((img * p.H + oh * p.SH) * p.W + ow * p.SW) * (unsigned)lda_b + chunk * 16;
            }
        } else {
#ifdef OCRVI_TIMING_RING_AWRAP   // (development, timing only: every row tile reads one of four -- the activations stay in L2)
            a_tile = uniform_ptr(A + (size_t)(mt & 3) * BM * lda_b);
#else
            a_tile = uniform_ptr(A + (size_t)mt * BM * lda_b);
#endif
            const int last = p.M - 1 - mt * BM;                   // last valid row of this tile
#pragma unroll
            for (int i = 0; i < NA; ++i) a_off[i] = (unsigned)(min((i * NW + wave) * 8 + prow, last) * lda_b + chunk * 16);
        }
    };
    // DMA the stage at the issue cursor into a ring slot, piece by piece: begin_issue fixes the stage's constants, issue_piece(P) sends
    // piece P of the G = NA + NB (A pieces first), end_issue advances the cursor.  The pieces of stage s + 2 are issued BETWEEN the MFMAs
    // of step s (one per weight-fragment row), not in front of them: the two waves of a SIMD run in lockstep behind the step's barrier, so
    // a block of G DMA statements (M0 save / set / restore around each) ahead of the MFMAs leaves the matrix pipe idle for its whole
    // issue time on both waves at once; spread over the MFMA gaps the scalar and VMEM issue slots are free.  (Measured neutral to +2 %;
    // the shader clock stays at 2.40 GHz in the fp32 ring, so the rest of its off-pipe time is not throttling: DESIGN.md section 5.)
    unsigned st_base = 0;
    const char *st_bk = nullptr, *st_ak = nullptr, *st_zero = nullptr;
    long long st_delta = 0;
    int st_tap = 0;
    auto begin_issue = [&](int slot) {
        st_base = lds0 + slot * STAGE;
        st_bk = b_tile + (size_t)i_ks * 128;
        if constexpr (C3) {
            const int r = i_tap / 3, c = i_tap - r * 3;
            st_delta = (long long)((r - 1) * p.W + (c - 1)) * lda_b + i_cb * 128;
            st_zero = (const char*)p.zero_page + (lane & 7) * 16;
            st_tap = i_tap;
        } else {
            st_ak = a_tile + (size_t)i_ks * 128;
        }
    };
    auto issue_piece = [&](auto P) {
        constexpr int pi = decltype(P)::value;
        if constexpr (pi < NA) {
            if constexpr (C3) {
                const char* src = ((a_mask[pi] >> st_tap) & 1u) ? a_ptr[pi] + st_delta : st_zero;
                glds16v(src, __builtin_amdgcn_readfirstlane(st_base + (pi * NW + wave) * 1024));
            } else {
                glds16(uniform_ptr(st_ak), a_off[pi], __builtin_amdgcn_readfirstlane(st_base + (pi * NW + wave) * 1024));
            }
        } else if constexpr (pi < G) {
            constexpr int i = pi - NA;
            // (uniform_ptr at the point of use: in one build hipcc otherwise carried the stage base in VGPRs across the specialised epilogue
            // bodies and could not form the asm statement's SGPR operand)
            glds16(uniform_ptr(st_bk + (size_t)(i * NW * 8) * ldw_b), b_off, __builtin_amdgcn_readfirstlane(st_base + BM * 128 + (i * NW + wave) * 1024));
        }
    };
    auto end_issue = [&]() {
        if constexpr (C3) {
            if (++i_cb == cpb) {
                i_cb = 0;
                ++i_tap;
            }
        }
        if (++i_ks == nk) {
            i_ks = 0;
            i_tap = 0;
            i_cb = 0;
            i_mt += Gm;
            if (i_mt < mtiles) setup_issue(i_mt);
        }
    };
    auto issue_stage = [&](int slot) {  // the whole stage at once (pipeline fill)
        begin_issue(slot);
        issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{}); issue_piece(IC<3>{});
        issue_piece(IC<4>{}); issue_piece(IC<5>{}); issue_piece(IC<6>{}); issue_piece(IC<7>{});
        end_issue();
    };
    static_assert(G <= 8 && G <= 2 * NI, "one DMA piece per weight-fragment row of a step");
    constexpr bool SPREAD = !(F32O && MI >= 4);

    // the activation is a template parameter (the launcher instantiates the three): no run-time selects inside the slice groups
    auto activate = [&](float (&v)[4]) {
        const int act = ACT >= 0 ? ACT : p.act;
        if (act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (act == ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
    };
    // Output stores go through a buffer descriptor: 32-bit byte offsets from the tensor base (one multiply per row instead of 64-bit
    // pointer arithmetic per fragment) and the hardware's range check drops the lanes past M or N_g -- the instruction still issues, so
    // the counted vmcnt protocol keeps its exact store count without a dump page.  (host contract: the output is < 4 GiB)
    const int osz_b = (F32O || sizeof(T) == 4) ? 4 : 2;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    auto a_row = [&](int a) { return f32o ? 16 * a : 32 * (a >> 1) + 4 * (a & 1); };
    // channel (inside the wave's 64) of acc[a][.][0] for this lane
    auto ch_of = [&](int a) { return f32o ? 16 * a + 4 * g : 32 * (a >> 1) + 8 * g + 4 * (a & 1); };

    // ---- per-workgroup constant: the bias of the workgroup's 128 channels, parked in LDS behind the ring.  In the fp32 / f16x2 builds a
    // tile's accumulators START at the bias (in the accumulator's scale: f16x2 weights carry a power-of-two factor, so the division is
    // exact) instead of at zero, so the epilogue neither holds the 16 bias values next to two accumulator sets nor adds them per fragment.
    // The 16-bit builds add it in the epilogue like every other kernel of the library: their 3x3 mode shares layers with
    // conv_gemm_kernel (by M), and the two must round identically for a page to come out the same alone and inside a batch.
    constexpr bool BIAS_FIRST = sizeof(T) == 4;
    float* const bias_s = (float*)(smem + NSTAGE * STAGE);
    if (tid < BN) {
        const float bsc = IsSplit<T>::value ? 1.f / p.wscale : 1.f;
        bias_s[tid] = (p.bias && nt * BN + tid < p.N_g) ? p.bias[nt * BN + tid] * bsc : 0.f;
    }
    __syncthreads();  // (also drains the bias loads: no VMEM op is in flight when the ring starts)

    typedef f32x4 Acc[NI][MI];
    Acc accA, accB;
    unsigned long long range_mask = 0;   // f16x2: lanes that packed a value fp16's exponent cannot carry (common.h)
    u32x4 res_r[SPS][4];  // residual of the group in flight: fp32 [j][a] = 4 floats; 16-bit [j][h] = 8 elements, h < 2

    // asm loads of slice group GRP's residual of row tile pmt: (uniform base) + 32-bit byte offset (host contract: a residual has the
    // output's geometry or a quarter of it, so it is < 4 GiB like the output).  Every lane loads; lanes past M or N_g read offset 0 --
    // what they compute is dropped by the range check of the output stores.
    const char* const res_base = uniform_ptr((const char*)p.res);
    auto load_group = [&](auto GRP, int pmt) {
#pragma unroll
        for (int j = 0; j < SPS; ++j) {
            const int m = pmt * BM + wm * TM + (decltype(GRP)::value * SPS + j) * 16 + lr;
            unsigned rrow = (unsigned)m;  // residual row of output row m
            if (p.res_mode == RES_UP2) {  // (neck.py:36-38) pixel (img, oh, ow) <- (img, oh / 2, ow / 2) of the half-resolution map
                const int mm = m < p.M ? m : 0;
                const int t = fastdiv(mm, p.mg_ow), ow = mm - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
                rrow = (unsigned)((img * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1));
            }
            const unsigned row_el = m < p.M ? rrow * (unsigned)p.ldr : 0xffffffffu;
            if constexpr (F32O) {
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int n = nb + ch_of(a);
                    const unsigned off = (row_el != 0xffffffffu && n < p.N_g) ? (row_el + (unsigned)n) * 4u : 0u;
                    gload16s(res_r[j][a], res_base, off);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = nb + 32 * h + 8 * g;
                    const unsigned off = (row_el != 0xffffffffu && n < p.N_g) ? (row_el + (unsigned)n) * 2u : 0u;
                    gload16s(res_r[j][h], res_base, off);
                }
            }
        }
    };
    auto bind_group = [&]() {
#pragma unroll
        for (int j = 0; j < SPS; ++j) {
            bind16(res_r[j][0]);
            bind16(res_r[j][1]);
            if (f32o) {
                bind16(res_r[j][2]);
                bind16(res_r[j][3]);
            }
        }
    };
    // bias / activation / residual / store of slice group GRP of the parked tile (accumulators pnd, row tile pmt).  Exactly S32 / S16
    // store instructions per wave: out-of-range lanes write to the dump page instead of being skipped.
    // The wave-uniform flags of the epilogue (residual? in which format? before or after the activation? output format?) are resolved ONCE per
    // slice group: run_group picks one of the specialised bodies below, inside which nothing branches (as run-time tests per fragment they
    // were ~10 scalar branches around ~30 vector instructions).
    const bool out_raw32 = !IsSplit<T>::value || p.out_f32;
    const int res_kind = !has_res ? 0 : ((IsSplit<T>::value && F32O && !p.res_f32) ? 2 : 1);
    auto group_body = [&](auto GRP, Acc& pnd, int pmt, auto RES, auto POST, auto OUTF32) {
        constexpr int q0 = decltype(GRP)::value * SPS;
        // (a negative template value = "read the flag at run time": the 16-bit-output builds keep the run-time form, their GELU variants
        // have no registers to spare for several specialised bodies)
        const int RESK = decltype(RES)::value >= 0 ? decltype(RES)::value : res_kind;   // 0 none, 1 raw fp32 (or T for the 16-bit builds), 2 f16x2 chunks
        const bool post = decltype(POST)::value >= 0 ? decltype(POST)::value != 0 : p.res_post != 0;   // activation before the residual add
        const bool outf32 = decltype(OUTF32)::value != 0;
        // (16-bit builds: the bias of the wave's four fragments, all LDS reads in flight together -- the operand fragments are dead
        // here, so the sixteen registers are free)
        float4 bv[NI];
        if constexpr (!BIAS_FIRST) {
#pragma unroll
            for (int a = 0; a < NI; ++a) bv[a] = *(const float4*)(bias_s + wn * TN + ch_of(a));
        }
        // one accumulator fragment (which started at the bias in the fp32 / f16x2 builds): weight scale, bias, activation (fragments are fenced with sched_barrier so that the scheduler does not
        // interleave all of a group's GELU polynomials: that costs more registers than the kernel has)
        auto frag = [&](float (&v)[4], const f32x4& c, int a) {
            v[0] = unscale<T>(c[0], p.wscale); v[1] = unscale<T>(c[1], p.wscale);
            v[2] = unscale<T>(c[2], p.wscale); v[3] = unscale<T>(c[3], p.wscale);
            if constexpr (!BIAS_FIRST) {
                v[0] += bv[a].x; v[1] += bv[a].y; v[2] += bv[a].z; v[3] += bv[a].w;
            }
            if (post) activate(v);
        };
#pragma unroll
        for (int j = 0; j < SPS; ++j) {
            const int m = pmt * BM + wm * TM + (q0 + j) * 16 + lr;
#ifdef OCRVI_TIMING_RING_OWRAP   // (development, timing only: the output rows wrap at 2048 -- the stores stay in L2)
            const unsigned row_b = m < p.M ? ((unsigned)(m & 2047) * (unsigned)p.ldo + (unsigned)(p.out_coff + nb)) * (unsigned)osz_b : OOB;
#else
            const unsigned row_b = m < p.M ? ((unsigned)m * (unsigned)p.ldo + (unsigned)(p.out_coff + nb)) * (unsigned)osz_b : OOB;
#endif
            if constexpr (F32O) {
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    float v[4];
                    frag(v, pnd[a][q0 + j], a);
                    if (IsSplit<T>::value && RESK == 2) {   // a residual in the operand format: [4 hi | 4 lo]
                        const u32x4 u = res_r[j][a];
                        float rv[4];
                        Chunk<T>::unpack(make_uint4(u.x, u.y, u.z, u.w), rv);
                        v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
                    } else if (RESK == 1) {
                        const u32x4 u = res_r[j][a];
                        v[0] += __uint_as_float(u.x); v[1] += __uint_as_float(u.y);
                        v[2] += __uint_as_float(u.z); v[3] += __uint_as_float(u.w);
                    }
                    if (!post) activate(v);
                    const int c = ch_of(a);
                    const unsigned off = (row_b != OOB && nb + c < p.N_g) ? row_b + (unsigned)c * 4u : OOB;
                    u32x4 pk;
                    if (IsSplit<T>::value && !outf32) {
                        range_mask |= f16x2_out_of_range(v);   // (scalar mask: raised once, at the end of the kernel)
                        const uint4 e = Chunk<T>::pack(v);
                        pk = (u32x4){e.x, e.y, e.z, e.w};
                    } else {
                        pk = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    }
                    // (nt: an output that outweighs the operands -- N >= 2 K -- written with the default policy pushes the activation rows the other
                    // column tiles still want out of the XCD's L2; launch_gemm_ring sets the flag)
                    if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, off, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, off, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v0[4], v1[4];
                    frag(v0, pnd[2 * h][q0 + j], 2 * h);
                    frag(v1, pnd[2 * h + 1][q0 + j], 2 * h + 1);
                    if (RESK != 0) {
                        union { u32x4 u; T e[8]; } rr;
                        rr.u = res_r[j][h];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v0[r] += to_f32<T>(rr.e[r]);
                            v1[r] += to_f32<T>(rr.e[4 + r]);
                        }
                    }
                    if (!post) {
                        activate(v0);
                        activate(v1);
                    }
                    const int c = 32 * h + 8 * g;
                    union { T e[8]; u32x4 u; } pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pk.e[r] = from_f32<T>(v0[r]);
                        pk.e[4 + r] = from_f32<T>(v1[r]);
                    }
                    const unsigned off = (row_b != OOB && nb + c < p.N_g) ? row_b + (unsigned)c * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(pk.u, orsrc, off, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // (F32O builds of the 16-bit types and of fp32 write raw fp32; only f16x2 has a second 4-byte output format)
    auto run_group = [&](auto GRP, Acc& pnd, int pmt) {
        constexpr bool SPEC = F32O && !(IsSplit<T>::value && BM < 256);
        if constexpr (!SPEC) {
            // 16-bit output, and the 128-row f16x2 build: flags at run time (see group_body)
            if constexpr (IsSplit<T>::value) {
                if (out_raw32) group_body(GRP, pnd, pmt, IC<-1>{}, IC<-1>{}, IC<1>{}); else group_body(GRP, pnd, pmt, IC<-1>{}, IC<-1>{}, IC<0>{});
            } else {
                group_body(GRP, pnd, pmt, IC<-1>{}, IC<-1>{}, IC<1>{});
            }
        } else {
            auto with_out = [&](auto RES, auto POST) {
                if constexpr (IsSplit<T>::value) {
                    if (out_raw32) group_body(GRP, pnd, pmt, RES, POST, IC<1>{}); else group_body(GRP, pnd, pmt, RES, POST, IC<0>{});
                } else {
                    group_body(GRP, pnd, pmt, RES, POST, IC<1>{});
                }
            };
            if (res_kind == 0) {
                with_out(IC<0>{}, IC<0>{});                       // (without a residual the order of the activation is moot)
            } else if (res_kind == 1) {
                if (p.res_post) with_out(IC<1>{}, IC<1>{}); else with_out(IC<1>{}, IC<0>{});
            } else {
                if constexpr (IsSplit<T>::value) {
                    if (p.res_post) with_out(IC<2>{}, IC<1>{}); else with_out(IC<2>{}, IC<0>{});
                }
            }
        }
    };

    // ---- persistent loop over (row tile, k-step)
    const int my_tiles = (mtiles - mt0 + Gm - 1) / Gm;  // >= 1: Gm <= mtiles
    const int nsteps = my_tiles * nk;
    int s = 0;                 // flat step counter: stage s lives in ring slot s % NSTAGE
    bool stored = false;       // the previous step issued a slice group's stores
    long long tk[5] = {0, 0, 0, 0, 0}, t0 = 0;
    auto tick = [&](int k) {
        if constexpr (PROF) {
            const long long t = clock64();
            tk[k] += t - t0;
            t0 = t;
        }
    };
    // One K-step: MFMAs of the current tile into acc; when GRP >= 0 and a tile is parked, also slice group GRP of that tile.
    auto step = [&](Acc& acc, Acc& pnd, auto GRP, bool parked, int pmt) {
        constexpr int grp = decltype(GRP)::value;
        // Wait until stage s has landed.  N = VMEM ops issued after stage s's DMAs that may still be outstanding: stage s+1's DMAs and
        // the previous step's stores (its residual loads were consumed, hence complete; VMEM ops retire in issue order).
        auto wait_stage = [&](auto NN) {
            constexpr int N = decltype(NN)::value;
#ifdef OCRVI_TIMING_RING_NOBAR   // (development: no barrier -- races, timing only)
            wait_vm_only<N>();
#else
            if constexpr (PROF) {   // the wave's own DMA pieces (slot 4 of the report), then the workgroup (slot 0)
                wait_vm_only<N>();
                tick(4);
                asm volatile("s_barrier" ::: "memory");
            } else {
                wait_vm_barrier<N>();
            }
#endif
        };
        if (AHEAD > 1 && s + 1 < nsteps) {
            if (!stored) wait_stage(IC<G>{}); else wait_stage(IC<G + SG>{});
        } else {
            if (!stored) wait_stage(IC<0>{}); else wait_stage(IC<SG>{});
        }
        tick(0);
        bool run = false;
        if constexpr (grp >= 0) {
            run = parked;
            if (run && has_res) load_group(GRP, pmt);
        }
        const bool dma = s + AHEAD < nsteps;
        if (dma) begin_issue((s + AHEAD) % NSTAGE);
        tick(1);
        const char* As = smem + (s % NSTAGE) * STAGE;
        const char* Bs = As + BM * 128;
        auto half = [&](auto H) {
            constexpr int h = decltype(H)::value;
            const int foa = h == 0 ? foa0 : foa1, fob = h == 0 ? fob0 : fob1;
            uint4 xf[MI], wf[NI];
#pragma unroll
            for (int b = 0; b < MI; ++b) xf[b] = *(const uint4*)(As + (wm * TM + b * 16 + lr) * 128 + foa);
#pragma unroll
            for (int a = 0; a < NI; ++a) wf[a] = *(const uint4*)(Bs + (wn * TN + a_row(a) + brow) * 128 + fob);
            auto row = [&](auto A) {
                constexpr int a = decltype(A)::value;
#pragma unroll
                for (int b = 0; b < MI; ++b) Mma<T>::half(wf[a], xf[b], acc[a][b]);
                if constexpr (SPREAD && h * NI + a < G) {   // piece h NI + a of stage s + 2 rides in the shadow of this row's MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    if (dma) issue_piece(IC<h * NI + a>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            row(IC<0>{}); row(IC<1>{}); row(IC<2>{}); row(IC<3>{});
            static_assert(NI == 4, "four weight-fragment rows per wave");
            // with an fp32 residual slice in flight the second half's fragments must not be hoisted over the first half's MFMAs:
            // the kernel sits at the 256-VGPR limit (the other wave of the SIMD covers the exposed LDS latency)
            if constexpr (F32O && MI >= 4) __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (!SPREAD) {   // (the fp32-output 256-row build sits at the 256-VGPR limit: the scheduling fences would make it spill)
            if (dma) {
                issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{}); issue_piece(IC<3>{});
                issue_piece(IC<4>{}); issue_piece(IC<5>{}); issue_piece(IC<6>{}); issue_piece(IC<7>{});
            }
        }
        if constexpr (IsSplit<T>::value) {
            // f16x2: both chunks of every fragment at once, regrouped into (hi, lo) quartets: three MFMAs per fragment pair (Mma<f16x2_t>)
            typedef typename Mma<T>::u4v U;
            // (row blocks in groups of XB: with all MI = 4 blocks' fragments live next to two accumulator sets the 256-row build spills; the
            // weight fragments are then read from LDS once per group, i.e. twice)
            constexpr int XB = MI >= 4 ? 2 : MI;
            // weight fragments are fetched one row AHEAD of the MFMAs that use them (raw chunks: 8 registers), so that a row's LDS latency and
            // regrouping moves fall under the previous row's MFMAs instead of in front of its own
            auto wread = [&](int a, uint4& c0, uint4& c1) {
                const char* r = Bs + (wn * TN + a_row(a) + brow) * 128;
                c0 = lds16(r + fob0);
                c1 = lds16(r + fob1);
            };
            auto sgroup = [&](auto B0) {
                constexpr int b0 = decltype(B0)::value;
                uint4 n0, n1;
                wread(0, n0, n1);
                U xH[XB], xL[XB];
#pragma unroll
                for (int b = 0; b < XB; ++b) {
                    const char* r = As + (wm * TM + (b0 + b) * 16 + lr) * 128;
                    Mma<T>::regroup(lds16(r + foa0), lds16(r + foa1), xH[b], xL[b]);
                }
                auto srow = [&](auto A) {
                    constexpr int a = decltype(A)::value;
                    U wH, wL;
                    Mma<T>::regroup(n0, n1, wH, wL);
                    if constexpr (a + 1 < NI) wread(a + 1, n0, n1);
#pragma unroll
                    for (int b = 0; b < XB; ++b) Mma<T>::three(wH, wL, xH[b], xL[b], acc[a][b0 + b]);
                    if constexpr (SPREAD && b0 == 0) {   // pieces 2a, 2a + 1 of stage s + 2 ride in the shadow of this row's MFMAs
                        __builtin_amdgcn_sched_barrier(0);
                        if (dma) {
                            issue_piece(IC<2 * a>{});
                            issue_piece(IC<2 * a + 1>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                srow(IC<0>{}); srow(IC<1>{}); srow(IC<2>{}); srow(IC<3>{});
            };
            if constexpr (MI > XB) {
                // steady step (no parked tile: its accumulator set and the epilogue's registers are dead here): every fragment of the step is
                // requested up front -- 16 reads, the weights once -- so one LDS latency is exposed per step instead of one per group plus a
                // partial one per weight row
                uint4 n0, n1, xr[MI][2];
                wread(0, n0, n1);
#pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const char* r = As + (wm * TM + b * 16 + lr) * 128;
                    xr[b][0] = lds16(r + foa0);
                    xr[b][1] = lds16(r + foa1);
                }
                __builtin_amdgcn_sched_barrier(0);
                U xH[MI], xL[MI];
#pragma unroll
                for (int b = 0; b < MI; ++b) Mma<T>::regroup(xr[b][0], xr[b][1], xH[b], xL[b]);
                // weight rows one row = 12 MFMAs ahead, requested BEFORE the row's MFMAs (fenced: hipcc otherwise sinks the reads to a few MFMAs
                // before their use and waits on them at once).  (Round 4 tried regrouping each pixel fragment in place -- two inline-asm v_swap_b32 --
                // right where weight row 0 first uses it: neutral in time, and an inline-asm write directly in front of the MFMA that reads it is
                // outside hipcc's wait-state bookkeeping -- the same construction fed conv_gemm stale registers.  Reverted.)
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    U wH, wL;
                    Mma<T>::regroup(n0, n1, wH, wL);
                    if (a + 1 < NI) wread(a + 1, n0, n1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < MI; ++b) Mma<T>::three(wH, wL, xH[b], xL[b], acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                sgroup(IC<0>{});
            }
        } else {
            half(IC<0>{});
            half(IC<1>{});
        }
        if (dma) end_issue();
        tick(2);
        if constexpr (grp >= 0) {
            if (run) {
                if (has_res) {  // the group's loads are older than this step's DMAs only
                    if (dma) wait_vm_only<G>(); else wait_vm_only<0>();
                    bind_group();
                }
                run_group(GRP, pnd, pmt);
            }
        }
        stored = run;
        ++s;
        tick(3);
    };
    // One tile into acc; the parked tile's NGRP slice groups ride on its first NGRP K-steps (NGRP <= nk by the host contract).
    auto tile = [&](Acc& acc, Acc& pnd, bool parked, int pmt) {
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (BIAS_FIRST) bv = *(const float4*)(bias_s + wn * TN + ch_of(a));
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){bv.x, bv.y, bv.z, bv.w};
        }
        if constexpr (NGRP >= 1) step(acc, pnd, IC<0>{}, parked, pmt);
        if constexpr (NGRP >= 2) step(acc, pnd, IC<1>{}, parked, pmt);
        if constexpr (NGRP >= 3) step(acc, pnd, IC<2>{}, parked, pmt);
        if constexpr (NGRP >= 4) step(acc, pnd, IC<3>{}, parked, pmt);
        static_assert(NGRP <= 4, "peel more steps");
        for (int ks = NGRP; ks < nk; ++ks) step(acc, pnd, IC<-1>{}, false, pmt);
    };
    auto drain = [&](Acc& pnd, int pmt) {  // the last tile's epilogue (no DMA is in flight any more)
        {   // the lane's fragment coordinates are needed again only here: recomputed from an opaque copy of the thread index, so that the
            // register allocator does not carry (or spill) them through the whole ring loop for this one use
            int l = threadIdx.x;
            asm volatile("" : "+v"(l));
            lr = l & 15;
            g = (l & 63) >> 4;
        }
        auto one = [&](auto GRP) {
            if (has_res) {
                load_group(GRP, pmt);
                wait_vm_only<0>();
                bind_group();
            }
            run_group(GRP, pnd, pmt);
        };
        if constexpr (NGRP >= 1) one(IC<0>{});
        if constexpr (NGRP >= 2) one(IC<1>{});
        if constexpr (NGRP >= 3) one(IC<2>{});
        if constexpr (NGRP >= 4) one(IC<3>{});
    };

    setup_issue(i_mt);
    issue_stage(0);
    if (AHEAD > 1 && nsteps > 1) issue_stage(1);
    if constexpr (PROF) t0 = clock64();
    int mt = mt0;
    bool parked = false;  // a finished tile sits in the other accumulator set
    for (int t = 0; t < my_tiles; t += 2) {
        tile(accA, accB, parked, mt - Gm);
        if (t + 1 == my_tiles) {
            drain(accA, mt);
            break;
        }
        tile(accB, accA, true, mt);
        mt += 2 * Gm;
        parked = true;
        if (t + 2 >= my_tiles) drain(accB, mt - Gm);
    }
    tick(3);
    if constexpr (IsSplit<T>::value) f16x2_raise(range_mask);
    if constexpr (PROF) {
        if (lane == 0)
            for (int k = 0; k < 5; ++k) atomicAdd((unsigned long long*)p.out2 + k, (unsigned long long)tk[k]);
    }
}

template <typename T> int launch_gemm_ring(const ConvParams& p, int amode, hipStream_t stream);
// true when (p, amode) is a pure GEMM this kernel handles
bool gemm_ring_eligible(const ConvParams& p, int amode, int dtype);
int ring_pages(const void** zero_page, void** dump_page);  // device scratch pages (allocated once per process)

}  // namespace ocrvi
