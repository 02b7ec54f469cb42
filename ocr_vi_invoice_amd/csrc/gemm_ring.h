// Persistent LDS-DMA ring GEMM for the pure-GEMM members of the conv family: stride-1 1x1 convolutions and every nn.Linear
// (out[m, n] = act(sum_k A[m, k] * Wt[n, k] + bias[n] (+ res)),  A row-major [M][lda], K % (128 bytes) == 0).
//
// Why a second kernel: conv_gemm stages operands through registers, so its prefetch depth is one K-step and a tile's prologue and
// epilogue latency is only hidden by other workgroups.  Here operands go global -> LDS directly (global_load_lds_dwordx4, no staging
// VGPRs) into a 3-stage ring that keeps streaming ACROSS tile boundaries: the next tile's first two K-steps are in flight while the
// current tile's epilogue runs.  256x128 (or 128x128) tile, 8 waves (4 along M x 2 along N), one workgroup per CU, persistent.
//
// Protocol per K-step s (slot = s % 3), every wave:
//   s_waitcnt vmcnt(N)   own DMAs of stage s have landed        (N counts exactly the younger VMEM ops: stage s+1's DMAs and, right
//   s_barrier            => everybody's have; everybody is also  after an epilogue, its stores -- which are made unconditional via a
//                           done reading slot (s-1) % 3          dump page so the count is exact)
//   issue stage s+2      into slot (s+2) % 3 == (s-1) % 3
//   ds_read + MFMA       on slot s % 3
// The DMA is issued from inline asm so the compiler's own wait-count bookkeeping never sees it (with the builtin it drains vmcnt(0)
// before every ds_read, which serialises the ring); every asm statement carries a "memory" clobber, so LDS reads cannot move across
// the wait+barrier.  LDS image per stage = the same [rows][128 B] XOR-swizzled tiles conv_gemm uses; the swizzle is applied on the
// per-lane SOURCE address because a DMA instruction writes 64 lanes x 16 B linearly (cdna_hip_programming.md rule 21).
#pragma once
#include "conv_gemm.h"

namespace ocrvi {

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <typename T, int BM>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(const ConvParams p) {
    constexpr int EPC = TypeInfo<T>::EPC, BKE = 8 * EPC;
    constexpr int BN = 128, TM = BM / 4, TN = 64, MI = TM / 16, NI = 4;
    constexpr int STAGE = (BM + BN) * 128, NSTAGE = 3;
    constexpr int NA = BM / 64, NB = BN / 64;  // 1-KiB DMA pieces per wave per stage (8 rows x 128 B each)
    constexpr int G = NA + NB;                 // VMEM ops per wave per stage
    constexpr int E = MI * NI;                 // VMEM ops per wave per epilogue (unconditional stores)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    const int sw = swz128(lr);
    const int fo0 = ((2 * g) ^ sw) << 4, fo1 = ((2 * g + 1) ^ sw) << 4;
    const int ntiles = p.Np / BN;
    const int total = ((p.M + BM - 1) / BM) * ntiles;
    const int nk = p.Kp / BKE;
    const int Gd = gridDim.x;
    const char* const A = (const char*)p.x + (size_t)p.cin_off * sizeof(T);
    const char* const Wt = (const char*)p.w;
    const size_t lda_b = (size_t)p.Cin * sizeof(T), ldw_b = (size_t)p.Kp * sizeof(T);

    // ---- DMA issue state: this lane's source pointers for its NA + NB pieces of the stage at the issue cursor
    const int prow = lane >> 3;                                   // row inside an 8-row piece
    const char* a_src[NA];
    unsigned a_step[NA];                                          // 128, or 0 for rows past M (they read the zero page)
    const char* b_src[NB];
    int i_tile = xcd_remap(blockIdx.x, Gd), i_ks = 0;             // issue cursor
    auto setup_issue = [&](int tile) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (i * 8 + wave) * 8 + prow;            // row inside the A tile
            const int m = mt * BM + row;
            const int chunk = (lane & 7) ^ swz128(row);           // source chunk that must land in LDS chunk (lane & 7)
            const bool ok = m < p.M;
            a_src[i] = ok ? A + (size_t)m * lda_b + chunk * 16 : (const char*)p.zero_page + (lane & 7) * 16;
            a_step[i] = ok ? 128u : 0u;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (i * 8 + wave) * 8 + prow;
            const int chunk = (lane & 7) ^ swz128(row);
            b_src[i] = Wt + (size_t)(nt * BN + row) * ldw_b + chunk * 16;
        }
    };
    auto issue_stage = [&](int slot) {  // DMA the stage at the issue cursor into ring slot `slot`, advance the cursor
        const unsigned base = lds0 + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            glds16(a_src[i], __builtin_amdgcn_readfirstlane(base + (i * 8 + wave) * 1024));
            a_src[i] += a_step[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            glds16(b_src[i], __builtin_amdgcn_readfirstlane(base + BM * 128 + (i * 8 + wave) * 1024));
            b_src[i] += 128;
        }
        if (++i_ks == nk) {
            i_ks = 0;
            i_tile += Gd;
            if (i_tile < total) setup_issue(i_tile);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto activate = [&](float (&v)[4]) {
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
    };
    // Exactly E = MI*NI store instructions per wave: out-of-range lanes write to the dump page instead of being skipped.
    auto epilogue = [&](int tile) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const bool f32o = p.out_f32 || sizeof(T) == 4;
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = mt * BM + wm * TM + b * 16 + lr;
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int n = nt * BN + wn * TN + a * 16 + 4 * g;
                const bool ok = m < p.M && n < p.N_g;
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias && ok) {
                    const float4 bv = *(const float4*)(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (p.res_post) activate(v);
                if (p.res_mode == RES_SAME && ok) {
                    const size_t ro = (size_t)m * p.ldr + n;
                    if (p.res_f32 || sizeof(T) == 4) {
                        const float4 rv = *(const float4*)((const float*)p.res + ro);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    } else {
                        union { uint2 u; T h[4]; } rr;
                        rr.u = *(const uint2*)((const T*)p.res + ro);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rr.h[r]);
                    }
                }
                if (!p.res_post) activate(v);
                const size_t oo = (size_t)m * p.ldo + p.out_coff + n;
                if (f32o) {
                    float* o = ok ? (float*)p.out + oo : (float*)p.dump_page + lane * 4;
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    union { T h[4]; uint2 u; } pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk.h[r] = from_f32<T>(v[r]);
                    T* o = ok ? (T*)p.out + oo : (T*)p.dump_page + lane * 4;
                    *(uint2*)o = pk.u;
                }
            }
        }
    };

    // ---- persistent flat loop over (tile, k-step)
    int c_tile = i_tile, c_ks = 0;  // compute cursor
    if (c_tile >= total) return;
    const int my_tiles = (total - c_tile + Gd - 1) / Gd;
    const int nsteps = my_tiles * nk;
    setup_issue(i_tile);
    issue_stage(0);
    if (nsteps > 1) issue_stage(1);
    bool after_epi = false;
    for (int s = 0; s < nsteps; ++s) {
        const bool younger = s + 1 < nsteps;  // stage s+1 is in flight behind stage s
        if (younger) {
            if (after_epi) wait_vm_barrier<G + E>(); else wait_vm_barrier<G>();
        } else {
            if (after_epi) wait_vm_barrier<E>(); else wait_vm_barrier<0>();
        }
        after_epi = false;
        if (s + 2 < nsteps) issue_stage((s + 2) % NSTAGE);
        const char* As = smem + (s % NSTAGE) * STAGE;
        const char* Bs = As + BM * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fo = h == 0 ? fo0 : fo1;
            uint4 xf[MI], wf[NI];
#pragma unroll
            for (int b = 0; b < MI; ++b) xf[b] = *(const uint4*)(As + (wm * TM + b * 16 + lr) * 128 + fo);
#pragma unroll
            for (int a = 0; a < NI; ++a) wf[a] = *(const uint4*)(Bs + (wn * TN + a * 16 + lr) * 128 + fo);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b) Mma<T>::half(wf[a], xf[b], acc[a][b]);
        }
        if (++c_ks == nk) {
            epilogue(c_tile);
            after_epi = true;
            c_ks = 0;
            c_tile += Gd;
        }
    }
}

template <typename T> int launch_gemm_ring(const ConvParams& p, hipStream_t stream);
// true when (p, amode) is a pure GEMM this kernel handles
bool gemm_ring_eligible(const ConvParams& p, int amode, int dtype);
int ring_pages(const void** zero_page, void** dump_page);  // device scratch pages (allocated once per process)

}  // namespace ocrvi
