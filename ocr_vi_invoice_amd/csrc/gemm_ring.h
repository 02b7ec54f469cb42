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

// Epilogue layout.  fp32 output: MFMA block a is channels 16a .. 16a+15, lane (lr, g) holds 4 consecutive ones -> a 16-byte access
// per lane, 64 contiguous bytes per pixel row per instruction.  16-bit output: the weight fragment of block a reads tile row
// 32*(a>>1) + 8*(lr>>2) + 4*(a&1) + (lr&3) of the wave's 64 instead (a permutation of the output channels, conflict-free under the
// same XOR swizzle), so lane (lr, g) ends up with channels [8g, 8g+8) and [32+8g, 32+8g+8): again 16 bytes per lane and 64
// contiguous bytes per row per instruction.  A residual has the output's element type (checked by gemm_ring_eligible).  Bias and residual are fetched with plain loads at the top of the tile's LAST K-step (all issued back to back, after
// that step's DMAs) and consumed after its MFMAs: one exposed memory round trip per tile at most, instead of one per fragment.
//
// PROF (development only, OCRVI_RING_PROF=1): every wave accumulates shader-clock cycles spent in wait+barrier / DMA issue /
// ds_read+MFMA / epilogue and adds them into p.out2 (uint64[4]) at exit.
template <typename T, int BM, bool PROF = false>
__global__ __launch_bounds__(512, 1) void gemm_ring_kernel(const ConvParams p) {
    constexpr int EPC = TypeInfo<T>::EPC, BKE = 8 * EPC;
    constexpr int BN = 128, TM = BM / 4, TN = 64, MI = TM / 16, NI = 4;
    constexpr int STAGE = (BM + BN) * 128, NSTAGE = 3;
    constexpr int NA = BM / 64, NB = BN / 64;  // 1-KiB DMA pieces per wave per stage (8 rows x 128 B each)
    constexpr int G = NA + NB;                 // VMEM ops per wave per stage
    constexpr int E32 = MI * 4, E16 = MI * 2;  // VMEM stores per wave per epilogue (unconditional; fp32 / 16-bit output)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    const bool f32o = p.out_f32 || sizeof(T) == 4;                // output (and residual) element type: fp32 or T
    const int swa = swz128(lr);                                   // A rows: b*16 + lr
    // B rows of MFMA block a: 16-bit output: 32*(a>>1) + 4*(a&1) + brow (channel permutation, see above); fp32 output: 16*a + lr
    const int brow = f32o ? lr : 8 * (lr >> 2) + (lr & 3);
    const int swb = swz128(brow);                                 // (the per-a offset touches neither row bit 1 nor bit 3)
    const int foa0 = ((2 * g) ^ swa) << 4, foa1 = ((2 * g + 1) ^ swa) << 4;
    const int fob0 = ((2 * g) ^ swb) << 4, fob1 = ((2 * g + 1) ^ swb) << 4;
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
    // channel (inside the wave's 64) of acc[a][.][0] for this lane
    auto a_row = [&](int a) { return f32o ? 16 * a : 32 * (a >> 1) + 4 * (a & 1); };
    auto ch_of = [&](int a) { return f32o ? 16 * a + 4 * g : 32 * (a >> 1) + 8 * g + 4 * (a & 1); };
    // ---- epilogue operands, fetched during the tile's last K-step.  Every load is unconditional (out-of-range lanes read the zero
    // page) so that the compiler issues them back to back instead of branching around each one.
    float4 bias_r[NI];
    uint4 res_r[MI][4];  // fp32 residual: [b][a] = 4 floats; 16-bit residual: [b][j] = 8 elements, j < 2
    auto prefetch_epi = [&](int tile) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const int nb = nt * BN + wn * TN;
        if (p.bias) {
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int n = nb + ch_of(a);
                const float* src = n < p.N_g ? p.bias + n : (const float*)p.zero_page;
                bias_r[a] = *(const float4*)src;
            }
        }
        if (p.res_mode == RES_SAME) {
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int m = mt * BM + wm * TM + b * 16 + lr;
                if (f32o) {
#pragma unroll
                    for (int a = 0; a < NI; ++a) {
                        const int n = nb + ch_of(a);
                        const float* src = (m < p.M && n < p.N_g) ? (const float*)p.res + (size_t)m * p.ldr + n : (const float*)p.zero_page;
                        res_r[b][a] = *(const uint4*)src;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = nb + 32 * j + 8 * g;
                        const T* src = (m < p.M && n < p.N_g) ? (const T*)p.res + (size_t)m * p.ldr + n : (const T*)p.zero_page;
                        res_r[b][j] = *(const uint4*)src;
                    }
                }
            }
        }
    };
    // Exactly E32 / E16 store instructions per wave: out-of-range lanes write to the dump page instead of being skipped.
    auto epilogue = [&](int tile) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const int nb = nt * BN + wn * TN;
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = mt * BM + wm * TM + b * 16 + lr;
            float v[NI][4];
#pragma unroll
            for (int a = 0; a < NI; ++a) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[a][r] = acc[a][b][r];
                acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    v[a][0] += bias_r[a].x; v[a][1] += bias_r[a].y; v[a][2] += bias_r[a].z; v[a][3] += bias_r[a].w;
                }
                if (p.res_post) activate(v[a]);
            }
            if (p.res_mode == RES_SAME) {
                if (f32o) {
#pragma unroll
                    for (int a = 0; a < NI; ++a) {
                        const uint4 u = res_r[b][a];
                        v[a][0] += __uint_as_float(u.x); v[a][1] += __uint_as_float(u.y);
                        v[a][2] += __uint_as_float(u.z); v[a][3] += __uint_as_float(u.w);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        union { uint4 u; T h[8]; } rr;
                        rr.u = res_r[b][j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[2 * j][r] += to_f32<T>(rr.h[r]);
                            v[2 * j + 1][r] += to_f32<T>(rr.h[4 + r]);
                        }
                    }
                }
            }
            if (!p.res_post) {
#pragma unroll
                for (int a = 0; a < NI; ++a) activate(v[a]);
            }
            const size_t row_off = (size_t)m * p.ldo + p.out_coff + nb;
            if (f32o) {
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int c = ch_of(a);
                    float* o = (m < p.M && nb + c < p.N_g) ? (float*)p.out + row_off + c : (float*)p.dump_page + lane * 4;
                    *(float4*)o = make_float4(v[a][0], v[a][1], v[a][2], v[a][3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = 32 * j + 8 * g;
                    union { T h[8]; uint4 u; } pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pk.h[r] = from_f32<T>(v[2 * j][r]);
                        pk.h[4 + r] = from_f32<T>(v[2 * j + 1][r]);
                    }
                    T* o = (m < p.M && nb + c < p.N_g) ? (T*)p.out + row_off + c : (T*)p.dump_page + lane * 8;
                    *(uint4*)o = pk.u;
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
    long long tk[4] = {0, 0, 0, 0}, t0 = 0;
    auto tick = [&](int k) {
        if constexpr (PROF) {
            const long long t = clock64();
            tk[k] += t - t0;
            t0 = t;
        }
    };
    if constexpr (PROF) t0 = clock64();
    for (int s = 0; s < nsteps; ++s) {
        // Wait until stage s has landed.  N = the VMEM ops issued after stage s's DMAs that may still be outstanding: stage s+1's DMAs
        // and, right after an epilogue, its stores (the epilogue's loads were consumed, hence complete).
        const bool younger = s + 1 < nsteps;
        if (younger) {
            if (!after_epi) wait_vm_barrier<G>();
            else if (f32o) wait_vm_barrier<G + E32>();
            else wait_vm_barrier<G + E16>();
        } else {
            if (!after_epi) wait_vm_barrier<0>();
            else if (f32o) wait_vm_barrier<E32>();
            else wait_vm_barrier<E16>();
        }
        after_epi = false;
        tick(0);
        if (s + 2 < nsteps) issue_stage((s + 2) % NSTAGE);
        const bool last_k = c_ks + 1 == nk;
        if (last_k) prefetch_epi(c_tile);
        tick(1);
        const char* As = smem + (s % NSTAGE) * STAGE;
        const char* Bs = As + BM * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int foa = h == 0 ? foa0 : foa1, fob = h == 0 ? fob0 : fob1;
            uint4 xf[MI], wf[NI];
#pragma unroll
            for (int b = 0; b < MI; ++b) xf[b] = *(const uint4*)(As + (wm * TM + b * 16 + lr) * 128 + foa);
#pragma unroll
            for (int a = 0; a < NI; ++a) wf[a] = *(const uint4*)(Bs + (wn * TN + a_row(a) + brow) * 128 + fob);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b) Mma<T>::half(wf[a], xf[b], acc[a][b]);
        }
        tick(2);
        if (last_k) {
            epilogue(c_tile);
            after_epi = true;
            c_ks = 0;
            c_tile += Gd;
            tick(3);
        } else {
            ++c_ks;
        }
    }
    if constexpr (PROF) {
        if (lane == 0)
            for (int k = 0; k < 4; ++k) atomicAdd((unsigned long long*)p.out2 + k, (unsigned long long)tk[k]);
    }
}

template <typename T> int launch_gemm_ring(const ConvParams& p, hipStream_t stream);
// true when (p, amode) is a pure GEMM this kernel handles
bool gemm_ring_eligible(const ConvParams& p, int amode, int dtype);
int ring_pages(const void** zero_page, void** dump_page);  // device scratch pages (allocated once per process)

}  // namespace ocrvi
