// "Duo" ring GEMM for the 4-byte operand types (f16x2, fp32): the persistent LDS-DMA ring of gemm_ring.h rebuilt around TWO wave groups
// per workgroup that work on different row tiles half a tile-period apart.
//
//   out[m, n] = act(sum_k A[m, k] * Wt[n, k] + bias[n] (+ res)),   K % 32 == 0 (one K-step = 128 bytes = 32 channels)
//
// Why (profiles/r03_variants.md): gemm_ring's f16x2 build keeps the matrix pipe 28 % busy.  Its eight waves run in
// lockstep -- the two waves of a SIMD stall on the same LDS reads, meet at the same barrier and run their epilogues at the same time --
// it reads 16 fragments per 48 MFMAs, needs a second (parked) accumulator set to hide the epilogue, and a 256 x 128 tile streams
// 48 KiB per 2.1 MFLOP step.  Here:
//   * a workgroup is two GROUPS of four waves (waves 0-3 and 4-7: one wave of each group per SIMD).  A group owns a 128-row x BN-column
//     tile (BN = 256 or 192: every wave 128 x 64 / 128 x 48) -- 24 fragment reads per 96 MFMAs, ONE accumulator set of 128 registers;
//   * both groups consume the SAME weight stream: the workgroup keeps one column tile for its whole life, so the weight K-steps repeat
//     with period nk, and a dot product does not care where in that cycle it starts.  Group 1 runs half a period behind group 0 and
//     simply starts (and ends) each of its tiles in the middle of the weight cycle.  The weight step is DMA'd into LDS once per step for
//     both groups: a step streams 32 + 16 + 16 KiB for 4.2 MFLOP, the byte / FLOP ratio of a 256 x 256 tile;
//   * a group's period is nk MFMA steps followed by EPI epilogue steps in which its waves convert, activate and store 32 / EPI fragments
//     each while the OTHER group's wave on the same SIMD keeps the matrix pipe busy: the epilogue (GELU: ~80 VALU instructions per
//     fragment) overlaps matrix work of another wave instead of sitting between two tiles, with no parked accumulators;
//   * one s_barrier per step for all eight waves, as in gemm_ring: wait for the own DMA pieces of the step (counted vmcnt), barrier, issue
//     the weight pieces of step s + 1 and the activation pieces of step s + 2, work.  LDS: two weight stages (always L2 hits: one step
//     of latency budget) + three activation stages per group (the stream that comes from HBM: two steps) = 2 BN 128 + 6 x 16 KiB =
//     160 KiB at BN = 256, all of the CU's LDS.  The bias lives in registers (a workgroup keeps its column tile) and the accumulators
//     start at it (as gemm_ring's 4-byte builds do: the two kernels round identically).
// The DMA / counted-vmcnt rules are gemm_ring.h's (inline-asm global_load_lds_dwordx4 the compiler cannot see, exact counts, XOR swizzle
// on the source address); tools/check_ring_isa.py checks this kernel's code objects the same way.
#pragma once
#include "gemm_ring.h"

namespace ocrvi {

struct DuoPlan {
    int gm = 1;        // row-tile lanes: workgroup (lane, nt) walks 128-row tiles lane, lane + gm, ...
    int ntiles = 1;    // column tiles of BN
    int mtiles = 1;    // ceil(M / 128)
};

template <int N> __device__ __forceinline__ void duo_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in {0, 2, 4} + {0, F}: the immediate must be exact (a larger one would not wait)
template <int F> __device__ __forceinline__ void duo_wait_dyn(int na, bool stored) {
    if (!stored) {
        if (na == 0) duo_wait_vm<0>(); else if (na == 2) duo_wait_vm<2>(); else duo_wait_vm<4>();
    } else {
        if (na == 0) duo_wait_vm<F>(); else if (na == 2) duo_wait_vm<F + 2>(); else duo_wait_vm<F + 4>();
    }
}

// RESK: 0 no residual, 1 raw fp32 residual, 2 residual in T's own format (f16x2 chunks; fp32: the same as 1).  OUTF32: raw fp32 output
// (always for T = float).  EPI: epilogue steps of a tile (32 / EPI fragments per wave and step).
template <typename T, int NI, int ACT, int RESK, bool OUTF32, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_duo_kernel(const ConvParams p, const DuoPlan plan) {
    static_assert(sizeof(T) == 4, "4-byte operand types");
    static_assert(NI == 4 || NI == 3, "64 or 48 columns per wave");
    static_assert(32 % EPI == 0, "fragments per epilogue step");
    constexpr int MI = 8, BN = 64 * NI, BSTAGE = BN * 128, ASTAGE = 128 * 128;
    constexpr int F = 32 / EPI;                       // fragments (= stores, = residual loads) per wave per epilogue step
    constexpr int BPB = F / 4 > 0 ? F / 4 : 1;        // (row blocks per epilogue step when F >= 4)
    static_assert(NI == 4 || F % NI == 0 || NI % F == 0 || true, "");
    constexpr int FR = MI * NI;                       // fragments per wave (24 at NI = 3: the last EPI steps of a 48-column build run short)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    constexpr unsigned A_BASE = 2 * BSTAGE;           // A stage (g, slot) at A_BASE + (g * 3 + slot) * ASTAGE

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    int lr = lane & 15, g = lane >> 4;
    const int swa = swz128(lr);
    const int foa0 = ((2 * g) ^ swa) << 4, foa1 = ((2 * g + 1) ^ swa) << 4;   // A rows b * 16 + lr and (fp32-output layout) B rows a * 16 + lr
    const int nk = p.Kp / 32;
    const int ntiles = plan.ntiles, Gm = plan.gm, mtiles = plan.mtiles;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % ntiles, mt0 = wg / ntiles;
    const int nb = nt * BN + wn * (16 * NI);          // first output channel of this wave
    const char* const A = (const char*)p.x + (size_t)p.cin_off * sizeof(T);
    const int lda_b = p.Cin * 4, ldw_b = p.Kp * 4;

    // ---- schedule.  The workgroup's row tiles mt0, mt0 + Gm, ... alternate between the groups: group 0 takes positions 0, 2, 4, ...
    const int n_my = mt0 < mtiles ? (mtiles - mt0 + Gm - 1) / Gm : 0;
    const int T0 = (n_my + 1) >> 1, T1 = n_my >> 1;
    const int P = nk + EPI;                            // a group's period: nk MFMA steps, EPI epilogue steps
    const int off1 = P >> 1;                           // group 1 runs this many steps behind group 0
    const int S = max(T0 * P, T1 > 0 ? off1 + T1 * P : 0);   // steps of this workgroup

    // ---- DMA sources (gemm_ring.h: piece = 8 rows x 128 B, XOR swizzle on the source chunk)
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ swz128(wave * 8 + prow);
    const unsigned b_off = (unsigned)((wave * 8 + prow) * ldw_b + chunk * 16);
    const char* const b_tile = uniform_ptr((const char*)p.w + (size_t)(nt * BN) * ldw_b);
    // issue cursor of each group's A stream: runs two steps ahead of the step counter
    unsigned a_off[2][2];
    const char* a_tile[2] = {nullptr, nullptr};
    int i_pos[2] = {2, 2 - off1};                      // position in the period of step s + 2 (negative: group 1's initial delay)
    int i_til[2] = {0, 0};                             // tile count of the group at the issue cursor
    int i_slot[2] = {0, 0};                            // A slot the next issued stage goes to (MFMA-step count mod 3)
    const int Tg[2] = {T0, T1};
    auto setup_a = [&](int gg, int til) {              // per-lane source offsets of group gg's tile number til
        const int mt = mt0 + (2 * til + gg) * Gm;
        if (p.SH != 1 || p.SW != 1) {                  // strided 1x1 (ResNet downsample): input pixel (img, oh * SH, ow * SW)
            a_tile[gg] = uniform_ptr(A);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = min(mt * 128 + (i * 8 + wave) * 8 + prow, p.M - 1);
                const int t = fastdiv(m, p.mg_ow), ow = m - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
                a_off[gg][i] = (unsigned)((img * p.H + oh * p.SH) * p.W + ow * p.SW) * (unsigned)lda_b + chunk * 16;
            }
        } else {
            a_tile[gg] = uniform_ptr(A + (size_t)mt * 128 * lda_b);
            const int last = p.M - 1 - mt * 128;       // rows past M read the last valid row (what they produce is never stored)
#pragma unroll
            for (int i = 0; i < 2; ++i) a_off[gg][i] = (unsigned)(min((i * 8 + wave) * 8 + prow, last) * lda_b + chunk * 16);
        }
    };
    int i_kb = 0;                                      // weight K-step of the B stage issued next (step s + 1), and of the A stages (s + 2)
    int i_ka = 0;
    // DMA pieces of one step, in program order: NI weight pieces of step s + 1, then two activation pieces per group that runs an MFMA
    // step at s + 2.  begin_issue fixes the step's constants and returns how many A pieces follow the B pieces.
    unsigned st_bdst = 0, st_adst[2] = {0, 0};
    const char *st_bk = nullptr, *st_ak[2] = {nullptr, nullptr};
    bool st_a[2] = {false, false};
    auto begin_issue = [&](int s) -> int {
        st_bdst = lds0 + ((s + 1) & 1) * BSTAGE;
        st_bk = b_tile + (size_t)i_kb * 128;
        int na = 0;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            st_a[gg] = i_pos[gg] >= 0 && i_pos[gg] < nk && i_til[gg] < Tg[gg];
            if (st_a[gg]) {
                st_ak[gg] = a_tile[gg] + (size_t)i_ka * 128;
                st_adst[gg] = lds0 + A_BASE + (gg * 3 + i_slot[gg]) * ASTAGE;
                na += 2;
            }
        }
        return na;
    };
    auto issue_piece = [&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        if constexpr (pi < NI) {
            glds16(uniform_ptr(st_bk + (size_t)(pi * 64) * ldw_b), b_off, __builtin_amdgcn_readfirstlane(st_bdst + (pi * 8 + wave) * 1024));
        } else if constexpr (pi < NI + 4) {
            constexpr int gg = (pi - NI) >> 1, i = (pi - NI) & 1;
            if (st_a[gg]) glds16(uniform_ptr(st_ak[gg]), a_off[gg][i], __builtin_amdgcn_readfirstlane(st_adst[gg] + (i * 8 + wave) * 1024));
        }
    };
    auto end_issue = [&]() {
        if (++i_kb == nk) i_kb = 0;
        if (++i_ka == nk) i_ka = 0;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            if (st_a[gg] && ++i_slot[gg] == 3) i_slot[gg] = 0;
            if (++i_pos[gg] == P) {
                i_pos[gg] = 0;
                if (++i_til[gg] < Tg[gg]) setup_a(gg, i_til[gg]);
            }
        }
    };
    auto issue_all = [&]() {
        issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{}); issue_piece(IC<3>{});
        issue_piece(IC<4>{}); issue_piece(IC<5>{}); issue_piece(IC<6>{}); issue_piece(IC<7>{});
    };

    // ---- epilogue constants
    auto activate = [&](float (&v)[4]) {
        if constexpr (ACT == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if constexpr (ACT == ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
    };
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const char* const res_base = uniform_ptr((const char*)p.res);
    // the bias of this lane's channels (16 a + 4 g .. + 3 of the wave's columns), in the accumulator's scale
    f32x4 bias_r[NI];
    {
        const float bsc = IsSplit<T>::value ? 1.f / p.wscale : 1.f;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = nb + 16 * a + 4 * g;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && n < p.N_g) bv = *(const float4*)(p.bias + n);
            bias_r[a] = (f32x4){bv.x * bsc, bv.y * bsc, bv.z * bsc, bv.w * bsc};
        }
    }
    __syncthreads();   // (drains the bias loads: no compiler-visible VMEM operation is in flight when the ring starts)

    f32x4 acc[NI][MI];
    u32x4 res_r[F];
    unsigned long long range_mask = 0;

    // ---- execution cursor of this wave's group
    int e_pos = grp == 0 ? 0 : -off1;                  // position in the period of step s
    int e_til = 0;
    int e_slot = 0;                                    // A slot of the group's next MFMA step
    const int myT = grp == 0 ? T0 : T1;

    // one MFMA step: all NI weight fragments up front, the eight row blocks streamed one ahead; DMA piece b rides behind row block b
    auto mfma_step = [&](const char* As, const char* Bs) {
        const char* Br = Bs + (wn * (16 * NI) + lr) * 128;
        const char* Ar = As + lr * 128;
        if constexpr (IsSplit<T>::value) {
            typedef typename Mma<T>::u4v U;
            uint4 wc[NI][2], xc[2][2];
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                wc[a][0] = lds16(Br + a * 2048 + foa0);
                wc[a][1] = lds16(Br + a * 2048 + foa1);
            }
            xc[0][0] = lds16(Ar + foa0);
            xc[0][1] = lds16(Ar + foa1);
            __builtin_amdgcn_sched_barrier(0);
            U wH[NI], wL[NI];
#pragma unroll
            for (int a = 0; a < NI; ++a) Mma<T>::regroup(wc[a][0], wc[a][1], wH[a], wL[a]);
            auto blk = [&](auto BB) {
                constexpr int b = decltype(BB)::value;
                if constexpr (b + 1 < MI) {
                    xc[(b + 1) & 1][0] = lds16(Ar + (b + 1) * 2048 + foa0);
                    xc[(b + 1) & 1][1] = lds16(Ar + (b + 1) * 2048 + foa1);
                }
                U xH, xL;
                Mma<T>::regroup(xc[b & 1][0], xc[b & 1][1], xH, xL);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NI; ++a) Mma<T>::three(wH[a], wL[a], xH, xL, acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(IC<b>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            blk(IC<0>{}); blk(IC<1>{}); blk(IC<2>{}); blk(IC<3>{}); blk(IC<4>{}); blk(IC<5>{}); blk(IC<6>{}); blk(IC<7>{});
        } else {
            uint4 wf[NI][2], xf[2][2];
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                wf[a][0] = lds16(Br + a * 2048 + foa0);
                wf[a][1] = lds16(Br + a * 2048 + foa1);
            }
            xf[0][0] = lds16(Ar + foa0);
            xf[0][1] = lds16(Ar + foa1);
            auto blk = [&](auto BB) {
                constexpr int b = decltype(BB)::value;
                if constexpr (b + 1 < MI) {
                    xf[(b + 1) & 1][0] = lds16(Ar + (b + 1) * 2048 + foa0);
                    xf[(b + 1) & 1][1] = lds16(Ar + (b + 1) * 2048 + foa1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NI; ++a) Mma<T>::run(wf[a], xf[b & 1], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(IC<b>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            blk(IC<0>{}); blk(IC<1>{}); blk(IC<2>{}); blk(IC<3>{}); blk(IC<4>{}); blk(IC<5>{}); blk(IC<6>{}); blk(IC<7>{});
        }
    };

    // fragment f of the wave (f = b * NI + a: the NI fragments of a row block are consecutive, so a step's stores cover whole 64 NI-byte
    // row segments) of row tile mt: residual load / epilogue body
    auto res_off = [&](int mt, int b, int a) -> unsigned {
        const int m = mt * 128 + b * 16 + lr;
        unsigned rrow = (unsigned)m;
        if (p.res_mode == RES_UP2) {                   // (neck.py:36-38) pixel (img, oh, ow) <- (img, oh / 2, ow / 2) of the half-resolution map
            const int mm = m < p.M ? m : 0;
            const int t = fastdiv(mm, p.mg_ow), ow = mm - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
            rrow = (unsigned)((img * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1));
        }
        const int n = nb + 16 * a + 4 * g;
        return (m < p.M && n < p.N_g) ? (rrow * (unsigned)p.ldr + (unsigned)n) * 4u : 0u;
    };
    auto epi_frag = [&](int mt, auto FF, auto JJ) {
        constexpr int f = decltype(FF)::value, b = f / NI, a = f % NI, j = decltype(JJ)::value;
        const f32x4 c = acc[a][b];
        float v[4] = {unscale<T>(c[0], p.wscale), unscale<T>(c[1], p.wscale), unscale<T>(c[2], p.wscale), unscale<T>(c[3], p.wscale)};
        if (RESK != 0 && p.res_post) activate(v);
        if constexpr (RESK == 2 && IsSplit<T>::value) {
            const u32x4 u = res_r[j];
            float rv[4];
            Chunk<T>::unpack(make_uint4(u.x, u.y, u.z, u.w), rv);
            v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
        } else if constexpr (RESK != 0) {
            const u32x4 u = res_r[j];
            v[0] += __uint_as_float(u.x); v[1] += __uint_as_float(u.y);
            v[2] += __uint_as_float(u.z); v[3] += __uint_as_float(u.w);
        }
        if (!(RESK != 0 && p.res_post)) activate(v);
        const int m = mt * 128 + b * 16 + lr, cn = 16 * a + 4 * g;
        const unsigned off = (m < p.M && nb + cn < p.N_g) ? ((unsigned)m * (unsigned)p.ldo + (unsigned)(p.out_coff + nb + cn)) * 4u : OOB;
        u32x4 pk;
        if constexpr (IsSplit<T>::value && !OUTF32) {
            range_mask |= f16x2_out_of_range(v);
            const uint4 e = Chunk<T>::pack(v);
            pk = (u32x4){e.x, e.y, e.z, e.w};
        } else {
            pk = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        }
        __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // epilogue step E0 of the group's finished tile: residual loads, this step's DMA pieces, then F fragments.  Returns with exactly
    // F stores issued after the DMA pieces (a step of a 48-column build that has run out of fragments pads with out-of-range stores).
    auto epi_step = [&](auto EE, int mt, int na) {
        constexpr int e = decltype(EE)::value;
        auto each = [&](auto fn) {
            if constexpr (F >= 1) fn(IC<0>{});
            if constexpr (F >= 2) fn(IC<1>{});
            if constexpr (F >= 4) { fn(IC<2>{}); fn(IC<3>{}); }
            if constexpr (F >= 8) { fn(IC<4>{}); fn(IC<5>{}); fn(IC<6>{}); fn(IC<7>{}); }
            if constexpr (F >= 16) { fn(IC<8>{}); fn(IC<9>{}); fn(IC<10>{}); fn(IC<11>{}); fn(IC<12>{}); fn(IC<13>{}); fn(IC<14>{}); fn(IC<15>{}); }
            static_assert(F <= 16, "unroll more");
        };
        if constexpr (RESK != 0) {
            each([&](auto J) {
                constexpr int j = decltype(J)::value, f = e * F + j;
                if constexpr (f < FR) gload16s(res_r[j], res_base, res_off(mt, f / NI, f % NI));
                else gload16s(res_r[j], res_base, 0u);
            });
        }
        issue_all();
        if constexpr (RESK != 0) {   // the residual loads are older than this step's NI + na DMA pieces only
            if (na == 0) duo_wait_vm<NI>(); else if (na == 2) duo_wait_vm<NI + 2>(); else duo_wait_vm<NI + 4>();
            each([&](auto J) { bind16(res_r[decltype(J)::value]); });
        }
        each([&](auto J) {
            constexpr int j = decltype(J)::value, f = e * F + j;
            if constexpr (f < FR) {
                epi_frag(mt, IC<f>{}, J);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, orsrc, OOB, 0, 0);
            }
        });
    };

    // ---- prologue: weight step 0, activation steps 0 and 1 of whichever group starts there
    if (n_my > 0) {
        setup_a(0, 0);
        if (T1 > 0) setup_a(1, 0);
    }
    int last_na = 0;
    bool last_stored = false;
    if (S > 0) {
        // (the generic issue path is driven with fake step numbers: B(0) goes to slot 0 as "step -1", A for steps 0 and 1 as "steps -2, -1")
        i_pos[0] = 0; i_pos[1] = -off1;
        {   // A stages of step 0
            st_bdst = lds0; st_bk = b_tile;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                st_a[gg] = i_pos[gg] >= 0 && i_pos[gg] < nk && i_til[gg] < Tg[gg];
                st_ak[gg] = a_tile[gg];
                st_adst[gg] = lds0 + A_BASE + (gg * 3 + i_slot[gg]) * ASTAGE;
            }
            issue_piece(IC<NI>{}); issue_piece(IC<NI + 1>{}); issue_piece(IC<NI + 2>{}); issue_piece(IC<NI + 3>{});
            // B of step 0 (its K-step counter advances here; the A cursor's in end_issue below)
            issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{});
            if constexpr (NI == 4) issue_piece(IC<3>{});
            const int kb_keep = i_kb;
            end_issue();               // A cursor -> step 1; (i_kb advanced to 1: B(1) is what step 0 issues)
            (void)kb_keep;
        }
        {   // A stages of step 1
            const int kb_keep = i_kb;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                st_a[gg] = i_pos[gg] >= 0 && i_pos[gg] < nk && i_til[gg] < Tg[gg];
                if (st_a[gg]) {
                    st_ak[gg] = a_tile[gg] + (size_t)i_ka * 128;
                    st_adst[gg] = lds0 + A_BASE + (gg * 3 + i_slot[gg]) * ASTAGE;
                }
            }
            last_na = (st_a[0] ? 2 : 0) + (st_a[1] ? 2 : 0);
            issue_piece(IC<NI>{}); issue_piece(IC<NI + 1>{}); issue_piece(IC<NI + 2>{}); issue_piece(IC<NI + 3>{});
            end_issue();
            i_kb = kb_keep;            // the prologue issued B(0) only
        }
    }

    for (int s = 0; s < S; ++s) {
        // stage s has landed: everything this wave issued up to B(s)'s pieces (what may still fly: the A pieces and the stores behind them)
        duo_wait_dyn<F>(last_na, last_stored);
        asm volatile("s_barrier" ::: "memory");
        const int na = begin_issue(s);
        const bool mf = e_pos >= 0 && e_pos < nk && e_til < myT;
        const bool ep = e_pos >= nk && e_til < myT;
        const int mt = mt0 + (2 * e_til + grp) * Gm;
        if (mf) {
            if (e_pos == 0) {
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b) acc[a][b] = bias_r[a];
            }
            mfma_step(smem + A_BASE + (grp * 3 + e_slot) * ASTAGE, smem + (s & 1) * BSTAGE);
            if (++e_slot == 3) e_slot = 0;
            last_stored = false;
        } else if (ep) {
            const int e = e_pos - nk;
            bool done = false;
            auto pick = [&](auto EE) {
                if (!done && e == decltype(EE)::value) {
                    epi_step(EE, mt, na);
                    done = true;
                }
            };
            if constexpr (EPI >= 1) pick(IC<0>{});
            if constexpr (EPI >= 2) pick(IC<1>{});
            if constexpr (EPI >= 4) { pick(IC<2>{}); pick(IC<3>{}); }
            if constexpr (EPI >= 8) { pick(IC<4>{}); pick(IC<5>{}); pick(IC<6>{}); pick(IC<7>{}); }
            if constexpr (EPI >= 16) { pick(IC<8>{}); pick(IC<9>{}); pick(IC<10>{}); pick(IC<11>{}); pick(IC<12>{}); pick(IC<13>{}); pick(IC<14>{}); pick(IC<15>{}); }
            if constexpr (EPI >= 32) {
                pick(IC<16>{}); pick(IC<17>{}); pick(IC<18>{}); pick(IC<19>{}); pick(IC<20>{}); pick(IC<21>{}); pick(IC<22>{}); pick(IC<23>{});
                pick(IC<24>{}); pick(IC<25>{}); pick(IC<26>{}); pick(IC<27>{}); pick(IC<28>{}); pick(IC<29>{}); pick(IC<30>{}); pick(IC<31>{});
            }
            last_stored = true;
        } else {
            issue_all();       // idle (group 1 before its first tile, a group after its last): the DMA duty stays
            last_stored = false;
        }
        end_issue();
        last_na = na;
        if (++e_pos == P) {
            e_pos = 0;
            ++e_til;
        }
    }
    duo_wait_vm<0>();          // no DMA may land in LDS after the workgroup has gone
    if constexpr (IsSplit<T>::value && !OUTF32) f16x2_raise(range_mask);
}

template <typename T> int launch_gemm_duo(const ConvParams& p, hipStream_t stream);
// true when p is a GEMM the duo kernel is built for (and faster at than gemm_ring)
bool gemm_duo_eligible(const ConvParams& p, int amode, int dtype);

}  // namespace ocrvi
