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
    int nt = 0;        // non-temporal output stores
    int dbg = 0;       // development (OCRVI_DUO_DBG, timing experiments with wrong results): 1 = every store out of range, 2 = no fragment
                       // reads / MFMAs (the DMA stream alone), 4 = no DMA (the arithmetic alone), 8 = no barrier, 16 = no fragment reads
};

template <int N> __device__ __forceinline__ void duo_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// RESK: 0 no residual, 1 raw fp32 residual, 2 residual in T's own format (f16x2 chunks; fp32: the same as 1).  OUTF32: raw fp32 output
// (always for T = float).  F: fragments (= stores, = residual loads) per wave and epilogue step; a tile's epilogue takes 8 NI / F steps.
// PROF (development, -DOCRVI_RING_PROF_BUILD + OCRVI_RING_PROF=1): shader-clock cycles per wave in the own-DMA wait / at the barrier / in MFMA
// steps / in epilogue steps / in idle steps / in the cursor bookkeeping, added into p.out2 (uint64[8]) at exit.
template <typename T, int NI, int ACT, int RESK, bool OUTF32, int F, bool PROF = false>
__global__ __launch_bounds__(512, 2) void gemm_duo_kernel(const ConvParams p, const DuoPlan plan) {
    static_assert(sizeof(T) == 4, "4-byte operand types");
    static_assert(NI == 4 || NI == 3, "64 or 48 columns per wave");
    constexpr int MI = 8, BN = 16 * NI * 4, BSTAGE = BN * 128, ASTAGE = 128 * 128;
    constexpr int FR = MI * NI;                       // fragments per wave
    static_assert(FR % F == 0 && F <= 16, "fragments per epilogue step");
    constexpr int EPI = FR / F;                       // epilogue steps per tile
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
    // group 1 runs this many steps behind group 0: enough that the two groups' epilogues never coincide (each overlaps MFMA steps of the
    // other group); no more, because group 1 idles that long at the start and group 0 at the end.  (>= 2: host contract nk >= 2, EPI >= 2)
    const int off1 = min(EPI, P >> 1);
    const int S = max(T0 * P, T1 > 0 ? off1 + T1 * P : 0);   // steps of this workgroup
    const int myT = grp == 0 ? T0 : T1;

    // ---- DMA duty (gemm_ring.h: piece = 8 rows x 128 B, XOR swizzle on the source chunk).  Every wave issues, per step, NI weight pieces
    // of step s + 1 (rows pi * 64 + wave * 8 + prow of the column tile) and -- when its OWN group runs an MFMA step at s + 2 -- four
    // activation pieces of that step (rows (i * 4 + wn) * 8 + prow of the group's tile): a wave's issue state is its own group's
    // execution cursor, nothing else.
    const int prow = lane >> 3;
    const int chunk = (lane & 7) ^ swz128(wave * 8 + prow);      // (rows' bits 1 and 3: prow bit 1 and the wave's bit 0, for both streams)
    unsigned b_off[NI];
#pragma unroll
    for (int pi = 0; pi < NI; ++pi) b_off[pi] = (unsigned)((pi * 64 + wave * 8 + prow) * ldw_b + chunk * 16);
    const char* const b_tile = uniform_ptr((const char*)p.w + (size_t)(nt * BN) * ldw_b);
    unsigned a_off[4];
    const char* a_tile = nullptr;
    auto setup_a = [&](int til) {                      // per-lane source offsets of the group's tile number til
        const int mt = mt0 + (2 * til + grp) * Gm;
        if (p.SH != 1 || p.SW != 1) {                  // strided 1x1 (ResNet downsample): input pixel (img, oh * SH, ow * SW)
            a_tile = uniform_ptr(A);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = min(mt * 128 + (i * 4 + wn) * 8 + prow, p.M - 1);
                const int t = fastdiv(m, p.mg_ow), ow = m - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
                a_off[i] = (unsigned)((img * p.H + oh * p.SH) * p.W + ow * p.SW) * (unsigned)lda_b + chunk * 16;
            }
        } else {
            a_tile = uniform_ptr(A + (size_t)mt * 128 * lda_b);
            const int last = p.M - 1 - mt * 128;       // rows past M read the last valid row (what they produce is never stored)
#pragma unroll
            for (int i = 0; i < 4; ++i) a_off[i] = (unsigned)(min((i * 4 + wn) * 8 + prow, last) * lda_b + chunk * 16);
        }
    };
    // this step's DMA constants: weight stage of step s + 1 (K-step kb1) and, if st_a, the group's activation stage of step s + 2 (ka2)
    unsigned st_bdst = 0, st_adst = 0;
    const char *st_bk = nullptr, *st_ak = nullptr;
    bool st_a = false;
    // the compiler makes no use of M0 in this kernel (checked by tools/check_ring_isa.py), so a piece sets it and leaves it
    auto dma16 = [&](const char* sbase, unsigned voff, unsigned lds_dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    auto issue_piece = [&](auto PI) {
        constexpr int pi = decltype(PI)::value;
        if (plan.dbg & 4) return;
        if constexpr (pi < NI) {
            dma16(st_bk, b_off[pi], __builtin_amdgcn_readfirstlane(st_bdst + (pi * 8 + wave) * 1024));
        } else if constexpr (pi < NI + 4) {
            constexpr int i = pi - NI;
            if (st_a) dma16(st_ak, a_off[i], __builtin_amdgcn_readfirstlane(st_adst + (i * 4 + wn) * 1024));
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
    int e_pos = grp == 0 ? 0 : -off1;                  // position in the period of step s (negative: group 1's initial delay)
    int e_til = 0;                                     // tiles the group has finished
    int e_slot = 0;                                    // A slot of the group's next MFMA step

    // one MFMA step: all NI weight fragments up front, the eight row blocks streamed one ahead; DMA piece b rides behind row block b
    auto mfma_step = [&](const char* As, const char* Bs) {
        const char* Br = Bs + (wn * (16 * NI) + lr) * 128;
        const char* Ar = As + lr * 128;
        if constexpr (IsSplit<T>::value) {
            // Fixed schedule (pinned with sched_barrier between every pair of MFMAs): a row block is 3 NI MFMAs = three products x NI column
            // blocks; the other work of the block -- the two fragment reads of row block b + 2, the regrouping of row block b + 1 into its
            // (hi, lo) quartets (a second register set) and one DMA piece -- sits BETWEEN MFMA pairs, so that a wave that has the SIMD's
            // matrix pipe to itself (its partner is in an epilogue step) still feeds it back to back instead of pausing at block borders.
            typedef typename Mma<T>::u4v U;
            uint4 wc[NI][2], xc[2][2];
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                wc[a][0] = lds16(Br + a * 2048 + foa0);
                wc[a][1] = lds16(Br + a * 2048 + foa1);
            }
            xc[0][0] = lds16(Ar + foa0);
            xc[0][1] = lds16(Ar + foa1);
            xc[1][0] = lds16(Ar + 2048 + foa0);
            xc[1][1] = lds16(Ar + 2048 + foa1);
            __builtin_amdgcn_sched_barrier(0);
            U wH[NI], wL[NI], xH[2], xL[2];
#pragma unroll
            for (int a = 0; a < NI; ++a) Mma<T>::regroup(wc[a][0], wc[a][1], wH[a], wL[a]);
            Mma<T>::regroup(xc[0][0], xc[0][1], xH[0], xL[0]);
            __builtin_amdgcn_sched_barrier(0);
            auto mm = [&](const U& w, const U& x, f32x4& c) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
            };
            auto blk = [&](auto BB) {
                constexpr int b = decltype(BB)::value, cur = b & 1, nxt = cur ^ 1;
                if (plan.dbg & 256) {   // (experiment) the SIMD partners alternate issue priority block by block
                    if ((b & 1) == grp) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
                }
                // products: wL.xH, wH.xL, wH.xH (Mma<f16x2_t>::three), column blocks in pairs
                mm(wL[0], xH[cur], acc[0][b]); mm(wL[1], xH[cur], acc[1][b]);
                if constexpr (b + 2 < MI) xc[cur][0] = lds16(Ar + (b + 2) * 2048 + foa0);     // (xc[cur] was regrouped during block b - 1)
                __builtin_amdgcn_sched_barrier(0);
                mm(wL[2], xH[cur], acc[2][b]);
                if constexpr (NI == 4) mm(wL[3], xH[cur], acc[3][b]);
                if constexpr (b + 2 < MI) xc[cur][1] = lds16(Ar + (b + 2) * 2048 + foa1);
                __builtin_amdgcn_sched_barrier(0);
                mm(wH[0], xL[cur], acc[0][b]); mm(wH[1], xL[cur], acc[1][b]);
                if constexpr (b + 1 < MI) xH[nxt] = (U){xc[nxt][0].x, xc[nxt][0].y, xc[nxt][1].x, xc[nxt][1].y};
                __builtin_amdgcn_sched_barrier(0);
                mm(wH[2], xL[cur], acc[2][b]);
                if constexpr (NI == 4) mm(wH[3], xL[cur], acc[3][b]);
                if constexpr (b + 1 < MI) xL[nxt] = (U){xc[nxt][0].z, xc[nxt][0].w, xc[nxt][1].z, xc[nxt][1].w};
                __builtin_amdgcn_sched_barrier(0);
                mm(wH[0], xH[cur], acc[0][b]); mm(wH[1], xH[cur], acc[1][b]);
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(IC<b>{});
                __builtin_amdgcn_sched_barrier(0);
                mm(wH[2], xH[cur], acc[2][b]);
                if constexpr (NI == 4) mm(wH[3], xH[cur], acc[3][b]);
                __builtin_amdgcn_sched_barrier(0);
            };
            // Issue priority: the two waves of a SIMD are in different groups.  With equal priority the older wave wins every arbitration, runs
            // its step at full speed and then waits at the barrier while the younger one works through its step alone, gaps unfilled.  Group 0
            // takes priority in the first half of a step and group 1 in the second, so both progress side by side and fill each other's gaps.
            const bool flip = !(plan.dbg & 64) && !(plan.dbg & 256);
            if (flip) { if (grp == 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
            blk(IC<0>{}); blk(IC<1>{}); blk(IC<2>{}); blk(IC<3>{});
            if (flip) { if (grp == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
            blk(IC<4>{}); blk(IC<5>{}); blk(IC<6>{}); blk(IC<7>{});
            if (flip) __builtin_amdgcn_s_setprio(0);
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
        unsigned off = (m < p.M && nb + cn < p.N_g) ? ((unsigned)m * (unsigned)p.ldo + (unsigned)(p.out_coff + nb + cn)) * 4u : OOB;
        if (plan.dbg & 1) off = OOB;
        u32x4 pk;
        if constexpr (IsSplit<T>::value && !OUTF32) {
            range_mask |= f16x2_out_of_range(v);
            const uint4 e = Chunk<T>::pack(v);
            pk = (u32x4){e.x, e.y, e.z, e.w};
        } else {
            pk = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        }
        // nt: the output of a GEMM with N >= 2 K outweighs what it reads; written with the default policy its lines push the activation
        // rows the other column tiles still want out of the XCD's L2 (measured: 292 -> 256 us at K = 256, N = 1024; 254 -> 279 us at K = 1536)
        if (plan.nt) __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // epilogue step e of the group's finished tile: residual loads, this step's DMA pieces, then F fragments.  Exactly F stores follow the
    // DMA pieces (out-of-range lanes are dropped by the buffer descriptor's range check, the instruction still issues).
    long long* tkp = nullptr;   // (PROF: the loop's stamp state, so that the epilogue can split its own time)
    long long* t0p = nullptr;
    auto etick = [&](int k) {
        if constexpr (PROF) {
            const long long t = clock64();
            tkp[k] += t - *t0p;
            *t0p = t;
        }
    };
    auto epi_step = [&](auto EE, int mt) {
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
                gload16s(res_r[j], res_base, res_off(mt, f / NI, f % NI));
            });
        }
        etick(6);
        // Issue priority: the epilogue's ~350 vector / memory instructions compete with the partner wave's MFMA stream for the SIMD's issue
        // slots, and arbitration is by priority, then AGE -- at equal priority waves 4-7 (the younger half) got the leftovers: their epilogue
        // steps took 19 k cycles against 3.3 k for waves 0-3 (phase stamps, K 1536), with everybody else waiting at the barrier.
        if (plan.dbg & 128) __builtin_amdgcn_s_setprio(3);
        issue_all();
        etick(7);
        if constexpr (RESK != 0) {   // the residual loads are older than this step's NI (+ 4) DMA pieces only
            if (st_a) duo_wait_vm<NI + 4>(); else duo_wait_vm<NI>();
            each([&](auto J) { bind16(res_r[decltype(J)::value]); });
        }
        each([&](auto J) {
            constexpr int j = decltype(J)::value, f = e * F + j;
            epi_frag(mt, IC<f>{}, J);
        });
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: the weight stage of step 0 (all waves) and the activation stages of steps 0 and 1 of group 0 (its waves; group 1's first
    // tile starts off1 >= 2 steps later and is issued from the loop)
    int kb = 0, kb1 = nk > 1 ? 1 : 0, ka2 = nk > 2 ? 2 : (2 % nk);   // K-step of step s / s + 1 / s + 2
    bool last_a = false, last_stored = false;
    if (S > 0) {
        if (myT > 0) setup_a(0);
        st_bdst = lds0; st_bk = b_tile;
        issue_piece(IC<0>{}); issue_piece(IC<1>{}); issue_piece(IC<2>{});
        if constexpr (NI == 4) issue_piece(IC<3>{});
        if (grp == 0) {
            st_a = true;
            st_ak = a_tile; st_adst = lds0 + A_BASE;
            issue_piece(IC<NI>{}); issue_piece(IC<NI + 1>{}); issue_piece(IC<NI + 2>{}); issue_piece(IC<NI + 3>{});
            st_ak = uniform_ptr(a_tile + (size_t)kb1 * 128); st_adst = lds0 + A_BASE + ASTAGE;
            issue_piece(IC<NI>{}); issue_piece(IC<NI + 1>{}); issue_piece(IC<NI + 2>{}); issue_piece(IC<NI + 3>{});
            last_a = true;   // (the wait of step 0 may leave the four pieces of step 1 in flight)
        }
    }

    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    auto tick = [&](int k) {
        if constexpr (PROF) {
            const long long t = clock64();
            tk[k] += t - t0;
            t0 = t;
        }
    };
    if constexpr (PROF) t0 = clock64();
    tkp = tk; t0p = &t0;
    for (int s = 0; s < S; ++s) {
        // stage s has landed: everything this wave issued up to the weight pieces of step s (what may still fly: its four activation
        // pieces of step s + 1 and the stores behind them)
        if (!last_stored) {
            if (last_a) duo_wait_vm<4>(); else duo_wait_vm<0>();
        } else {
            if (last_a) duo_wait_vm<F + 4>(); else duo_wait_vm<F>();
        }
        tick(0);
        if (!(plan.dbg & 8)) asm volatile("s_barrier" ::: "memory");
        tick(1);
        const bool live = e_til < myT;
        const bool mf = live && e_pos >= 0 && e_pos < nk;
        const bool ep = live && e_pos >= nk;
        const int e = e_pos - nk;
        // DMA plan of the step
        st_bdst = lds0 + ((s + 1) & 1) * BSTAGE;
        st_bk = uniform_ptr(b_tile + (size_t)kb1 * 128);
        int aslot = -1;                                 // A slot the group's stage of step s + 2 goes to (-1: none)
        if (mf) {
            if (e_pos + 2 < nk) aslot = e_slot + 2;
        } else if (ep) {
            if (e == 0 && e_til + 1 < myT) setup_a(e_til + 1);           // (the finished tile's last activation stage was issued two steps ago)
            if (e >= EPI - 2 && e_til + 1 < myT) aslot = e_slot + (e - (EPI - 2));
        } else if (live && e_pos >= -2) {
            aslot = e_pos + 2;                          // group 1's first tile: its steps 0 and 1
        }
        st_a = aslot >= 0;
        if (st_a) {
            if (aslot >= 3) aslot -= 3;
            st_ak = uniform_ptr(a_tile + (size_t)ka2 * 128);
            st_adst = lds0 + A_BASE + (grp * 3 + aslot) * ASTAGE;
        }
        const int mt = mt0 + (2 * e_til + grp) * Gm;
        if (mf) {
            if (e_pos == 0) {
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b) acc[a][b] = bias_r[a];
            }
            if (plan.dbg & 2) issue_all();
            else mfma_step(smem + A_BASE + (grp * 3 + e_slot) * ASTAGE, smem + (s & 1) * BSTAGE);
            if (++e_slot == 3) e_slot = 0;
            last_stored = false;
            tick(2);
        } else if (ep) {
            bool done = false;
            auto pick = [&](auto EE) {
                if (!done && e == decltype(EE)::value) {
                    epi_step(EE, mt);
                    done = true;
                }
            };
            if constexpr (EPI > 0) pick(IC<0>{});
            if constexpr (EPI > 1) pick(IC<1>{});
            if constexpr (EPI > 2) pick(IC<2>{});
            if constexpr (EPI > 3) pick(IC<3>{});
            if constexpr (EPI > 4) pick(IC<4>{});
            if constexpr (EPI > 5) pick(IC<5>{});
            if constexpr (EPI > 6) pick(IC<6>{});
            if constexpr (EPI > 7) pick(IC<7>{});
            if constexpr (EPI > 8) pick(IC<8>{});
            if constexpr (EPI > 9) pick(IC<9>{});
            if constexpr (EPI > 10) pick(IC<10>{});
            if constexpr (EPI > 11) pick(IC<11>{});
            if constexpr (EPI > 12) pick(IC<12>{});
            if constexpr (EPI > 13) pick(IC<13>{});
            if constexpr (EPI > 14) pick(IC<14>{});
            if constexpr (EPI > 15) pick(IC<15>{});
            static_assert(EPI <= 16, "more epilogue steps than the dispatch chain covers");
            last_stored = true;
            tick(3);
        } else {
            issue_all();       // idle (group 1 before its first tile, a group after its last): the weight DMA duty stays
            last_stored = false;
            tick(4);
        }
        last_a = st_a;
        kb = kb1; kb1 = ka2;
        if (++ka2 == nk) ka2 = 0;
        if (++e_pos == P) {
            e_pos = 0;
            ++e_til;
        }
        tick(5);
    }
    if constexpr (PROF) {
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long*)p.out2 + grp * 8 + k, (unsigned long long)tk[k]);
    }
    duo_wait_vm<0>();          // no DMA may land in LDS after the workgroup has gone
    if constexpr (IsSplit<T>::value && !OUTF32) f16x2_raise(range_mask);
}

template <typename T> int launch_gemm_duo(const ConvParams& p, hipStream_t stream);
template <> int launch_gemm_duo<f16x2_t>(const ConvParams& p, hipStream_t stream);   // duo_f16x2.hip
// column tile for Np output columns: 256 (64 per wave) when it divides, else 192 (48 per wave), else 0 = not a duo shape
static inline int duo_bn_for(int Np) { return Np % 256 == 0 ? 256 : (Np % 192 == 0 ? 192 : 0); }
// true when p is a GEMM the duo kernel is built for (and faster at than gemm_ring)
bool gemm_duo_eligible(const ConvParams& p, int amode, int dtype);

}  // namespace ocrvi
