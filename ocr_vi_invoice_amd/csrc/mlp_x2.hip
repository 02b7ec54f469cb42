// Fused MixingBlock MLP for the f16x2 mode, D = 128 / 256 (svtrv2.py:28-39,100):   x <- x + fc2(gelu(fc1(LayerNorm(x))))   [+ the NEXT LayerNorm]
//
// The f16x2 counterpart of mlp_fused.hip, re-cut for operands that are twice as wide: unfused, the block is a LayerNorm launch and two
// ring GEMMs whose 4 D hidden tensor makes a round trip through HBM (M x 4 D x 4 B written and read: 503 MB per 256 crops at D = 256).
// One persistent workgroup per CU, 8 waves; a workgroup owns 128 tokens, a wave 16 of them for the whole block:
//   * prologue: the wave's 16 token rows (fp32 residual stream) -> LayerNorm statistics over the 4 lanes that share a token -> the
//     normalised rows as MFMA B-operand fragments in REGISTERS, (hi, lo) quartets: D / 4 VGPRs per lane;
//   * the hidden dimension in chunks of 32: GEMM1 (32 x D slice of fc1, three MFMAs per fragment pair) -> bias + exact GELU in registers
//     -> hi / lo split -> the lane's 8 hidden values ARE its 8 k-slots of GEMM2's B operand (fc2's K order is permuted at pack time:
//     slot (g, j) <-> hidden 16 (j >> 2) + 4 g + (j & 3)), so h never touches LDS; GEMM2 (D x 32 slice of fc2) accumulates all D outputs
//     of the wave's tokens in registers (D / 4 VGPRs);
//   * the only streamed operand is the weights, identical for every tile: one 256 D-byte unit per chunk (fc1 slice, then fc2 slice), packed
//     on the host in LDS image order -- quartet form, XOR-swizzled -- so the LDS-DMA copies it linearly; a ring of 2 (D = 256) / 4 (D = 128)
//     units runs ahead across chunk and tile boundaries, counted vmcnt, one barrier per unit;
//   * epilogue: + bias + fp32 residual -> x; optionally LayerNorm of the result with the next block's norm1 (or a plain cast) -> xn (f16x2,
//     range-checked), which removes that LayerNorm / cast kernel too.
// D = 384 does not fit: the normalised rows and the output accumulators alone are 192 of a wave's 256 registers there.
#include <stdlib.h>
#include <string.h>

#include "gemm_ring.h"
#include "kernels.h"

namespace ocrvi {
OCRVI_RANGE_FLAG_TU()

struct MlpX2Params {
    float* x = nullptr;            // [M][D] fp32 residual stream, updated in place
    void* xn = nullptr;            // optional [M][D] f16x2: LayerNorm_next(x_new) (next_g != null) or f16x2(x_new) (next_g == null)
    const float* ln_g = nullptr;   // norm2 of this block
    const float* ln_b = nullptr;
    const float* next_g = nullptr;
    const float* next_b = nullptr;
    const char* wstream = nullptr; // pack_mlp_x2_stream: 4 D / 32 + 1 units of 256 D bytes, then {wscale1, wscale2}
    const float* b1 = nullptr;     // [4D]
    const float* b2 = nullptr;     // [D]
    int M = 0;
    int stagger = 0;               // shader cycles between the start phases of the workgroups (see the kernel)
    int dbg = 0;                   // OCRVI_MLPX2_DBG (development, wrong results): 1 no x stores, 2 no xn stores, 4 no residual loads
    unsigned long long* prof = nullptr;   // development (OCRVI_MLPX2_PROF=1): cycles per wave in unit wait+barrier / GEMM1 (+ GELU) / GEMM2 / tile epilogue / prologue / drain
};

// TB = 16-token blocks per wave: 1 -> 8 waves of 16 tokens (two per SIMD, 256 registers each); 2 -> 4 waves of 32 tokens (one per SIMD with the
// whole 512-entry file: the only way the normalised rows and the output accumulators of D = 384 fit -- 2 x 96 + 2 x 96 registers -- and every
// weight fragment read from LDS feeds two token blocks).  SPLIT: the ring's unit is one HALF of a stream unit (the fc1 slice or the fc2 slice,
// 128 D bytes; a wait + barrier before each GEMM) instead of the pair: a pair is 96 KB at D = 384, two of them do not fit the LDS.
template <int D, int R, int TB, bool SPLIT>
__global__ __launch_bounds__(512 / TB, TB == 2 ? 1 : 2) void mlp_x2_kernel(const MlpX2Params p) {
    typedef f16x2_t T;
    typedef Mma<T>::u4v U;
    constexpr int NWV = 8 / TB, NT = 64 * NWV;   // waves, threads
    constexpr int KS = D / 32;         // 128-byte K-steps of GEMM1
    constexpr int NCH = 4 * D / 32;    // hidden chunks of 32
    constexpr int NB2 = D / 16;        // 16-channel output blocks of GEMM2
    constexpr int W1B = 32 * D * 4;    // bytes of a stream unit's fc1 slice: [KS][32 hidden rows][128 B]; the fc2 slice [D output rows][128 B] is as long
    constexpr int UNIT = SPLIT ? W1B : 2 * W1B;   // the ring's unit
    constexpr int IPW = UNIT / 1024 / NWV;        // DMA instructions per wave per ring unit
    constexpr int PF = R - 1;          // ring units in flight
    constexpr int NU = (NCH + 1) * (SPLIT ? 2 : 1);   // ring units per token tile
    static_assert(UNIT % (1024 * NWV) == 0 && R >= 2, "ring geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    float* const c_b1 = (float*)(smem + R * UNIT);     // [4D]
    float* const c_b2 = c_b1 + 4 * D;                  // [D]
    float* const c_g = c_b2 + D;                       // norm2 gamma, beta, next gamma, beta: [D] each
    float* const c_be = c_g + D;
    float* const c_ng = c_be + D;
    float* const c_nb = c_ng + D;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const bool next_ln = p.next_g != nullptr;
    for (int i = tid; i < 4 * D; i += NT) c_b1[i] = p.b1[i];
    for (int i = tid; i < D; i += NT) {
        c_b2[i] = p.b2[i];
        c_g[i] = p.ln_g[i];
        c_be[i] = p.ln_b[i];
        c_ng[i] = next_ln ? p.next_g[i] : 1.f;
        c_nb[i] = next_ln ? p.next_b[i] : 0.f;
    }
    const float* const tail = (const float*)(p.wstream + (size_t)(NCH + 1) * 2 * W1B);
    const float ws1 = tail[0], ws2 = tail[1];
    __syncthreads();  // (also drains those loads: no VMEM op is in flight when the ring starts)

    const int ntiles = (p.M + 127) >> 7;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;

    // ---- weight ring: ring unit q of the stream is unit q % NU of the packed buffer, already in LDS image order; this wave copies 1-KiB pieces
    // wave, wave + NWV, ... of it.  Branch-free: past the end of the workgroup's stream it keeps fetching units nobody reads (their slots are
    // free), drained before the kernel ends.
    const char* const wbase = uniform_ptr(p.wstream);
    int prod_slot = 0, prod_mod = 0, cons_slot = 0;
    auto issue_unit = [&]() {
        const char* src = wbase + (size_t)prod_mod * UNIT + wave * 1024;
        const unsigned dst = lds0 + prod_slot * UNIT + wave * 1024;
#pragma unroll
        for (int j = 0; j < IPW; ++j) glds16(src + j * (NWV * 1024), (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(dst + j * (NWV * 1024)));
        prod_slot = prod_slot + 1 == R ? 0 : prod_slot + 1;
        prod_mod = prod_mod + 1 == NU ? 0 : prod_mod + 1;
    };
    auto next_unit = [&]() -> const char* {  // wait for the oldest unit in flight, free the slot before it, keep the ring full
        wait_vm_barrier<(PF - 1) * IPW>();
        issue_unit();
        const char* s = smem + cons_slot * UNIT;
        cons_slot = cons_slot + 1 == R ? 0 : cons_slot + 1;
        return s;
    };
    for (int i = 0; i < PF; ++i) issue_unit();
    // (Experiment, OCRVI_MLPX2_STAGGER=cycles, off by default: four start phases `stagger` cycles apart, to test whether the tile epilogues of all
    // CUs -- which fall into the same few microseconds -- starve each other of HBM bandwidth.  They do not: 12 k ... 60 k cycles of stagger only add
    // their own delay; removing every load and store of the epilogue saves 7 %.  The epilogue's cost is its ~1 300 vector instructions.)
    if (p.stagger > 0) {
        const long long until = clock64() + (long long)(blockIdx.x & 3) * p.stagger;
        while (clock64() < until) __builtin_amdgcn_s_sleep(64);
    }

    long long tk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tk0 = p.prof ? clock64() : 0;
    auto tick = [&](int k) {
        if (p.prof) {
            const long long c = clock64();
            tk[k] += c - tk0;
            tk0 = c;
        }
    };
    const int sw = swz128(lr);
    const int fo0 = ((2 * g) ^ sw) << 4, fo1 = ((2 * g + 1) ^ sw) << 4;   // hi / lo quartet of this lane's 8 k-slots in a [row][128 B] image
    unsigned long long range_mask = 0;

    for (int t = 0; t < my_tiles; ++t) {
        const int tile = (int)blockIdx.x + t * G;
        const int tok0 = tile * 128 + wave * (16 * TB);
        // ---- prologue: LayerNorm(x) of this wave's tokens -> B-operand fragments.  Lane (lr, g) of token block b owns channels 32 ks + 8 g .. + 8
        // of token tok0 + 16 b + lr for every K-step ks: the lane's 8 k-slots.
        U xH[KS][TB], xL[KS][TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            float xv[KS][8];
            const int tok = min(tok0 + 16 * b + lr, p.M - 1);
            const float* xr = p.x + (size_t)tok * D + 8 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float4 v0 = *(const float4*)(xr + 32 * ks), v1 = *(const float4*)(xr + 32 * ks + 4);
                xv[ks][0] = v0.x; xv[ks][1] = v0.y; xv[ks][2] = v0.z; xv[ks][3] = v0.w;
                xv[ks][4] = v1.x; xv[ks][5] = v1.y; xv[ks][6] = v1.z; xv[ks][7] = v1.w;
            }
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += xv[ks][e];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum / (float)D;
            float sq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xv[ks][e] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            const float rstd = rsqrtf(sq / (float)D + 1e-5f);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float o[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 gv = *(const float4*)(c_g + 32 * ks + 8 * g + 4 * q), bv = *(const float4*)(c_be + 32 * ks + 8 * g + 4 * q);
                    o[4 * q] = (xv[ks][4 * q] - mean) * rstd * gv.x + bv.x;
                    o[4 * q + 1] = (xv[ks][4 * q + 1] - mean) * rstd * gv.y + bv.y;
                    o[4 * q + 2] = (xv[ks][4 * q + 2] - mean) * rstd * gv.z + bv.z;
                    o[4 * q + 3] = (xv[ks][4 * q + 3] - mean) * rstd * gv.w + bv.w;
                }
                const uint4 c0 = Chunk<T>::pack(o), c1 = Chunk<T>::pack(o + 4);
                Mma<T>::regroup(c0, c1, xH[ks][b], xL[ks][b]);
            }
        }
        tick(4);
        // every VMEM op issued so far by this wave (ring DMAs, the previous tile's stores, the loads above) has completed: the counted
        // waits of the main loop start from the DMAs issued from here on (any older unit has landed)
        wait_vm_only<0>();
        tick(5);

        f32x4 acc2[NB2][TB];
#pragma unroll
        for (int a = 0; a < NB2; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // Software pipeline over the chunks: unit k of the stream = {fc1 slice of chunk k, fc2 slice of chunk k - 1}; while the matrix pipe runs
        // GEMM1(k), the wave's vector instructions do bias + GELU of chunk k - 1 (a few values behind every K-step's MFMAs), then GEMM2(k - 1):
        //   G1(0);   for k = 1 .. NCH - 1: [G1(k) | gelu(k - 1)]; G2(k - 1);   gelu(NCH - 1); G2(NCH - 1)
        f32x4 hacc[2][TB], hprev[2][TB];
        constexpr int NV = 8 * TB;       // GELU values per chunk and lane: i = 8 b + 4 a + r, in place in hprev[a][b][r]
        auto gelu_vals = [&](int i0, int i1, const float* bp) {   // bp = c_b1 + 32 c + 4 g (read where used: the biases are not kept in registers)
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const int b = i >> 3, a = (i >> 2) & 1, r = i & 3;
                hprev[a][b][r] = gelu_erf(hprev[a][b][r] * ws1 + bp[16 * a + r]);
            }
        };
        auto g1 = [&](const char* U1, bool with_gelu, const float* bv) {   // GEMM1 of this unit's chunk into hacc (+ the GELU of hprev)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b) hacc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            constexpr int P1 = TB == 2 ? 1 : 2;   // K-steps read ahead (TB = 2: a fragment feeds twice the MFMAs, and the registers are needed elsewhere)
            uint4 wr[P1 + 1][2][2];   // [set][hidden block][hi, lo]
            auto rd1 = [&](int ks, int set) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const char* r = U1 + (ks * 32 + 16 * a + lr) * 128;
                    wr[set][a][0] = *(const uint4*)(r + fo0);
                    wr[set][a][1] = *(const uint4*)(r + fo1);
                }
            };
#pragma unroll
            for (int i = 0; i < P1; ++i) rd1(i, i);
            // (fenced: left alone hipcc sinks every fragment read to just in front of its first MFMA and waits lgkmcnt(0) for it -- one exposed
            // LDS round trip per read, 22 cycles per MFMA instead of 17)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + P1 < KS) rd1(ks + P1, (ks + P1) % (P1 + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < TB; ++b)
                        Mma<T>::three(Mma<T>::as_u4v(wr[ks % (P1 + 1)][a][0]), Mma<T>::as_u4v(wr[ks % (P1 + 1)][a][1]), xH[ks][b], xL[ks][b], hacc[a][b]);
                if (with_gelu) gelu_vals(NV * ks / KS, NV * (ks + 1) / KS, bv);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto g2 = [&](const char* U2) {   // hprev (chunk k - 1, after its GELU) -> quartets -> GEMM2 into acc2
            // Output blocks in groups of GB = 2 / TB (two independent accumulators per group: a block's three products chain on ONE accumulator),
            // fragments PFG groups ahead, fenced like GEMM1's.
            constexpr int GB = 2 / TB, NG = NB2 / GB, PFG = 2;
            uint4 w2r[PFG + 1][GB][2];   // [set][block of the group][hi, lo]
            auto rd2 = [&](int grp, int set) {
#pragma unroll
                for (int i = 0; i < GB; ++i) {
                    const char* r = U2 + (16 * (grp * GB + i) + lr) * 128;
                    w2r[set][i][0] = *(const uint4*)(r + fo0);
                    w2r[set][i][1] = *(const uint4*)(r + fo1);
                }
            };
#pragma unroll
            for (int i = 0; i < PFG; ++i) rd2(i, i);
            U hH[TB], hL[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                // k-slot j = 4 a + r of this lane <-> hidden 32 c + 16 a + 4 g + r
                const float h0[4] = {hprev[0][b][0], hprev[0][b][1], hprev[0][b][2], hprev[0][b][3]};
                const float h1[4] = {hprev[1][b][0], hprev[1][b][1], hprev[1][b][2], hprev[1][b][3]};
                range_mask |= f16x2_out_of_range(h0) | f16x2_out_of_range(h1);
                const uint4 c0 = Chunk<T>::pack(h0), c1 = Chunk<T>::pack(h1);
                Mma<T>::regroup(c0, c1, hH[b], hL[b]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                if (grp + PFG < NG) rd2(grp + PFG, (grp + PFG) % (PFG + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GB; ++i)
#pragma unroll
                    for (int b = 0; b < TB; ++b)
                        Mma<T>::three(Mma<T>::as_u4v(w2r[grp % (PFG + 1)][i][0]), Mma<T>::as_u4v(w2r[grp % (PFG + 1)][i][1]), hH[b], hL[b], acc2[grp * GB + i][b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto keep = [&]() {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b) hprev[a][b] = hacc[a][b];
        };
        // the fc2 slice of a stream unit: behind its fc1 slice in the same ring unit, or (SPLIT) the next ring unit
        auto second = [&](const char* U1) -> const char* {
            if constexpr (SPLIT) {
                const char* u = next_unit();
                tick(0);
                return u;
            } else {
                return U1 + W1B;
            }
        };
        const float* const bias_g = c_b1 + 4 * g;
        {
            const char* const U1 = next_unit();
            tick(0);
            g1(U1, false, bias_g);
            tick(1);
            (void)second(U1);    // (the fc2 half of stream unit 0 is zeros: nothing to multiply yet)
        }
        keep();
        for (int k = 1; k < NCH; ++k) {
            const char* const U1 = next_unit();
            tick(0);
            g1(U1, true, bias_g + 32 * (k - 1));
            tick(1);
            g2(second(U1));
            tick(2);
            keep();
        }
        {
            const char* const U1 = next_unit();   // (its fc1 half is zeros)
            tick(0);
            gelu_vals(0, NV, bias_g + 32 * (NCH - 1));
            g2(second(U1));
            tick(2);
        }

        // ---- epilogue: x <- x + fc2(..) + b2, then the optional next norm / cast.  The MFMA leaves lane 16 g + lr with channels 16 a + 4 g .. + 4 of
        // token lr: four consecutive lanes then hold four DIFFERENT tokens, and a 16-byte access per lane touches four cache lines per lane quad
        // (the memory pipe works quad by quad: 4 x the requests for the same bytes; the residual loads, the x stores and the xn stores took 60 k of a
        // tile's 260 k cycles at D = 256).  One ds_bpermute per accumulator dword moves the values to lane 4 lr + g first: a quad is then one token's
        // 64 contiguous bytes, and everything after it (bias, residual, statistics over a quad, xn) happens in that layout.
        const int pn_tok = lane >> 2, pn_g = lane & 3;                      // this lane after the permutation: token pn_tok, channels 16 a + 4 pn_g .. + 4
        const int pn_src = 4 * (16 * pn_g + pn_tok);                        // ds_bpermute byte address of the lane that holds them now
#pragma unroll
        for (int b = 0; b < TB; ++b) {
#pragma unroll
            for (int a = 0; a < NB2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc2[a][b][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(pn_src, __float_as_int(acc2[a][b][r])));
            const int tok = tok0 + 16 * b + pn_tok;
            const bool ok = tok < p.M;
            float* xr = p.x + (size_t)(ok ? tok : 0) * D + 4 * pn_g;
            float sum = 0.f;
            // (all residual loads first and unconditional -- a token past M reads row 0 and stores nothing: under `if (ok)` every load got its own
            // branch and its own vmcnt(0), and x is read and written through the same pointer)
            float4 rvs[NB2];
#pragma unroll
            for (int a = 0; a < NB2; ++a) rvs[a] = (p.dbg & 4) ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(xr + 16 * a);
            if (p.prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tick(6); }
#pragma unroll
            for (int a = 0; a < NB2; ++a) {
                const float4 bv2 = *(const float4*)(c_b2 + 16 * a + 4 * pn_g);
                const float4 rv = rvs[a];
                f32x4 v = acc2[a][b];
                v[0] = v[0] * ws2 + bv2.x + rv.x; v[1] = v[1] * ws2 + bv2.y + rv.y; v[2] = v[2] * ws2 + bv2.z + rv.z; v[3] = v[3] * ws2 + bv2.w + rv.w;
                acc2[a][b] = v;
                if (ok && !(p.dbg & 1)) *(float4*)(xr + 16 * a) = make_float4(v[0], v[1], v[2], v[3]);
                sum += v[0] + v[1] + v[2] + v[3];
            }
            tick(7);
            if (p.xn) {
                float mean = 0.f, rstd = 1.f;
                if (next_ln) {      // a token's D values sit in the four lanes of a quad
                    sum += __shfl_xor(sum, 1);
                    sum += __shfl_xor(sum, 2);
                    mean = sum / (float)D;
                    float sq = 0.f;
#pragma unroll
                    for (int a = 0; a < NB2; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = acc2[a][b][r] - mean; sq += d * d; }
                    sq += __shfl_xor(sq, 1);
                    sq += __shfl_xor(sq, 2);
                    rstd = rsqrtf(sq / (float)D + 1e-5f);
                }
                tick(8);
                T* nr = (T*)p.xn + (size_t)(ok ? tok : 0) * D + 4 * pn_g;
#pragma unroll
                for (int a = 0; a < NB2; ++a) {
                    const float4 gv = *(const float4*)(c_ng + 16 * a + 4 * pn_g), bvn = *(const float4*)(c_nb + 16 * a + 4 * pn_g);
                    const f32x4 v = acc2[a][b];
                    const float o[4] = {(v[0] - mean) * rstd * gv.x + bvn.x, (v[1] - mean) * rstd * gv.y + bvn.y, (v[2] - mean) * rstd * gv.z + bvn.z,
                                        (v[3] - mean) * rstd * gv.w + bvn.w};
                    range_mask |= f16x2_out_of_range(o);
                    if (ok && !(p.dbg & 2)) *(uint4*)(nr + 16 * a) = Chunk<T>::pack(o);
                }
            }
        }
        tick(3);
    }
    wait_vm_only<0>();  // the ring's run-ahead fetches
    f16x2_raise(range_mask);
    if (p.prof && lane == 0)
        for (int k = 0; k < 9; ++k) atomicAdd(p.prof + k, (unsigned long long)tk[k]);
}

// ---------------------------------------------------------------- host: packing + launch
bool mlp_x2_eligible(int dtype, int D) {
    static const bool off = getenv("OCRVI_MLP_X2") && atoi(getenv("OCRVI_MLP_X2")) == 0;
    return !off && dtype == OCRVI_F16X2 && (D == 128 || D == 256 || D == 384);
}

// power of two that puts the largest |w| into [2^13, 2^14) (finish_pack's rule, host_util.hip)
static float x2_scale(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, fabsf(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    int e = 0;
    (void)frexpf(mx, &e);
    return ldexpf(1.0f, std::min(std::max(14 - e, -100), 100));
}

// Units in consumption order (unit k = fc1 slice of chunk k + fc2 slice of chunk k - 1, k = 0 .. 4 D / 32: the kernel's software pipeline),
// each in LDS image order: [KS][32 hidden rows][128 B] of fc1, then [D output rows][128 B] of fc2 with the 32
// hidden units of the chunk in the k-slot order the accumulator lanes produce (position 8 g + j <-> hidden 16 (j >> 2) + 4 g + (j & 3)).
// Every 128-byte row: (hi, lo) quartets (chunk 2 q = hi halves of k-slots 8 q .. 8 q + 7, chunk 2 q + 1 their lo halves), chunk ch stored at
// ch ^ swz128(row).  Tail: the two weight scales (fp32).
void pack_mlp_x2_stream(const float* w1, const float* w2, int D, std::vector<char>& out) {
    const int H4 = 4 * D, NCH = H4 / 32, KS = D / 32;
    const float s1 = x2_scale(w1, (size_t)H4 * D), s2 = x2_scale(w2, (size_t)D * H4);
    const size_t rows_per_unit = (size_t)KS * 32 + D, nrows = (size_t)(NCH + 1) * rows_per_unit;
    std::vector<float> buf(nrows * 32, 0.f);
    std::vector<int> rowidx(nrows);
    size_t o = 0, ri = 0;
    for (int k = 0; k <= NCH; ++k) {    // unit k: fc1 slice of chunk k (zeros for k = NCH), fc2 slice of chunk k - 1 (zeros for k = 0)
        for (int ks = 0; ks < KS; ++ks)
            for (int r = 0; r < 32; ++r) {
                rowidx[ri++] = ks * 32 + r;
                for (int e = 0; e < 32; ++e, ++o)
                    if (k < NCH) buf[o] = w1[(size_t)(32 * k + r) * D + 32 * ks + e] * s1;
            }
        for (int n = 0; n < D; ++n) {
            rowidx[ri++] = n;
            for (int pos = 0; pos < 32; ++pos, ++o) {
                const int g = pos >> 3, j = pos & 7;
                if (k >= 1) buf[o] = w2[(size_t)n * H4 + 32 * (k - 1) + 16 * (j >> 2) + 4 * g + (j & 3)] * s2;
            }
        }
    }
    std::vector<char> chunks(buf.size() * 4);
    convert_to_dtype(buf.data(), buf.size(), OCRVI_F16X2, chunks.data());     // [4 hi | 4 lo] per 4 consecutive elements
    out.assign(chunks.size() + 8, 0);
    for (size_t r = 0; r < nrows; ++r) {
        const uint64_t* src = (const uint64_t*)(chunks.data() + r * 128);      // 16 pieces of 8 bytes: (hi4, lo4) x 8 chunks
        uint64_t q[16];
        for (int k = 0; k < 4; ++k) {   // 32-byte group k: [hi4 lo4 | hi4' lo4'] -> [hi4 hi4' | lo4 lo4']
            q[4 * k] = src[4 * k]; q[4 * k + 1] = src[4 * k + 2]; q[4 * k + 2] = src[4 * k + 1]; q[4 * k + 3] = src[4 * k + 3];
        }
        const int row = rowidx[r], sw = ((row >> 1) & 1) | (((row >> 3) & 1) << 2);   // swz128
        uint64_t* dst = (uint64_t*)(out.data() + r * 128);
        for (int ch = 0; ch < 8; ++ch) {
            dst[2 * (ch ^ sw)] = q[2 * ch];
            dst[2 * (ch ^ sw) + 1] = q[2 * ch + 1];
        }
    }
    const float tail[2] = {1.0f / s1, 1.0f / s2};
    memcpy(out.data() + chunks.size(), tail, 8);
}

template <int D, int R, int TB, bool SPLIT>
static int launch_mlp_x2(const MlpX2Params& p, hipStream_t s) {
    const int smem = R * (SPLIT ? 128 : 256) * D + 9 * D * 4;
    auto kern = mlp_x2_kernel<D, R, TB, SPLIT>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const int ntiles = (p.M + 127) / 128;
    int grid = std::min(ntiles, n_cu);
    grid = cdiv(ntiles, cdiv(ntiles, grid));  // equal tile counts
    static const int stagger_env = getenv("OCRVI_MLPX2_STAGGER") ? atoi(getenv("OCRVI_MLPX2_STAGGER")) : -1;
    MlpX2Params ps = p;
    ps.stagger = stagger_env > 0 ? stagger_env : 0;   // (experiment, off: see the kernel)
    static const int dbg_env = getenv("OCRVI_MLPX2_DBG") ? atoi(getenv("OCRVI_MLPX2_DBG")) : 0;
    ps.dbg = dbg_env;
    static const bool prof = getenv("OCRVI_MLPX2_PROF") && atoi(getenv("OCRVI_MLPX2_PROF"));
    if (prof) {   // development: phase cycles, printed per launch (synchronises)
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) OCRVI_HIP(hipMalloc((void**)&dbuf, 72));
        OCRVI_HIP(hipMemsetAsync(dbuf, 0, 72, s));
        MlpX2Params q = ps;
        q.prof = dbuf;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512 / TB), smem, s, q);
        unsigned long long h[9];
        OCRVI_HIP(hipMemcpyAsync(h, dbuf, 72, hipMemcpyDeviceToHost, s));
        OCRVI_HIP(hipStreamSynchronize(s));
        const double wv = (8.0 / TB) * grid, units = (double)ntiles / grid * (4 * D / 32 + 1);
        fprintf(stderr, "mlp_x2 D %d TB %d split %d M %d grid %d: cycles per wave and stream unit: wait + barrier %.0f, GEMM1 (+ GELU) %.0f, GEMM2 %.0f; per tile: epilogue (x loads %.0f, x update + stores %.0f, statistics %.0f, xn %.0f), prologue (loads + LayerNorm) %.0f, drain %.0f\n", D, TB,
                (int)SPLIT, p.M, grid, h[0] / wv / units, h[1] / wv / units, h[2] / wv / units, h[6] / wv / ((double)ntiles / grid), h[7] / wv / ((double)ntiles / grid), h[8] / wv / ((double)ntiles / grid), h[3] / wv / ((double)ntiles / grid),
                h[4] / wv / ((double)ntiles / grid), h[5] / wv / ((double)ntiles / grid));
        return OCRVI_OK;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512 / TB), smem, s, ps);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

int k_mlp_x2(float* x, void* xn, const float* ln_g, const float* ln_b, const float* next_g, const float* next_b, const void* wstream, const float* b1,
             const float* b2, int M, int D, hipStream_t s) {
    OCRVI_CHECK((D == 128 || D == 256 || D == 384) && x && ln_g && ln_b && wstream && b1 && b2 && M > 0 && M < (1 << 24), OCRVI_EINVAL,
                "mlp_x2: bad argument (D %d, M %d)", D, M);
    MlpX2Params p;
    p.x = x; p.xn = xn; p.ln_g = ln_g; p.ln_b = ln_b; p.next_g = next_g; p.next_b = next_b; p.wstream = (const char*)wstream; p.b1 = b1; p.b2 = b2; p.M = M;
    char tag[64];
    snprintf(tag, sizeof(tag), "mlp_fused_d%d_f16x2", D);
    ProfScope ps(tag, 2.0 * M * 8.0 * D * D, (double)M * D * (8.0 + (xn ? 4.0 : 0.0)) + 8.0 * D * D * 4.0, s);
    static const int tb2 = getenv("OCRVI_MLPX2_TB2") ? atoi(getenv("OCRVI_MLPX2_TB2")) : 0;   // (A/B: the 4-wave layout at D <= 256 too)
    if (D == 128) return tb2 ? launch_mlp_x2<128, 8, 2, true>(p, s) : launch_mlp_x2<128, 4, 1, false>(p, s);
    if (D == 256) return tb2 ? launch_mlp_x2<256, 4, 2, true>(p, s) : launch_mlp_x2<256, 2, 1, false>(p, s);
    return launch_mlp_x2<384, 3, 2, true>(p, s);
}

}  // namespace ocrvi
