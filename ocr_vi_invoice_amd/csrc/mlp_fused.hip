// Fused MixingBlock MLP for the 16-bit modes (svtrv2.py:28-39,100):   x <- x + fc2(gelu(fc1(LayerNorm(x))))   [+ the NEXT LayerNorm]
//
// Unfused, the 4*D hidden activation makes a round trip through HBM (189 MB per 256 crops at D = 384) and LayerNorm is a kernel of
// its own; the two GEMMs are then bound by the latency of their operand streams.  Here one workgroup owns 128
// tokens for the whole block:
//   * the tokens' LayerNorm'ed activations are built once, in REGISTERS, as MFMA B-operand fragments (fp32 x -> statistics over the
//     4 lanes that share a token -> 16-bit), D/4 VGPRs per lane;
//   * the hidden dimension is walked in chunks of 64: GEMM1 (64 x D slice of fc1) -> bias + exact GELU in registers -> the fp32
//     accumulators become, 16-bit packed, the B operand of GEMM2 (D x 64 slice of fc2) with no LDS round trip: an accumulator lane
//     holds 4 consecutive hidden units of one token, and fc2's columns are permuted at pack time so that those are exactly the
//     k-slots the lane owns in the next MFMA (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand");
//   * fc2 accumulates all D outputs of a wave's tokens in registers (D/4 VGPRs for the shipped 16 tokens per wave, two waves per SIMD;
//     D/2 for the first layout, 32 tokens per wave with one wave per SIMD owning the 512-entry file -- template parameter TB);
//   * the only streamed operand is the weights, identical for every tile: they are packed at load time into 16-KiB "pieces" in LDS
//     image order and DMA'd (global_load_lds_dwordx4) into a ring that runs ahead across chunk and tile boundaries, counted vmcnt,
//     one barrier per piece (the gemm_ring protocol with a deeper ring);
//   * epilogue: + bias + fp32 residual -> x; optionally LayerNorm of the result with the NEXT block's norm1 (or a plain cast) -> xn,
//     which removes that LayerNorm / cast kernel too.
// Per token tile the CU streams 4*D/64 * 256*D bytes of weights from L2 (2.36 MB at D = 384) for 2*128*8*D*D FLOP: 131 FLOP/B.
#include <stdlib.h>
#include <string.h>

#include <utility>

#include "gemm_ring.h"
#include "kernels.h"

namespace ocrvi {

struct MlpParams {
    float* x = nullptr;            // [M][D] fp32 residual stream, updated in place
    void* xn = nullptr;            // optional [M][D] T: LayerNorm_next(x_new) (next_g != null) or T(x_new) (next_g == null)
    const float* ln_g = nullptr;   // norm2 of this block
    const float* ln_b = nullptr;
    const float* next_g = nullptr;
    const float* next_b = nullptr;
    const void* wstream = nullptr; // packed pieces (pack_mlp_stream)
    const float* b1 = nullptr;     // [4D]
    const float* b2 = nullptr;     // [D]
    int M = 0;
};

constexpr int kPiece = 16384;

template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    union { T h[2]; uint32_t u; } r;
    r.h[0] = (T)a; r.h[1] = (T)b;
    return r.u;
}

template <int I> struct ICm { static constexpr int value = I; };
template <typename F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ICm<Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Software pipeline over the hidden chunks (one wave per SIMD cannot overlap its own GELU with its own MFMAs unless the instruction
// stream interleaves them): while chunk c's bias + GELU runs on the VALU, the matrix pipe works on GEMM1 of chunk c + 1 and GEMM2 of
// chunk c - 1.  Two GEMM1 accumulator sets and two packed-h sets alternate (chunk parity), so every register index is static.
//   G1(0);  [G1(1) | gelu_a(0)];  gelu_b(0);
//   for c = 1 .. NCH-2:  [G1(c+1) | gelu_a(c)];  [G2(c-1) | gelu_b(c)];
//   gelu_a(NCH-1);  [G2(NCH-2) | gelu_b(NCH-1)];  G2(NCH-1)
// The weight stream is packed in exactly that order (pack_mlp_stream).
// TB = 16-token blocks per wave: 2 -> 4 waves of 32 tokens (one per SIMD, the whole register file each); 1 -> 8 waves of 16 tokens (two
// per SIMD, 256 registers each: one wave's LDS / barrier waits and GELU fall under the other's MFMAs, at twice the LDS fragment traffic).
template <typename T, int D, int WPS, int R, int TB = 2>
__global__ __launch_bounds__(512 / TB, WPS) void mlp_fused_kernel(const MlpParams p) {
    constexpr int NWV = 8 / TB, NT = 64 * NWV;   // waves, threads
    constexpr int IPW = 16 / NWV;                  // DMA instructions per wave per 16-KiB piece
    constexpr int KS = D / 64;         // 128-byte K-steps of GEMM1
    constexpr int NP = D / 128;        // pieces per chunk and GEMM (W1: 2 K-steps x 64 rows; W2: 128 rows x 1 K-step)
    constexpr int NCH = 4 * D / 64;    // hidden chunks of 64
    constexpr int PPT = NCH * 2 * NP;  // pieces per token tile
    constexpr int UNIT = NP * kPiece;  // one chunk-GEMM's weights: the ring's sync unit (one wait + barrier per unit, not per piece)
    constexpr int PF = R - 1;          // units in flight
    constexpr int NB2 = D / 16;        // 16-channel output blocks
    constexpr int GR = 4 * NP;         // groups of 8 MFMAs in one GEMM part
    static_assert(D % 128 == 0 && D <= 384 && NCH % 2 == 0 && NCH >= 4, "D");
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
    __syncthreads();  // (also drains those loads: no VMEM op is in flight when the ring starts)

    const int ntiles = (p.M + 127) >> 7;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;

    // ---- weight ring: piece q of the stream is piece (q % PPT) of the packed buffer; this wave issues DMA instructions wave, wave + 4,
    // wave + 8, wave + 12 of its 16 (8 row-slots x 128 B each); the XOR swizzle is applied on the source chunk (rule 21).  The ring
    // step is branch-free: past the end of the workgroup's stream it keeps fetching pieces nobody reads (their slots are free), which
    // are drained before the kernel ends.
    const int prow = lane >> 3;
    const unsigned voff = (unsigned)((8 * wave + prow) * 128 + (((lane & 7) ^ swz128(8 * wave + prow)) << 4));
    const char* const wbase = uniform_ptr((const char*)p.wstream);
    constexpr int UPT = PPT / NP;      // units per token tile
    int prod_slot = 0, prod_mod = 0, cons_slot = 0;
    auto issue_unit = [&]() {
        const char* src = wbase + (size_t)prod_mod * UNIT;
        const unsigned dst = lds0 + prod_slot * UNIT + wave * 1024;
#pragma unroll
        for (int j = 0; j < IPW * NP; ++j) glds16(src + j * (NWV * 1024), voff, __builtin_amdgcn_readfirstlane(dst + j * (NWV * 1024)));
        prod_slot = prod_slot + 1 == R ? 0 : prod_slot + 1;
        prod_mod = prod_mod + 1 == UPT ? 0 : prod_mod + 1;
    };
    auto next_unit = [&]() -> const char* {  // wait for the oldest unit in flight, free the slot before it, keep the ring full
        wait_vm_barrier<(PF - 1) * IPW * NP>();
        issue_unit();
        const char* s = smem + cons_slot * UNIT;
        cons_slot = cons_slot + 1 == R ? 0 : cons_slot + 1;
        return s;
    };
    for (int i = 0; i < PF; ++i) issue_unit();

    const int swa = swz128(lr);
    const int fo0 = ((2 * g) ^ swa) << 4, fo1 = ((2 * g + 1) ^ swa) << 4;

    for (int t = 0; t < my_tiles; ++t) {
        const int tile = (int)blockIdx.x + t * G;
        const int tok0 = tile * 128 + wave * (16 * TB);
        // ---- prologue: LayerNorm(x) of this wave's 32 tokens -> B-operand fragments.  Lane (lr, g) of token block b owns channels
        // 64 ks + 16 g .. + 16 of token tok0 + 16 b + lr for every K-step ks: its two 8-element halves are the two MFMA k-slots.
        uint4 xf[KS][2][TB];  // [ks][half][token block]
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            float xv[KS][16];
            const int tok = min(tok0 + 16 * b + lr, p.M - 1);
            const float* xr = p.x + (size_t)tok * D + 16 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *(const float4*)(xr + 64 * ks + 4 * q);
                    xv[ks][4 * q] = v.x; xv[ks][4 * q + 1] = v.y; xv[ks][4 * q + 2] = v.z; xv[ks][4 * q + 3] = v.w;
                }
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += xv[ks][e];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum / (float)D;
            float sq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = xv[ks][e] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            const float rstd = rsqrtf(sq / (float)D + 1e-5f);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float o[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 gv = *(const float4*)(c_g + 64 * ks + 16 * g + 4 * q), bv = *(const float4*)(c_be + 64 * ks + 16 * g + 4 * q);
                    o[4 * q] = (xv[ks][4 * q] - mean) * rstd * gv.x + bv.x;
                    o[4 * q + 1] = (xv[ks][4 * q + 1] - mean) * rstd * gv.y + bv.y;
                    o[4 * q + 2] = (xv[ks][4 * q + 2] - mean) * rstd * gv.z + bv.z;
                    o[4 * q + 3] = (xv[ks][4 * q + 3] - mean) * rstd * gv.w + bv.w;
                }
                xf[ks][0][b] = make_uint4(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7]));
                xf[ks][1][b] = make_uint4(pack2<T>(o[8], o[9]), pack2<T>(o[10], o[11]), pack2<T>(o[12], o[13]), pack2<T>(o[14], o[15]));
            }
        }
        // every VMEM op issued so far by this wave (ring DMAs, the previous tile's stores, the loads above) has completed: the counted
        // waits of the main loop start from the DMAs issued from here on (any older piece has landed)
        wait_vm_only<0>();

        f32x4 acc2[NB2][TB];
#pragma unroll
        for (int a = 0; a < NB2; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 accA[4][TB], accB[4][TB];
        uint32_t hfA[2][TB][4], hfB[2][TB][4];   // [k-slot half][token block][word]
        constexpr int PH = 4 * TB;               // GELU pairs per k-slot half

        // bias + exact GELU of pair `PP` (0..15) of chunk c: half hh = PP >> 3 (the k-slot half of GEMM2 it lands in), token block
        // b = (PP >> 2) & 1, word jp = PP & 3  <->  hidden 16 (2 hh + (jp >> 1)) + 4 g + 2 (jp & 1) + {0, 1}
        auto gelu_pair = [&](auto PP, const f32x4 (&acc)[4][TB], uint32_t (&hf)[2][TB][4], int c) {
            constexpr int pp = decltype(PP)::value, hh = pp / PH, b = (pp >> 2) % TB, jp = pp & 3, a = 2 * hh + (jp >> 1), r0 = 2 * (jp & 1);
            const float2 bv = *(const float2*)(c_b1 + 64 * c + 16 * a + 4 * g + r0);
            hf[hh][b][jp] = pack2<T>(gelu_erf(acc[a][b][r0] + bv.x), gelu_erf(acc[a][b][r0 + 1] + bv.y));
        };
        // GELU pairs [8 HALF + 8 k / GR, 8 HALF + 8 (k + 1) / GR) ride behind MFMA group k of a part
        auto gelu_slice = [&](auto HALF, auto K, const f32x4 (&acc)[4][TB], uint32_t (&hf)[2][TB][4], int c) {
            constexpr int half = decltype(HALF)::value, k = decltype(K)::value;
            if constexpr (half >= 0) {
                constexpr int lo = PH * k / GR, hi = PH * (k + 1) / GR;
                static_for<hi - lo>([&](auto J) { gelu_pair(ICm<PH * half + lo + decltype(J)::value>{}, acc, hf, c); });
            }
        };
        // GEMM1 of one chunk into `dst` (its NP pieces come next in the stream); GELU half HALF (-1: none) of chunk c from `src` rides along
        auto g1 = [&](f32x4 (&dst)[4][TB], auto HALF, const f32x4 (&src)[4][TB], uint32_t (&hf)[2][TB][4], int c) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b) dst[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char* const U1 = next_unit();
            static_for<NP>([&](auto P1) {
                constexpr int p1 = decltype(P1)::value;
                const char* S = U1 + p1 * kPiece;
                // fragments of group q = kk * 2 + h are read one group ahead of their MFMAs (two register sets)
                uint4 wf[2][4];
                auto rd = [&](auto Q) {
                    constexpr int q = decltype(Q)::value;
#pragma unroll
                    for (int a = 0; a < 4; ++a) wf[q & 1][a] = *(const uint4*)(S + ((q >> 1) * 64 + a * 16 + lr) * 128 + ((q & 1) ? fo1 : fo0));
                };
                rd(ICm<0>{});
                static_for<4>([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    if constexpr (q + 1 < 4) rd(ICm<q + 1>{});
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < TB; ++b) Mma<T>::half(wf[q & 1][a], xf[2 * p1 + (q >> 1)][q & 1][b], dst[a][b]);
                    gelu_slice(HALF, ICm<4 * p1 + q>{}, src, hf, c);
                });
            });
        };
        // GEMM2 of one chunk (h = `hin`) into acc2; GELU half HALF of chunk c from `src` into `hf` rides along
        auto g2 = [&](const uint32_t (&hin)[2][TB][4], auto HALF, const f32x4 (&src)[4][TB], uint32_t (&hf)[2][TB][4], int c) {
            uint4 hv[2][TB];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < TB; ++b) hv[h][b] = make_uint4(hin[h][b][0], hin[h][b][1], hin[h][b][2], hin[h][b][3]);
            const char* const U2 = next_unit();
            static_for<NP>([&](auto P2) {
                constexpr int p2 = decltype(P2)::value;
                const char* S = U2 + p2 * kPiece;
                uint4 wf[2][4];  // group q = h * 2 + (output blocks 0-3 | 4-7), read one group ahead
                auto rd = [&](auto Q) {
                    constexpr int q = decltype(Q)::value;
#pragma unroll
                    for (int a = 0; a < 4; ++a) wf[q & 1][a] = *(const uint4*)(S + (((q & 1) * 4 + a) * 16 + lr) * 128 + ((q >> 1) ? fo1 : fo0));
                };
                rd(ICm<0>{});
                static_for<4>([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    if constexpr (q + 1 < 4) rd(ICm<q + 1>{});
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < TB; ++b) Mma<T>::half(wf[q & 1][a], hv[q >> 1][b], acc2[p2 * 8 + (q & 1) * 4 + a][b]);
                    gelu_slice(HALF, ICm<4 * p2 + q>{}, src, hf, c);
                });
            });
        };
        auto gelu_only = [&](auto HALF, const f32x4 (&src)[4][TB], uint32_t (&hf)[2][TB][4], int c) {
            static_for<PH>([&](auto J) { gelu_pair(ICm<PH * decltype(HALF)::value + decltype(J)::value>{}, src, hf, c); });
        };
        using NONE = ICm<-1>;
        g1(accA, NONE{}, accA, hfA, 0);                 // G1(0)
        g1(accB, ICm<0>{}, accA, hfA, 0);               // G1(1) | gelu_a(0)
        gelu_only(ICm<1>{}, accA, hfA, 0);              // gelu_b(0)
        for (int c = 1; c <= NCH - 3; c += 2) {
            g1(accA, ICm<0>{}, accB, hfB, c);           // G1(c+1) | gelu_a(c)        (c odd: its GEMM1 sits in accB)
            g2(hfA, ICm<1>{}, accB, hfB, c);            // G2(c-1) | gelu_b(c)
            g1(accB, ICm<0>{}, accA, hfA, c + 1);       // G1(c+2) | gelu_a(c+1)
            g2(hfB, ICm<1>{}, accA, hfA, c + 1);        // G2(c)   | gelu_b(c+1)
        }
        gelu_only(ICm<0>{}, accB, hfB, NCH - 1);        // gelu_a(NCH-1)
        g2(hfA, ICm<1>{}, accB, hfB, NCH - 1);          // G2(NCH-2) | gelu_b(NCH-1)
        g2(hfB, NONE{}, accB, hfB, NCH - 1);            // G2(NCH-1)

        // ---- epilogue: x <- x + fc2(..) + b2 (lane: channels 16 a + 4 g .. + 4 of token tok0 + 16 b + lr), then the optional next norm
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int tok = tok0 + 16 * b + lr;
            const bool ok = tok < p.M;
            float* xr = p.x + (size_t)(ok ? tok : 0) * D + 4 * g;
            float sum = 0.f;
            // (all residual loads first: x is read and written through the same pointer, so inside one loop every load would wait behind the
            // previous block's store -- NB2 serial round trips to memory)
            float4 rvs[NB2];
#pragma unroll
            for (int a = 0; a < NB2; ++a) rvs[a] = *(const float4*)(xr + 16 * a);   // (unconditional: a token past M reads row 0 and stores nothing --
                                                                                    // under `if (ok)` every load got its own branch and its own vmcnt(0))
#pragma unroll
            for (int a = 0; a < NB2; ++a) {
                const float4 bv = *(const float4*)(c_b2 + 16 * a + 4 * g);
                const float4 rv = rvs[a];
                f32x4 v = acc2[a][b];
                v[0] += bv.x + rv.x; v[1] += bv.y + rv.y; v[2] += bv.z + rv.z; v[3] += bv.w + rv.w;
                acc2[a][b] = v;
                if (ok) *(float4*)(xr + 16 * a) = make_float4(v[0], v[1], v[2], v[3]);
                sum += v[0] + v[1] + v[2] + v[3];
            }
            if (p.xn) {
                float mean = 0.f, rstd = 1.f;
                if (next_ln) {
                    sum += __shfl_xor(sum, 16);
                    sum += __shfl_xor(sum, 32);
                    mean = sum / (float)D;
                    float sq = 0.f;
#pragma unroll
                    for (int a = 0; a < NB2; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = acc2[a][b][r] - mean; sq += d * d; }
                    sq += __shfl_xor(sq, 16);
                    sq += __shfl_xor(sq, 32);
                    rstd = rsqrtf(sq / (float)D + 1e-5f);
                }
                T* nr = (T*)p.xn + (size_t)(ok ? tok : 0) * D + 4 * g;
#pragma unroll
                for (int a = 0; a < NB2; ++a) {
                    const float4 gv = *(const float4*)(c_ng + 16 * a + 4 * g), bv = *(const float4*)(c_nb + 16 * a + 4 * g);
                    const f32x4 v = acc2[a][b];
                    uint2 o;
                    o.x = pack2<T>((v[0] - mean) * rstd * gv.x + bv.x, (v[1] - mean) * rstd * gv.y + bv.y);
                    o.y = pack2<T>((v[2] - mean) * rstd * gv.z + bv.z, (v[3] - mean) * rstd * gv.w + bv.w);
                    if (ok) *(uint2*)(nr + 16 * a) = o;
                }
            }
        }
    }
    wait_vm_only<0>();  // the ring's run-ahead fetches
}

// ---------------------------------------------------------------- host: packing + launch
// Pieces: a chunk's fc1 slice is NP W1 pieces (piece p1 = K-steps 2 p1 and 2 p1 + 1, each 64 rows [hidden 64 hc + r] x 64
// elements of K), its fc2 slice NP W2 pieces (piece p2 = output rows 128 p2 .. + 128, each 64 elements = hidden chunk hc in the k-slot order
// the accumulator lanes produce: position 16 g + 8 h + j  <->  hidden 16 (2 h + (j >> 2)) + 4 g + (j & 3)).
void pack_mlp_stream(const float* w1, const float* w2, int D, int dtype, std::vector<char>& out) {
    if (dtype == OCRVI_F16X2) return pack_mlp_x2_stream(w1, w2, D, out);
    const int H4 = 4 * D, NCH = H4 / 64, NP = D / 128;
    const size_t esz = dtype_size(dtype);
    std::vector<float> buf((size_t)NCH * 2 * NP * (kPiece / esz));
    size_t o = 0;
    auto put_w1 = [&](int hc) {
        for (int p1 = 0; p1 < NP; ++p1)
            for (int kk = 0; kk < 2; ++kk)
                for (int r = 0; r < 64; ++r)
                    for (int e = 0; e < 64; ++e) buf[o++] = w1[(size_t)(64 * hc + r) * D + 64 * (2 * p1 + kk) + e];
    };
    auto put_w2 = [&](int hc) {
        for (int p2 = 0; p2 < NP; ++p2)
            for (int n = 0; n < 128; ++n)
                for (int pos = 0; pos < 64; ++pos) {
                    const int g = pos >> 4, h = (pos >> 3) & 1, j = pos & 7;
                    const int hid = 16 * (2 * h + (j >> 2)) + 4 * g + (j & 3);
                    buf[o++] = w2[(size_t)(128 * p2 + n) * H4 + 64 * hc + hid];
                }
    };
    // consumption order of the software pipeline: G1(0), G1(1), then G1(c+1), G2(c-1) for c = 1 .. NCH-2, then G2(NCH-2), G2(NCH-1)
    put_w1(0);
    put_w1(1);
    for (int c = 1; c <= NCH - 2; ++c) {
        put_w1(c + 1);
        put_w2(c - 1);
    }
    put_w2(NCH - 2);
    put_w2(NCH - 1);
    out.resize(buf.size() * esz);
    convert_to_dtype(buf.data(), buf.size(), dtype, out.data());
}

bool mlp_fused_eligible(int dtype, int D) {
    if (dtype == OCRVI_F16X2) return mlp_x2_eligible(dtype, D);
    return (dtype == OCRVI_BF16 || dtype == OCRVI_F16) && D % 128 == 0 && D >= 128 && D <= 384;
}

template <typename T, int D, int WPS, int R, int TB = 2>
static int launch_mlp(const MlpParams& p, hipStream_t s) {
    const int smem = R * (D / 128) * kPiece + (4 * D + 5 * D) * 4;
    auto kern = mlp_fused_kernel<T, D, WPS, R, TB>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const int ntiles = (p.M + 127) / 128;
    int grid = std::min(ntiles, n_cu * WPS);
    grid = cdiv(ntiles, cdiv(ntiles, grid));  // equal tile counts
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512 / TB), smem, s, p);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T> static int mlp_dt(const MlpParams& p, int D, hipStream_t s) {
    // 8 waves of 16 tokens (two per SIMD: TB = 1) beat 4 waves of 32 (one per SIMD, the layout the register budget was first designed
    // around) by 8-9 % at every D (MI355X, 256 crops: 223 -> 202, 284 -> 266, 271 -> 248 us): a wave's ring waits and its GELU stretches
    // fall under the other wave's MFMAs, which outweighs reading every weight fragment from LDS twice as often.  The 4-wave layout is no
    // longer instantiated: its D = 384 build spilled 25 VGPRs into a kernel whose LDS-DMA ring is synchronised by hand-counted vmcnt.
    switch (D) {
        case 128: return launch_mlp<T, 128, 1, 9, 1>(p, s);    // ring: 9 units of 16 KiB
        case 256: return launch_mlp<T, 256, 1, 4, 1>(p, s);    //       4 units of 32 KiB
        case 384: return launch_mlp<T, 384, 1, 3, 1>(p, s);    //       3 units of 48 KiB
    }
    set_error("mlp_fused: D=%d unsupported", D);
    return OCRVI_EINVAL;
}

int k_mlp_fused(int dtype, float* x, void* xn, const float* ln_g, const float* ln_b, const float* next_g, const float* next_b, const void* wstream,
                const float* b1, const float* b2, int M, int D, hipStream_t s) {
    OCRVI_CHECK(mlp_fused_eligible(dtype, D) && x && ln_g && ln_b && wstream && b1 && b2 && M > 0 && M < (1 << 24), OCRVI_EINVAL,
                "mlp_fused: bad argument (dtype %d, D %d, M %d)", dtype, D, M);
    if (dtype == OCRVI_F16X2) return k_mlp_x2(x, xn, ln_g, ln_b, next_g, next_b, wstream, b1, b2, M, D, s);
    MlpParams p;
    p.x = x; p.xn = xn; p.ln_g = ln_g; p.ln_b = ln_b; p.next_g = next_g; p.next_b = next_b; p.wstream = wstream; p.b1 = b1; p.b2 = b2; p.M = M;
    char tag[64];
    snprintf(tag, sizeof(tag), "mlp_fused_d%d_%s", D, dtype_name(dtype));
    const double esz = (double)dtype_size(dtype);
    ProfScope ps(tag, 2.0 * M * 8.0 * D * D, (double)M * D * (8.0 + (xn ? esz : 0.0)) + 8.0 * D * D * esz, s);
    if (dtype == OCRVI_BF16) return mlp_dt<bf16_t>(p, D, s);
    return mlp_dt<f16_t>(p, D, s);
}

}  // namespace ocrvi
