// Fused multi-head self-attention for head_dim = 32 (always the case in this model: heads = dim/32,
// svtrv2.py:70-72,169-171).  One workgroup = one (sequence, head): K [N][32] and V^T [32][N] of the head are
// staged once into LDS (N <= 512: 480 tokens in stage 1, 240 in stage 2, 80 per FRM row at 48x320 input).
// Each wave then takes 16-query tiles:
//   S^T = K Q^T  (MFMA A = K rows, B = Q)  -> accumulator column = query, rows = keys: a query's whole score row
//                                             sits in ONE lane column (4 lanes x 4*NT registers) -> 2 shuffles per reduce
//   softmax in registers (exp2 with the 1/sqrt(32) scale folded in), un-normalised P kept in the accumulators
//   O^T = V^T P^T (MFMA A = V^T rows = head-dim, B = P straight from the accumulators; the K-order of the product is
//                  permuted identically on both operands, cdna_hip_programming.md section 3)
#include <stdlib.h>

#include "kernels.h"

namespace ocrvi {
OCRVI_RANGE_FLAG_TU()   // binds this unit's f16x2 range-flag pointer (common.h)


template <typename T> struct AttnCfg;
template <> struct AttnCfg<float> {
    static constexpr int KROW = 128;
    static constexpr __host__ __device__ int vstride(int NP) { return NP * 4 + 16; }
};
template <> struct AttnCfg<bf16_t> {
    static constexpr int KROW = 64;
    static constexpr __host__ __device__ int vstride(int NP) { return (NP * 2 + 255) / 256 * 256 + 16; }  // = 16 B mod 256 B: conflict-free b64 column reads
};
template <> struct AttnCfg<f16_t> : AttnCfg<bf16_t> {};
template <> struct AttnCfg<f16x2_t> : AttnCfg<float> {};   // 4-byte elements: the fp32 LDS geometry, byte for byte

__device__ __forceinline__ int swz64(int row) { return (-(row >> 2)) & 3; }

template <typename T> __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<bf16_t>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<f16_t>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two fp32 values -> their packed fp16 hi halves and packed fp16 lo halves (round to nearest even both): 4 vector instructions per pair.
// (the conversions are left to the compiler -- v_cvt_pk_f16_f32 -- and not written as inline asm: `a` and `b` come straight from v_exp_f32,
// and the wait state a transcendental result needs before a vector instruction reads it is only inserted for instructions hipcc can see)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& H, unsigned& L) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    H = __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){a, b}, h2v));
    const float la = fma_mix_lo(-1.0f, H, a), lb = fma_mix_hi(-1.0f, H, b);   // exact in fp32
    L = __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){la, lb}, h2v));
}

// MAXT = number of 16-key tiles held in registers; MASK = false when N == 16*MAXT exactly (480 / 240 / 80 tokens at 48x320 input):
// then no key masking is generated at all.
// NWV = waves per workgroup (they share the staged K / V^T): 8 for the fp32 480-key case, whose 123 KB of LDS allow one workgroup per CU --
// with 4 waves that is one wave per SIMD and nothing to hide a wave's LDS latency and softmax behind.
// (f16x2 up to 256 keys: 8 waves per workgroup at no more than 128 registers, so that the two workgroups a CU's LDS holds give every SIMD four waves:
// a wave's QK -> softmax -> PV chain is latency-bound and needs neighbours)
template <typename T, int MAXT, bool MASK, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, (IsSplit<T>::value && MAXT <= 16 && NWV == 8) ? 4 : 1) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int heads, int kbase, int Nk,
                                                                                                               float* __restrict__ po, float2* __restrict__ pml) {
    // (kbase, Nk): the key range [kbase, kbase + Nk) this launch attends to -- (0, N) normally.  Sequences beyond one workgroup's LDS
    // (4-byte types: > 512 keys) run as several launches over key chunks in PARTIAL mode (po != null): the un-normalised output rows go to
    // po [B][N][D] fp32 and (row max, row sum) to pml [B][heads][N]; attention_combine_kernel merges the chunks (flash-attention's split-K).
    constexpr int EPC = TypeInfo<T>::EPC;
    constexpr int KROW = AttnCfg<T>::KROW;
    constexpr int CH = KROW / 16;  // 16-byte chunks per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // heads of one sequence run back to back on one XCD: the 64-byte q/k/v slices of neighbouring heads share 128-byte lines in its L2
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int h = L % heads, b = L / heads;
    const int D = heads * 32, ld = 3 * D;
    constexpr int NP = ((MAXT + 1) / 2) * 32;  // key rows staged in LDS: whole 32-key PV steps (zero-filled past N)
    constexpr int VS = AttnCfg<T>::vstride(NP);
    char* Ks = smem;
    char* Vt = smem + (size_t)NP * KROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* base = qkv + (size_t)b * N * ld + h * 32;

    // ---- stage K (row-major, swizzled) and V^T
    for (int idx = tid; idx < NP * CH; idx += NWV * 64) {
        const int key = idx / CH, ch = idx % CH;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < Nk) {
            kv = *(const uint4*)(base + (size_t)(kbase + key) * ld + D + ch * EPC);
            vv = *(const uint4*)(base + (size_t)(kbase + key) * ld + 2 * D + ch * EPC);
        }
        const int sw = sizeof(T) == 4 ? swz128(key) : swz64(key);
        if constexpr (IsSplit<T>::value) {
            // f16x2: both operands are staged in the form the three-product MFMAs consume -- per lane 32 bytes = [hi of its 8 k-slots | lo of
            // them] -- so that a fragment is two ds_read_b128 and NO register regrouping (the K and V fragments of a query tile cost 6 v_mov
            // each in the chunk format: 480 of the ~1000 VALU instructions per 16 queries at 240 keys, in a kernel that is VALU-bound).
            // K row `key`: chunk 2 j holds the hi halves of channels 8 j .. 8 j + 7, chunk 2 j + 1 their lo halves: the hi (lo) half of memory
            // chunk ch goes to bytes 8 (ch & 1) of LDS chunk (ch & ~1) (+ 1).
            char* kr = Ks + key * KROW + (ch & 1) * 8;
            *(uint2*)(kr + (((ch & ~1) ^ sw) << 4)) = make_uint2(kv.x, kv.y);
            *(uint2*)(kr + (((ch | 1) ^ sw) << 4)) = make_uint2(kv.z, kv.w);
            // V^T row d: 128 bytes per 32-key step s, lane group g at 32 g: [hi of keys 32 s + 4 g .. + 3 and 32 s + 16 + 4 g .. + 3 | lo of them]
            union { uint4 u; f16_t h[8]; } r;
            r.u = vv;
            const int vo = (key >> 5) * 128 + ((key >> 2) & 3) * 32 + (((key >> 4) & 1) * 4 + (key & 3)) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                char* dst = Vt + (ch * 4 + e) * VS + vo;
                *(f16_t*)dst = r.h[e];
                *(f16_t*)(dst + 16) = r.h[4 + e];
            }
        } else {
            *(uint4*)(Ks + key * KROW + ((ch ^ sw) << 4)) = kv;
            union { uint4 u; T e[EPC]; } r;
            r.u = vv;
#pragma unroll
            for (int e = 0; e < EPC; ++e) *(T*)(Vt + (ch * EPC + e) * VS + key * (int)sizeof(T)) = r.e[e];
        }
    }
    __syncthreads();

    const int lr = lane & 15, g = lane >> 4;
    const float c2 = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 * log2(e)
    const int NQ = (N + 15) >> 4;
    for (int qt = wave; qt < NQ; qt += NWV) {
        const int q = qt * 16 + lr;
        const bool qok = q < N;
        f32x4 acc[MAXT];
        // K / V^T fragments are invariant across q-tiles; left alone, the compiler hoists all of them into registers (240 VGPRs at
        // 480 keys -> 1 wave per SIMD).  An opaque zero offset per q-tile keeps them as LDS reads inside the loop.
        int lds_off = 0;
        if (MAXT > 8) asm volatile("" : "+v"(lds_off));
        const char* Ksq = Ks + lds_off;
        const char* Vtq = Vt + lds_off;
        if constexpr (sizeof(T) == 4) {
            uint4 qf[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            if (qok) {
                qf[0] = *(const uint4*)(base + (size_t)q * ld + 8 * g);
                qf[1] = *(const uint4*)(base + (size_t)q * ld + 8 * g + 4);
            }
            const int sw = swz128(lr);
            const int o0 = ((2 * g) ^ sw) << 4, o1 = ((2 * g + 1) ^ sw) << 4;
            if constexpr (IsSplit<T>::value) {   // three MFMAs per 16-key tile: (hi, lo) quartets of the 8 head-dim slots of this lane
                typedef typename Mma<T>::u4v U;
                U qH, qL;
                Mma<T>::regroup(qf[0], qf[1], qH, qL);
                // K fragments (staged as (hi, lo) quartets) one key tile ahead of their MFMAs, fenced: left alone hipcc reads each tile's two
                // fragments right in front of its three MFMAs and waits lgkmcnt(0) for them -- fifteen exposed LDS round trips per query tile
                uint4 kq[2][2];
                auto kread = [&](int t, int set) {
                    const char* kr = Ksq + (t * 16 + lr) * KROW;
                    kq[set][0] = *(const uint4*)(kr + o0);
                    kq[set][1] = *(const uint4*)(kr + o1);
                };
                // (the 8-wave builds capped at 128 registers have no room for the second fragment set at 16 key tiles or with key masks: they
                // read in place as before)
                constexpr bool PRE = !(MAXT <= 16 && NWV == 8 && (MASK || MAXT == 16));
                if constexpr (PRE) {
                    kread(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if constexpr (PRE) {
                        if (t + 1 < MAXT) kread(t + 1, (t + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        kread(t, t & 1);
                    }
                    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const uint4 k0 = kq[t & 1][0], k1 = kq[t & 1][1];
                    const U kH = {k0.x, k0.y, k0.z, k0.w}, kL = {k1.x, k1.y, k1.z, k1.w};
                    Mma<T>::three(kH, kL, qH, qL, acc[t]);
                    if constexpr (PRE) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const char* kr = Ksq + (t * 16 + lr) * KROW;
                    const uint4 kf0 = *(const uint4*)(kr + o0), kf1 = *(const uint4*)(kr + o1);
                    Mma<T>::half(kf0, qf[0], acc[t]);
                    Mma<T>::half(kf1, qf[1], acc[t]);
                }
            }
        } else {
            uint4 qf = make_uint4(0, 0, 0, 0);
            if (qok) qf = *(const uint4*)(base + (size_t)q * ld + 8 * g);
            const int o0 = (g ^ swz64(lr)) << 4;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[t] = mfma16<T>(*(const uint4*)(Ksq + (t * 16 + lr) * KROW + o0), qf, acc[t]);
            }
        }
        // ---- softmax over keys (this lane: keys 16t + 4g + r of query lr)
        float mx = -INFINITY;
        if constexpr (MASK) {
            int nlim = Nk - 4 * g;
            asm volatile("" : "+v"(nlim));  // opaque per q-tile: stops LICM from hoisting 4*MAXT compare masks into (spilled) SGPRs
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = (t * 16 + r >= nlim) ? -INFINITY : acc[t][r];
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // f16x2: P = exp2(.. + 12), i.e. scaled by 2^12 (at most 4096) so that the lo halves of small probabilities stay normal fp16
        // numbers; the row sum carries the same factor and cancels it in 1 / sum
        const float nm = -mx * c2 + (IsSplit<T>::value ? 12.0f : 0.0f);
        float sum = 0.f;
        // (the 8-wave build at 16 masked key tiles sits at its 128-register cap: the packed form, which wants its operand pairs in adjacent
        // registers, spills there -- it keeps the scalar form)
        if constexpr (IsSplit<T>::value && !(MASK && MAXT == 16 && NWV == 8)) {
            // the kernel is bound by vector issue (one v_exp_f32 and the hi / lo split per score against 1.5 MFMAs): the scale-and-shift and the
            // row sum run as packed fp32 instructions (v_pk_fma_f32, v_pk_add_f32: two scores each)
            typedef float f2v __attribute__((ext_vector_type(2)));
            const f2v c2v = {c2, c2}, nmv = {nm, nm};
            f2v sum2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f2v e = __builtin_elementwise_fma((f2v){acc[t][2 * h], acc[t][2 * h + 1]}, c2v, nmv);
                    e[0] = __builtin_amdgcn_exp2f(e[0]);
                    e[1] = __builtin_amdgcn_exp2f(e[1]);
                    sum2 += e;
                    acc[t][2 * h] = e[0];
                    acc[t][2 * h + 1] = e[1];
                }
            }
            sum = sum2[0] + sum2[1];
        } else {
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // exp2((s - max) * scale*log2e): one fma + raw v_exp_f32 (argument <= 0: no overflow; underflow flushes to 0)
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[t][r], c2, nm));
                    acc[t][r] = p;
                    if constexpr (sizeof(T) == 4) sum += p;
                }
            }
        }
        // ---- O^T = V^T P^T
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        f32x4 osum = (f32x4){0.f, 0.f, 0.f, 0.f};  // 16-bit modes: row sums by MFMA against a ones fragment
        if constexpr (IsSplit<T>::value) {
            // V^T fragments of step s + 1 are requested before step s's split + MFMAs (fenced, like the K fragments above)
            constexpr int NS = (MAXT + 1) / 2;
            uint4 vq[2][2][2];   // [set][dt][hi, lo]
            auto vread = [&](int s, int set) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vr = Vtq + (dt * 16 + lr) * VS + 128 * s + 32 * g;   // the step's keys of head-dim row 16 dt + lr, staged as (hi, lo) quartets
                    vq[set][dt][0] = *(const uint4*)vr;
                    vq[set][dt][1] = *(const uint4*)(vr + 16);
                }
            };
            constexpr bool PRE = !(MAXT <= 16 && NWV == 8 && (MASK || MAXT == 16));   // (as for the K fragments)
            if constexpr (PRE) {
                vread(0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {       // 32 keys per step: the P chunks of tiles 2s and 2s + 1 form one (hi, lo) quartet pair
                if constexpr (PRE) {
                    if (s + 1 < NS) vread(s + 1, (s + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    vread(s, s & 1);
                }
                typedef typename Mma<T>::u4v U;
                // the (hi, lo) quartets of P straight from the accumulators, two scores per conversion (what Chunk<T>::pack + regroup produce,
                // bit for bit: hi = RNE(p), lo = RNE(p - hi)): keys 32 s + 4 g .. + 3 of query lr, then 32 s + 16 + 4 g .. + 3 (nothing past the last tile)
                unsigned h0, h1, h2 = 0u, h3 = 0u, l0, l1, l2 = 0u, l3 = 0u;
                split_pair(acc[2 * s][0], acc[2 * s][1], h0, l0);
                split_pair(acc[2 * s][2], acc[2 * s][3], h1, l1);
                if (2 * s + 1 < MAXT) {
                    const int t1 = 2 * s + 1 < MAXT ? 2 * s + 1 : 0;
                    split_pair(acc[t1][0], acc[t1][1], h2, l2);
                    split_pair(acc[t1][2], acc[t1][3], h3, l3);
                }
                const U pH = {h0, h1, h2, h3}, pL = {l0, l1, l2, l3};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const uint4 v0 = vq[s & 1][dt][0], v1 = vq[s & 1][dt][1];
                    const U vH = {v0.x, v0.y, v0.z, v0.w}, vL = {v1.x, v1.y, v1.z, v1.w};
                    Mma<T>::three(vH, vL, pH, pL, o[dt]);
                }
                if constexpr (PRE) __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const float4 vf = *(const float4*)(Vtq + (dt * 16 + lr) * VS + (t * 16 + 4 * g) * 4);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, acc[t][0], o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, acc[t][1], o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, acc[t][2], o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, acc[t][3], o[dt], 0, 0, 0);
                    }
                }
        } else {
            const T one = from_f32<T>(1.0f);
            union { T e[8]; uint4 u; } ones;
#pragma unroll
            for (int r = 0; r < 8; ++r) ones.e[r] = one;
#pragma unroll
            for (int s = 0; s < (MAXT + 1) / 2; ++s) {
                    // B operand k-slot (g, j): key 32s + 16*(j>>2) + 4g + (j&3)  <-> accumulators of tiles 2s, 2s+1
                    union { T e[8]; uint4 u; } pf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pf.e[r] = from_f32<T>(acc[2 * s][r]);
                        pf.e[4 + r] = (2 * s + 1 < MAXT) ? from_f32<T>(acc[2 * s + 1 < MAXT ? 2 * s + 1 : 0][r]) : from_f32<T>(0.f);
                    }
                    osum = mfma16<T>(ones.u, pf.u, osum);  // every row of osum = sum_k P[k][q] (of the ROUNDED P that PV uses)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const char* vr = Vtq + (dt * 16 + lr) * VS + (32 * s + 4 * g) * 2;
                        const uint2 lo = *(const uint2*)vr, hi = *(const uint2*)(vr + 32);
                        o[dt] = mfma16<T>(make_uint4(lo.x, lo.y, hi.x, hi.y), pf.u, o[dt]);
                    }
                }
        }
        if constexpr (sizeof(T) == 4) {
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
        } else {
            sum = osum[0];
        }
        if (po) {      // partial mode (4-byte types): un-normalised rows + (max, sum) of this key chunk
            if (qok) {
                float* prow = po + ((size_t)b * N + q) * D + h * 32 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) *(float4*)(prow + dt * 16) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
                if (g == 0) pml[((size_t)b * heads + h) * N + q] = make_float2(mx, sum);
            }
            continue;
        }
        const float inv = 1.f / sum;
        if (qok) {
            T* orow = out + ((size_t)b * N + q) * D + h * 32 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if constexpr (sizeof(T) == 4) {
                    const float ov[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
                    store4<T>(orow + dt * 16, ov);
                } else {
                    union { T e[4]; uint2 u; } pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk.e[r] = from_f32<T>(o[dt][r] * inv);
                    *(uint2*)(orow + dt * 16) = pk.u;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// 16-bit types: streaming two-pass kernel.  With head_dim = 32 one exp covers only 64 multiply-adds, so the kernel is bound by the
// VALU (v_exp_f32 is quarter rate), not by MFMA issue: what matters is that the matrix pipe, the VALU and the LDS all stay busy at
// once.  Holding a query tile's whole score row in registers (the kernel above: 120 VGPRs at 480 keys) caps the occupancy at two
// waves per SIMD that run QK -> softmax -> PV as three serial phases.  Here a wave owns 32 queries (two 16-query MFMA tiles that
// share every K / V fragment read) and streams the keys twice in blocks of 32:
//   pass 1   S^T = K Q^T per 16-key tile -> running element-wise max (the scores are not kept)
//   pass 2   S^T again -> p = exp2((s - max) * scale) -> 16-bit P straight from the accumulators into the PV MFMA; row sums by an
//            MFMA against a ones fragment (sums of the ROUNDED P, as before)
// ~90 VGPRs -> 8-wave workgroups, two per CU = 4 waves per SIMD; pass 1's extra QK MFMAs ride on the otherwise idle matrix pipe.
// K is staged row-major [key][64 B] (XOR-swizzled, ds_read_b128 per fragment).  V is staged row-major too and read with the
// hardware transpose (ds_read_b64_tr_b16): the A operand of O^T = V^T P^T needs, per lane, 4 consecutive keys of one head-dim column;
// rows whose (key >> 2) is odd are stored with the two 8-byte halves of every 16-byte chunk swapped, which makes the 8 rows a
// half-wave reads land on disjoint banks.  The 16 head-dim columns of MFMA block dt are 8 (i >> 2) + 4 dt + (i & 3), so a lane ends up
// with 8 consecutive output channels (one 16-byte store, 64 contiguous bytes per query).  Workgroups are numbered so that the heads
// of one sequence run back to back on one XCD: the 64-byte q/k/v slices of neighbouring heads share 128-byte lines through its L2.
typedef short short4v __attribute__((ext_vector_type(4)));

template <typename T, int NW, bool MASK>
__global__ __launch_bounds__(NW * 64) void attention16_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int h = L % heads, b = L / heads;
    const int D = heads * 32, ld = 3 * D;
    const int NB = (N + 31) >> 5;           // 32-key blocks (rows past N are zero-filled)
    char* const Ks = smem;
    char* const Vs = smem + (size_t)NB * 32 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* base = qkv + (size_t)b * N * ld + h * 32;

    // ---- stage K and V rows (64 B each per key)
    for (int idx = tid; idx < NB * 32 * 8; idx += NW * 64) {
        const int key = idx >> 3, c = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < N) v = *(const uint4*)(base + (size_t)key * ld + (c < 4 ? D + c * 8 : 2 * D + (c - 4) * 8));
        if (c < 4) {
            *(uint4*)(Ks + key * 64 + ((c ^ swz64(key)) << 4)) = v;
        } else {
            if ((key >> 2) & 1) v = make_uint4(v.z, v.w, v.x, v.y);
            *(uint4*)(Vs + key * 64 + ((c - 4) << 4)) = v;
        }
    }
    __syncthreads();

    const int lr = lane & 15, g = lane >> 4;
    const float c2 = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 * log2(e)
    const char* const kbase = Ks + lr * 64 + ((g ^ swz64(lr)) << 4);          // + 16-key tile * 1024
    // transposed V read: lane 16 g + 4 q + p supplies row (key) 4 g + q of the block, head-dim columns 8 p + 4 dt .. + 4
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const unsigned vaddr0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vs + (4 * g + tq) * 64 + 16 * tp;
    const unsigned vsel = (g & 1) * 8;      // half swap of rows with odd (key >> 2)
    const T one = from_f32<T>(1.0f);
    union { T e[8]; uint4 u; } ones;
#pragma unroll
    for (int r = 0; r < 8; ++r) ones.e[r] = one;

    const int NT = (N + 31) >> 5;           // 32-query tasks
    for (int task = wave; task < NT; task += NW) {
        uint4 qf[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = task * 32 + 16 * qt + lr;
            qf[qt] = make_uint4(0, 0, 0, 0);
            if (q < N) qf[qt] = *(const uint4*)(base + (size_t)q * ld + 8 * g);
        }
        // ---- pass 1: row maxima
        f32x4 m4[2] = {(f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}, (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}};
        auto tile_max = [&](int t, bool mask) {
            const uint4 kf = *(const uint4*)(kbase + t * 1024);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 sc = mfma16<T>(kf, qf[qt], (f32x4){0.f, 0.f, 0.f, 0.f});
                if (mask) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[r] = (16 * t + 4 * g + r >= N) ? -INFINITY : sc[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) m4[qt][r] = fmaxf(m4[qt][r], sc[r]);
            }
        };
        const int full16 = MASK ? (N >> 4) : 2 * NB;   // 16-key tiles without padding
        for (int t = 0; t < full16; ++t) tile_max(t, false);
        if constexpr (MASK) {
            for (int t = full16; t < 2 * NB; ++t) tile_max(t, true);
        }
        float nm[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = fmaxf(fmaxf(m4[qt][0], m4[qt][1]), fmaxf(m4[qt][2], m4[qt][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            nm[qt] = -mx * c2;
        }
        // ---- pass 2: P and O^T = V^T P^T
        f32x4 o[2][2], osum[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            o[qt][0] = o[qt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            osum[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        auto block = [&](int s, bool mask) {
            const uint4 kf0 = *(const uint4*)(kbase + s * 2048), kf1 = *(const uint4*)(kbase + s * 2048 + 1024);
            uint4 vf[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned a = vaddr0 + s * 2048 + ((8 * dt) ^ vsel);
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(size_t)a);
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(size_t)(a + 1024));
                union { short4v v[2]; uint4 u; } pk;
                pk.v[0] = lo; pk.v[1] = hi;
                vf[dt] = pk.u;
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 s0 = mfma16<T>(kf0, qf[qt], (f32x4){0.f, 0.f, 0.f, 0.f});
                f32x4 s1 = mfma16<T>(kf1, qf[qt], (f32x4){0.f, 0.f, 0.f, 0.f});
                if (mask) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s0[r] = (32 * s + 4 * g + r >= N) ? -INFINITY : s0[r];
                        s1[r] = (32 * s + 16 + 4 * g + r >= N) ? -INFINITY : s1[r];
                    }
                }
                union { T e[8]; uint4 u; } pf;   // k-slot (g, j)  <->  key 32 s + 16 (j >> 2) + 4 g + (j & 3)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf.e[r] = from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c2, nm[qt])));
                    pf.e[4 + r] = from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c2, nm[qt])));
                }
                osum[qt] = mfma16<T>(ones.u, pf.u, osum[qt]);
                o[qt][0] = mfma16<T>(vf[0], pf.u, o[qt][0]);
                o[qt][1] = mfma16<T>(vf[1], pf.u, o[qt][1]);
            }
        };
        const int full32 = MASK ? (N >> 5) : NB;
        for (int s = 0; s < full32; ++s) block(s, false);
        if constexpr (MASK) {
            for (int s = full32; s < NB; ++s) block(s, true);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = task * 32 + 16 * qt + lr;
            if (q < N) {
                const float inv = 1.f / osum[qt][0];
                union { T e[8]; uint4 u; } pk;   // head-dim 8 g + 4 dt + r
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pk.e[r] = from_f32<T>(o[qt][0][r] * inv);
                    pk.e[4 + r] = from_f32<T>(o[qt][1][r] * inv);
                }
                *(uint4*)(out + ((size_t)b * N + q) * D + h * 32 + 8 * g) = pk.u;
            }
        }
    }
}

template <typename T, int NW>
static int launch_attn16(const void* qkv, void* out, int B, int N, int heads, hipStream_t s) {
    const int NB = (N + 31) / 32;
    const int smem = NB * 32 * 128;
    if (N % 32 == 0) {
        auto kern = attention16_kernel<T, NW, false>;
        OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
        hipLaunchKernelGGL(kern, dim3(heads * B), dim3(NW * 64), smem, s, (const T*)qkv, (T*)out, N, heads);
    } else {
        auto kern = attention16_kernel<T, NW, true>;
        OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
        hipLaunchKernelGGL(kern, dim3(heads * B), dim3(NW * 64), smem, s, (const T*)qkv, (T*)out, N, heads);
    }
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}
template <typename T>
static int attn16_dt(const void* qkv, void* out, int B, int N, int heads, hipStream_t s) {
    // only dispatched above 256 keys (k_attention), i.e. at least nine 32-query tasks per (sequence, head): always the 8-wave build
    return launch_attn16<T, 8>(qkv, out, B, N, heads, s);
}

template <typename T, int MAXT, bool MASK>
static int launch_attn_m(const void* qkv, void* out, int B, int N, int heads, hipStream_t s, int kbase, int Nk, float* po, float2* pml) {
    constexpr int NP = ((MAXT + 1) / 2) * 32;
    const int smem = NP * AttnCfg<T>::KROW + 32 * AttnCfg<T>::vstride(NP);
    if constexpr (sizeof(T) == 4 && MAXT >= 30) {
        static const bool w8 = !(getenv("OCRVI_ATTN_F32_W8") && atoi(getenv("OCRVI_ATTN_F32_W8")) == 0);   // A/B switch
        if (w8) {
            auto kern8 = attention_kernel<T, MAXT, MASK, 8>;
            OCRVI_TRY(ensure_max_smem((const void*)kern8, smem));
            hipLaunchKernelGGL(kern8, dim3(heads * B), dim3(512), smem, s, (const T*)qkv, (T*)out, N, heads, kbase, Nk, po, pml);
            OCRVI_HIP(hipGetLastError());
            return OCRVI_OK;
        }
    }
    if constexpr (IsSplit<T>::value && MAXT >= 15 && MAXT <= 16) {
        static const bool w8 = !(getenv("OCRVI_ATTN_X2_W8") && atoi(getenv("OCRVI_ATTN_X2_W8")) == 0);   // A/B switch
        if (w8) {
            auto kern8 = attention_kernel<T, MAXT, MASK, 8>;
            OCRVI_TRY(ensure_max_smem((const void*)kern8, smem));
            hipLaunchKernelGGL(kern8, dim3(heads * B), dim3(512), smem, s, (const T*)qkv, (T*)out, N, heads, kbase, Nk, po, pml);
            OCRVI_HIP(hipGetLastError());
            return OCRVI_OK;
        }
    }
    auto kern = attention_kernel<T, MAXT, MASK, 4>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(heads * B), dim3(256), smem, s, (const T*)qkv, (T*)out, N, heads, kbase, Nk, po, pml);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T, int MAXT>
static int launch_attn(const void* qkv, void* out, int B, int N, int heads, hipStream_t s, int kbase, int Nk, float* po, float2* pml) {
    if (Nk == MAXT * 16) return launch_attn_m<T, MAXT, false>(qkv, out, B, N, heads, s, kbase, Nk, po, pml);
    return launch_attn_m<T, MAXT, true>(qkv, out, B, N, heads, s, kbase, Nk, po, pml);
}

// keys [kbase, kbase + Nk) of sequences of N tokens (Nk <= 512); po / pml non-null = partial mode
template <typename T>
static int attn_range(const void* qkv, void* out, int B, int N, int heads, hipStream_t s, int kbase, int Nk, float* po, float2* pml) {
    const int NT = (Nk + 15) >> 4;
    if (NT <= 4) return launch_attn<T, 4>(qkv, out, B, N, heads, s, kbase, Nk, po, pml);   // 32x256 input: FRM rows, W/4 = 64
    if (NT <= 5) return launch_attn<T, 5>(qkv, out, B, N, heads, s, kbase, Nk, po, pml);   // FRM rows, W/4 = 80
    if (NT <= 8) return launch_attn<T, 8>(qkv, out, B, N, heads, s, kbase, Nk, po, pml);   // 32x256 input: stage 2, 128 tokens
    if (NT <= 15) return launch_attn<T, 15>(qkv, out, B, N, heads, s, kbase, Nk, po, pml); // stage 2, 240 tokens
    if (NT <= 16) return launch_attn<T, 16>(qkv, out, B, N, heads, s, kbase, Nk, po, pml); // 32x256 input: stage 1, 256 tokens
    if (NT <= 30) return launch_attn<T, 30>(qkv, out, B, N, heads, s, kbase, Nk, po, pml); // stage 1, 480 tokens
    return launch_attn<T, 32>(qkv, out, B, N, heads, s, kbase, Nk, po, pml);
}

// Merge of the key chunks of one long sequence: out = sum_c w_c o_c / sum_c w_c l_c with w_c = exp2((m_c - max_c m_c) scale log2e) -- the
// chunks' un-normalised rows all carry exp(-their own max) (and, in f16x2, the same 2^12), which w_c brings onto the common maximum.
template <typename T>
__global__ void attention_combine_kernel(const float* __restrict__ po, const float2* __restrict__ pml, T* __restrict__ out, int B, int N, int heads, int nchunk) {
    const int D = heads * 32;
    const size_t total = (size_t)B * N * heads * 8;          // one thread per (row, head, 4 channels)
    const size_t rowsD = (size_t)B * N * D, rowsH = (size_t)B * heads * N;
    const float c2 = 0.17677669529663687f * 1.4426950408889634f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i & 7);
        const size_t t = i >> 3;
        const int h = (int)(t % heads);
        const size_t bq = t / heads;                          // b * N + q
        const size_t b = bq / N, q = bq % N;
        float m = -INFINITY;
        for (int c = 0; c < nchunk; ++c) m = fmaxf(m, pml[c * rowsH + (b * heads + h) * N + q].x);
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float2 ml = pml[c * rowsH + (b * heads + h) * N + q];
            const float w = __builtin_amdgcn_exp2f((ml.x - m) * c2);
            const float4 o = *(const float4*)(po + c * rowsD + bq * D + h * 32 + c4 * 4);
            acc[0] += w * o.x; acc[1] += w * o.y; acc[2] += w * o.z; acc[3] += w * o.w;
            den += w * ml.y;
        }
        const float inv = 1.f / den;
        const float ov[4] = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
        store4<T>(out + bq * D + h * 32 + c4 * 4, ov);
    }
}

static inline int attn_chunks(int N) { return (N + 511) / 512; }
static inline int attn_chunk_keys(int N) { return (int)align_up((size_t)cdiv(N, attn_chunks(N)), 32); }

template <typename T>
static int attn_dt(const void* qkv, void* out, int B, int N, int heads, hipStream_t s, void* scratch) {
    if (N <= 512) return attn_range<T>(qkv, out, B, N, heads, s, 0, N, nullptr, nullptr);
    if constexpr (sizeof(T) == 4) {
        OCRVI_CHECK(scratch, OCRVI_EINVAL, "attention: %d keys need the chunk scratch (attention_scratch_bytes)", N);
        const int nc = attn_chunks(N), ck = attn_chunk_keys(N);
        const size_t rowsD = (size_t)B * N * heads * 32, rowsH = (size_t)B * heads * N;
        float* po = (float*)scratch;
        float2* pml = (float2*)(po + (size_t)nc * rowsD);
        for (int c = 0; c < nc; ++c) {
            const int k0 = c * ck, nk = std::min(ck, N - k0);
            OCRVI_TRY(attn_range<T>(qkv, out, B, N, heads, s, k0, nk, po + (size_t)c * rowsD, pml + (size_t)c * rowsH));
        }
        const size_t total = rowsH * 8;
        hipLaunchKernelGGL(attention_combine_kernel<T>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, s, po, pml, (T*)out, B, N, heads, nc);
        OCRVI_HIP(hipGetLastError());
        return OCRVI_OK;
    } else {
        set_error("attention: the register-resident 16-bit kernel takes at most 512 keys");
        return OCRVI_EINVAL;
    }
}

size_t attention_scratch_bytes(int dtype, int B, int N, int heads) {
    if (dtype_size(dtype) != 4 || N <= 512) return 0;
    return (size_t)attn_chunks(N) * ((size_t)B * N * heads * 32 * 4 + (size_t)B * heads * N * 8);
}

int k_attention(int dtype, const void* qkv, void* out, int B, int N, int heads, hipStream_t s, void* scratch) {
    OCRVI_CHECK(qkv && out && B > 0 && B < 65536 && heads > 0 && N > 0, OCRVI_EINVAL, "attention: bad shape B=%d N=%d heads=%d", B, N, heads);
    // 16-bit types: the streaming kernel stages every key (128 B each) in one workgroup's LDS; 4-byte types: key chunks of <= 512 + a merge
    OCRVI_CHECK(N <= (dtype_size(dtype) == 2 ? 1024 : 4096), OCRVI_EINVAL, "attention: sequence length %d unsupported (at most %d tokens in this mode)", N,
                dtype_size(dtype) == 2 ? 1024 : 4096);
    char tag[64];
    snprintf(tag, sizeof(tag), "attention_hd32_%s", dtype_name(dtype));
    const double esz = (double)dtype_size(dtype);
    ProfScope ps(tag, 4.0 * B * heads * (double)N * N * 32, (double)B * N * heads * 32 * 4 * esz, s);
    // 16-bit types: the streaming two-pass kernel wins from ~256 keys up (measured on MI355X: 480 keys 127 us vs 145 us for 256 x 8 heads;
    // 240 keys 75 vs 65 us), the register-resident kernel below that.  OCRVI_ATTN_STREAM=0 forces the latter (A/B switch).
    static const bool streaming = !(getenv("OCRVI_ATTN_STREAM") && atoi(getenv("OCRVI_ATTN_STREAM")) == 0);
    OCRVI_CHECK((size_t)B * heads < ((size_t)1 << 31), OCRVI_EINVAL, "attention: too many (sequence, head) pairs");
    switch (dtype) {
        case OCRVI_F32: return attn_dt<float>(qkv, out, B, N, heads, s, scratch);
        case OCRVI_BF16: return (streaming && N > 256) || N > 512 ? attn16_dt<bf16_t>(qkv, out, B, N, heads, s) : attn_dt<bf16_t>(qkv, out, B, N, heads, s, scratch);
        case OCRVI_F16: return (streaming && N > 256) || N > 512 ? attn16_dt<f16_t>(qkv, out, B, N, heads, s) : attn_dt<f16_t>(qkv, out, B, N, heads, s, scratch);
        case OCRVI_F16X2: return attn_dt<f16x2_t>(qkv, out, B, N, heads, s, scratch);
    }
    set_error("unknown dtype %d", dtype);
    return OCRVI_EINVAL;
}

}  // namespace ocrvi
