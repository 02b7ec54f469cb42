// Fused multi-head self-attention for head_dim = 32 (always the case in this model: heads = dim/32,
// svtrv2.py:70-72,169-171).  One workgroup = one (sequence, head): K [N][32] and V^T [32][N] of the head are
// staged once into LDS (N <= 512: 480 tokens in stage 1, 240 in stage 2, 80 per FRM row at 48x320 input).
// Each wave then takes 16-query tiles:
//   S^T = K Q^T  (MFMA A = K rows, B = Q)  -> accumulator column = query, rows = keys: a query's whole score row
//                                             sits in ONE lane column (4 lanes x 4*NT registers) -> 2 shuffles per reduce
//   softmax in registers (exp2 with the 1/sqrt(32) scale folded in), un-normalised P kept in the accumulators
//   O^T = V^T P^T (MFMA A = V^T rows = head-dim, B = P straight from the accumulators; the K-order of the product is
//                  permuted identically on both operands, cdna_hip_programming.md section 3)
#include "kernels.h"

namespace ocrvi {

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

__device__ __forceinline__ int swz64(int row) { return (-(row >> 2)) & 3; }

template <typename T> __device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<bf16_t>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<f16_t>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// MAXT = number of 16-key tiles held in registers; MASK = false when N == 16*MAXT exactly (480 / 240 / 80 tokens at 48x320 input):
// then no key masking is generated at all.
template <typename T, int MAXT, bool MASK>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int N, int heads) {
    constexpr int EPC = TypeInfo<T>::EPC;
    constexpr int KROW = AttnCfg<T>::KROW;
    constexpr int CH = KROW / 16;  // 16-byte chunks per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int D = heads * 32, ld = 3 * D;
    constexpr int NP = ((MAXT + 1) / 2) * 32;  // key rows staged in LDS: whole 32-key PV steps (zero-filled past N)
    constexpr int VS = AttnCfg<T>::vstride(NP);
    char* Ks = smem;
    char* Vt = smem + (size_t)NP * KROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* base = qkv + (size_t)b * N * ld + h * 32;

    // ---- stage K (row-major, swizzled) and V^T
    for (int idx = tid; idx < NP * CH; idx += 256) {
        const int key = idx / CH, ch = idx % CH;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < N) {
            kv = *(const uint4*)(base + (size_t)key * ld + D + ch * EPC);
            vv = *(const uint4*)(base + (size_t)key * ld + 2 * D + ch * EPC);
        }
        const int sw = sizeof(T) == 4 ? swz128(key) : swz64(key);
        *(uint4*)(Ks + key * KROW + ((ch ^ sw) << 4)) = kv;
        union { uint4 u; T e[EPC]; } r;
        r.u = vv;
#pragma unroll
        for (int e = 0; e < EPC; ++e) *(T*)(Vt + (ch * EPC + e) * VS + key * (int)sizeof(T)) = r.e[e];
    }
    __syncthreads();

    const int lr = lane & 15, g = lane >> 4;
    const float c2 = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 * log2(e)
    const int NQ = (N + 15) >> 4;
    for (int qt = wave; qt < NQ; qt += 4) {
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
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const char* kr = Ksq + (t * 16 + lr) * KROW;
                const uint4 kf[2] = {*(const uint4*)(kr + o0), *(const uint4*)(kr + o1)};
                Mma<float>::run(kf, qf, acc[t]);
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
            int nlim = N - 4 * g;
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
        const float nm = -mx * c2;
        float sum = 0.f;
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
        // ---- O^T = V^T P^T
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        f32x4 osum = (f32x4){0.f, 0.f, 0.f, 0.f};  // 16-bit modes: row sums by MFMA against a ones fragment
        if constexpr (sizeof(T) == 4) {
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
        const float inv = 1.f / sum;
        if (qok) {
            T* orow = out + ((size_t)b * N + q) * D + h * 32 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if constexpr (sizeof(T) == 4) {
                    *(float4*)(orow + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
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

template <typename T, int MAXT, bool MASK>
static int launch_attn_m(const void* qkv, void* out, int B, int N, int heads, hipStream_t s) {
    constexpr int NP = ((MAXT + 1) / 2) * 32;
    const int smem = NP * AttnCfg<T>::KROW + 32 * AttnCfg<T>::vstride(NP);
    auto kern = attention_kernel<T, MAXT, MASK>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(heads, B), dim3(256), smem, s, (const T*)qkv, (T*)out, N, heads);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T, int MAXT>
static int launch_attn(const void* qkv, void* out, int B, int N, int heads, hipStream_t s) {
    if (N == MAXT * 16) return launch_attn_m<T, MAXT, false>(qkv, out, B, N, heads, s);
    return launch_attn_m<T, MAXT, true>(qkv, out, B, N, heads, s);
}

template <typename T>
static int attn_dt(const void* qkv, void* out, int B, int N, int heads, hipStream_t s) {
    const int NT = (N + 15) >> 4;
    if (NT <= 4) return launch_attn<T, 4>(qkv, out, B, N, heads, s);   // 32x256 input: FRM rows, W/4 = 64
    if (NT <= 5) return launch_attn<T, 5>(qkv, out, B, N, heads, s);   // FRM rows, W/4 = 80
    if (NT <= 8) return launch_attn<T, 8>(qkv, out, B, N, heads, s);   // 32x256 input: stage 2, 128 tokens
    if (NT <= 15) return launch_attn<T, 15>(qkv, out, B, N, heads, s); // stage 2, 240 tokens
    if (NT <= 16) return launch_attn<T, 16>(qkv, out, B, N, heads, s); // 32x256 input: stage 1, 256 tokens
    if (NT <= 30) return launch_attn<T, 30>(qkv, out, B, N, heads, s); // stage 1, 480 tokens
    return launch_attn<T, 32>(qkv, out, B, N, heads, s);
}

int k_attention(int dtype, const void* qkv, void* out, int B, int N, int heads, hipStream_t s) {
    OCRVI_CHECK(qkv && out && B > 0 && B < 65536 && heads > 0 && N > 0, OCRVI_EINVAL, "attention: bad shape B=%d N=%d heads=%d", B, N, heads);
    OCRVI_CHECK(N <= 512, OCRVI_EINVAL, "attention: sequence length %d > 512 unsupported (crop wider than ~340 px at height 48)", N);
    char tag[64];
    snprintf(tag, sizeof(tag), "attention_hd32_%s", dtype_name(dtype));
    const double esz = (double)dtype_size(dtype);
    ProfScope ps(tag, 4.0 * B * heads * (double)N * N * 32, (double)B * N * heads * 32 * 4 * esz, s);
    switch (dtype) {
        case OCRVI_F32: return attn_dt<float>(qkv, out, B, N, heads, s);
        case OCRVI_BF16: return attn_dt<bf16_t>(qkv, out, B, N, heads, s);
        case OCRVI_F16: return attn_dt<f16_t>(qkv, out, B, N, heads, s);
    }
    set_error("unknown dtype %d", dtype);
    return OCRVI_EINVAL;
}

}  // namespace ocrvi
