// Numerics probe for an fp32-equivalent GEMM on the 16-bit matrix pipe (results: profiles/r03_split_gemm.md).
//   C[m][n] = sum_k A[m][k] B[n][k]   (A = activations, B = weights), one wave per 16x16 tile, operands converted on the fly
// Modes:
//   0  v_mfma_f32_16x16x4_f32            (what dtype f32 runs today: an fp32 fmaf chain)
//   1  f16 hi/lo split, weights scaled by a power of two: x = hi + lo, lane operand [4 hi | 4 lo] of 4 consecutive k,
//      two v_mfma_f32_16x16x32_f16 per 16 k:  (ah,al).(bh,bl) = hh + ll   and   (ah,al).(bl,bh) = hl + lh
//   2  the same without the weight scale (shows fp16's subnormal floor on the lo halves)
//   3  plain f16 (one MFMA per 32 k)
//   4  bf16 hi/mid/lo, six v_mfma_f32_16x16x32_bf16 per 32 k (hh, hm, mh, hl, lh, mm)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/split_probe tools/split_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline void split16(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}

template <int MODE>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, float bscale) {
    const int lane = threadIdx.x, lr = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const float* a = A + (size_t)(m0 + lr) * K;
    const float* b = B + (size_t)(n0 + lr) * K;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[k + g], a[k + g], acc, 0, 0, 0);
    } else if constexpr (MODE == 1 || MODE == 2) {
        for (int k = 0; k < K; k += 16) {   // lane: k + 4g .. + 3
            f16x8 av, bv, bs;
            for (int j = 0; j < 4; ++j) {
                _Float16 h, l;
                split16(a[k + 4 * g + j], h, l);
                av[j] = h; av[4 + j] = l;
                split16(b[k + 4 * g + j] * bscale, h, l);
                bv[j] = h; bv[4 + j] = l;
                bs[j] = l; bs[4 + j] = h;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bs, av, acc, 0, 0, 0);
        }
        const float inv = 1.0f / bscale;
        for (int r = 0; r < 4; ++r) acc[r] *= inv;
    } else if constexpr (MODE == 3) {
        for (int k = 0; k < K; k += 32) {
            f16x8 av, bv;
            for (int j = 0; j < 8; ++j) { av[j] = (_Float16)a[k + 8 * g + j]; bv[j] = (_Float16)b[k + 8 * g + j]; }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 32) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                float x = a[k + 8 * g + j];
                ah[j] = (__bf16)x; x -= (float)ah[j]; am[j] = (__bf16)x; x -= (float)am[j]; al[j] = (__bf16)x;
                x = b[k + 8 * g + j];
                bh[j] = (__bf16)x; x -= (float)bh[j]; bm[j] = (__bf16)x; x -= (float)bm[j]; bl[j] = (__bf16)x;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, acc, 0, 0, 0);   // small terms first
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc, 0, 0, 0);
        }
    }
    // D: column (lane & 15) = B-operand row (pixel m), rows 4g + r = A-operand row (channel n)
    for (int r = 0; r < 4; ++r) C[(size_t)(m0 + lr) * N + n0 + 4 * g + r] = acc[r];
}

struct Stat { double max_rel = 0, rms_rel = 0; };

static Stat compare(const std::vector<float>& c, const std::vector<double>& ref) {
    double se = 0, sr = 0, mx = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double e = (double)c[i] - ref[i];
        se += e * e; sr += ref[i] * ref[i];
        if (fabs(e) > mx) mx = fabs(e);
    }
    const double rms = sqrt(sr / c.size());
    Stat s; s.max_rel = mx / rms; s.rms_rel = sqrt(se / c.size()) / rms;
    return s;
}

int main() {
    const int M = 256, N = 256;
    const char* names[5] = {"mfma_f32 (fp32 chain)", "f16 hi/lo, scaled W", "f16 hi/lo, raw W", "plain f16", "bf16 x3 (6 MFMA)"};
    const char* dists[4] = {"A~N(0,1)", "A=relu(N(0,1))", "A~0.01*N(0,1)", "A~N(0,1)*lognormal(2)"};
    printf("| K | activations | kernel | max err / rms(C) | rms err / rms(C) |\n|---|---|---|---|---|\n");
    for (int K : {256, 1024, 4608}) {
        for (int dist = 0; dist < 4; ++dist) {
            std::mt19937 rng(1234 + K + dist);
            std::normal_distribution<float> nd(0.f, 1.f);
            std::vector<float> A((size_t)M * K), B((size_t)N * K), Cc((size_t)M * N);
            for (auto& v : A) {
                float x = nd(rng);
                if (dist == 1) x = x > 0 ? x : 0.f;
                if (dist == 2) x *= 0.01f;
                if (dist == 3) x *= expf(2.f * nd(rng));
                v = x;
            }
            float bmax = 0.f;
            for (auto& v : B) { v = nd(rng) / sqrtf((float)K); bmax = fmaxf(bmax, fabsf(v)); }
            int e; frexpf(bmax, &e);                       // bmax in [2^(e-1), 2^e)
            const float bscale = ldexpf(1.0f, 14 - e);     // scaled max in [2^13, 2^14)
            std::vector<double> ref((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
                    ref[(size_t)m * N + n] = s;
                }
            float *dA, *dB, *dC;
            CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, Cc.size() * 4));
            CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            for (int mode = 0; mode < 5; ++mode) {
                dim3 grid(N / 16, M / 16);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(probe<0>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f); break;
                    case 1: hipLaunchKernelGGL(probe<1>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, bscale); break;
                    case 2: hipLaunchKernelGGL(probe<2>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f); break;
                    case 3: hipLaunchKernelGGL(probe<3>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f); break;
                    case 4: hipLaunchKernelGGL(probe<4>, grid, dim3(64), 0, 0, dA, dB, dC, M, N, K, 1.f); break;
                }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(Cc.data(), dC, Cc.size() * 4, hipMemcpyDeviceToHost));
                const Stat s = compare(Cc, ref);
                printf("| %d | %s | %s | %.3e | %.3e |\n", K, dists[dist], names[mode], s.max_rel, s.rms_rel);
            }
            CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
        }
    }
    return 0;
}
