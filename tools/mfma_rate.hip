// Issue-rate probe for v_mfma_f32_16x16x32_f16 on gfx950: register operands only (no LDS, no memory), the accumulation patterns of the
// f16x2 kernels.  Reports the achieved 16-bit MFMA TFLOP/s of the whole chip against 2500 and the implied cycles per MFMA per SIMD.
//   DIST  = number of independent accumulators visited round-robin (a dependent MFMA follows DIST - 1 independent ones)
//   waves = waves per workgroup (one workgroup per CU): 4 = one wave per SIMD, 8 = two
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DIST, bool MOVS>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters) {
    f32x4 acc[DIST];
    for (int i = 0; i < DIST; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.001f + i);
        b[i] = (_Float16)(threadIdx.x * 0.002f - i);
    }
    unsigned junk[8] = {threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < DIST; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (MOVS) {   // six independent VALU moves per row, as the regrouping of a weight-row fragment pair
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(junk[i]) : "v"(junk[i + 1]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < DIST; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.f || junk[0] == 0xdeadbeef) out[0] = s;
}

template <int DIST, bool MOVS>
static void run(const char* name, int waves, float* out) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<DIST, MOVS>), dim3(grid), dim3(waves * 64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<DIST, MOVS>), dim3(grid), dim3(waves * 64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * waves * iters * 4 * 3 * DIST;        // MFMAs
    const double tf = n * 16384.0 / (ms * 1e-3) / 1e12;
    const double cyc = (ms * 1e-3) * 2.4e9 / ((double)iters * 4 * 3 * DIST * waves / 4.0);   // cycles per MFMA per SIMD at 2.4 GHz
    printf("%-34s waves/CU %d: %8.3f ms  %7.1f TFLOP/s (%.0f %% of 2500)  %.1f cycles per MFMA per SIMD @2.4GHz\n", name, waves, ms, tf, tf / 25.0, cyc);
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    for (int waves : {4, 8}) {
        run<1, false>("dist 1 (dependent chain)", waves, out);
        run<2, false>("dist 2", waves, out);
        run<4, false>("dist 4", waves, out);
        run<8, false>("dist 8", waves, out);
        run<4, true>("dist 4 + 6 v_mov per 12 MFMA", waves, out);
        run<8, true>("dist 8 + 6 v_mov per 24 MFMA", waves, out);
    }
    return 0;
}
