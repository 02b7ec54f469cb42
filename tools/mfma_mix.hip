// What keeps an f16x2 GEMM step off the matrix pipe?  The duo / ring step body rebuilt feature by feature on register-resident operands:
//   V0  96 MFMAs per step (8 row blocks x 4 column blocks x 3 products, the duo kernel's order), nothing else
//   V1  + the 6 regrouping v_mov per row block
//   V2  + the step's 24 ds_read_b128 (fragments really come from LDS; swizzled rows as in the kernels)
//   V3  + one s_barrier per step
//   V4  + ~120 scalar bookkeeping instructions per step
//   V5  + 8 LDS-DMA pieces per step (global_load_lds_dwordx4 from an L2-resident buffer, counted vmcnt)
// One workgroup of 8 waves per CU (two waves per SIMD), 3-product f16x2 arithmetic; reports MFMA TFLOP/s of the chip against 2500 and
// cycles per MFMA per SIMD at 2.4 GHz.  hipcc -O3 --offload-arch=gfx950 tools/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz128(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

// V6..V9: V5 + every 12th step is an "epilogue step" of one of the two groups (waves 0-3 / 4-7 in turn): 8 stores of one accumulator
// fragment each instead of the MFMAs, to a 512 MiB output with 4096-byte rows.  V6: 16 rows x 64 B per store (the kernels' fragment
// layout), V7: the same non-temporal, V8: 4 rows x 256 B per store (whole lines), V9: the stores skipped (out of range)
template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe(float* out, const char* gbuf, int steps, char* obuf = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4, wn = wave & 3, grp = wave >> 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // LDS image: B stage 256 rows x 128 B at 0 (two slots), A stages at 64 KiB
    for (int i = tid; i < 160 * 1024 / 16; i += WAVES * 64) {
        union { uint4 u; _Float16 h[8]; } r;
        for (int e = 0; e < 8; ++e) r.h[e] = (_Float16)(((i * 8 + e) * 2654435761u >> 20 & 255) * (1.f / 256.f) - 0.5f);
        ((uint4*)smem)[i] = r.u;
    }
    __syncthreads();
    f32x4 acc[4][8];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int sw = swz128(lr);
    const int fo0 = ((2 * g) ^ sw) << 4, fo1 = ((2 * g + 1) ^ sw) << 4;
    u4v wH[4], wL[4], xH[2], xL[2];
    uint4 xc[2][2];
    {   // initial operands
        const char* Br = smem + (wn * 64 + lr) * 128;
        for (int a = 0; a < 4; ++a) {
            const uint4 c0 = *(const uint4*)(Br + a * 2048 + fo0), c1 = *(const uint4*)(Br + a * 2048 + fo1);
            wH[a] = (u4v){c0.x, c0.y, c1.x, c1.y};
            wL[a] = (u4v){c0.z, c0.w, c1.z, c1.w};
        }
        const char* Ar = smem + 65536 + grp * 49152 + lr * 128;
        xc[0][0] = *(const uint4*)(Ar + fo0); xc[0][1] = *(const uint4*)(Ar + fo1);
        xc[1][0] = *(const uint4*)(Ar + 2048 + fo0); xc[1][1] = *(const uint4*)(Ar + 2048 + fo1);
        xH[0] = (u4v){xc[0][0].x, xc[0][0].y, xc[0][1].x, xc[0][1].y};
        xL[0] = (u4v){xc[0][0].z, xc[0][0].w, xc[0][1].z, xc[0][1].w};
        xH[1] = xH[0]; xL[1] = xL[0];
    }
    auto mm = [&](const u4v& w, const u4v& x, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    };
    const int prow = lane >> 3, chunk = (lane & 7) ^ swz128(wave * 8 + prow);
    const unsigned voff = (unsigned)((wave * 8 + prow) * 1536 + chunk * 16);
    int book[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
    const long long tstart = clock64();
    for (int s = 0; s < steps; ++s) {
        if constexpr (V >= 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (V >= 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (V >= 3) asm volatile("s_barrier" ::: "memory");
        if constexpr (V >= 4) {   // scalar bookkeeping: dependent uniform arithmetic the compiler cannot fold (values come from memory-like opaque asm)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                int v = book[i];
                asm volatile("s_add_i32 %0, %0, 3\n\ts_mul_i32 %0, %0, 5\n\ts_and_b32 %0, %0, 0xffff\n\ts_add_i32 %0, %0, 7\n\ts_lshr_b32 %0, %0, 1\n\t"
                             "s_add_i32 %0, %0, 3\n\ts_xor_b32 %0, %0, 0x55\n\ts_add_i32 %0, %0, 1\n\ts_and_b32 %0, %0, 0xfff\n\ts_add_i32 %0, %0, 9" : "+s"(v));
                book[i] = v;
            }
        }
        // V6..V9: in 4 of every 12 steps one group (in turn) runs an "epilogue step": every row block's 12 MFMAs are replaced by one store of
        // an accumulator fragment (8 stores per wave and step); reads, regrouping and DMA duty unchanged
        const bool is_epi = V >= 6 && V != 12 && (s % 12) >= 8 && ((s / 12) & 1) == grp;
        const unsigned tile = (unsigned)((blockIdx.x * 64 + (s / 12) % 64) * 4 + (s % 12 - 8));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obuf, 0, V == 9 ? 16 : 0x20000000, 0x00020000);
        const char* Ar = smem + 65536 + grp * 49152 + (s % 3) * 16384 + lr * 128;
        const char* Br = smem + (s & 1) * 32768 + (wn * 64 + lr) * 128;
        if constexpr (V >= 2) {
            uint4 wc[4][2];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                wc[a][0] = *(const uint4*)(Br + a * 2048 + fo0);
                wc[a][1] = *(const uint4*)(Br + a * 2048 + fo1);
            }
            xc[0][0] = *(const uint4*)(Ar + fo0); xc[0][1] = *(const uint4*)(Ar + fo1);
            xc[1][0] = *(const uint4*)(Ar + 2048 + fo0); xc[1][1] = *(const uint4*)(Ar + 2048 + fo1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                wH[a] = (u4v){wc[a][0].x, wc[a][0].y, wc[a][1].x, wc[a][1].y};
                wL[a] = (u4v){wc[a][0].z, wc[a][0].w, wc[a][1].z, wc[a][1].w};
            }
            xH[0] = (u4v){xc[0][0].x, xc[0][0].y, xc[0][1].x, xc[0][1].y};
            xL[0] = (u4v){xc[0][0].z, xc[0][0].w, xc[0][1].z, xc[0][1].w};
            if constexpr (V == 11 || V == 12) {
                xH[1] = (u4v){xc[1][0].x, xc[1][0].y, xc[1][1].x, xc[1][1].y};
                xL[1] = (u4v){xc[1][0].z, xc[1][0].w, xc[1][1].z, xc[1][1].w};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (V == 11 || V == 12) {
            // dist-8 order: two row blocks at a time, every accumulator is revisited after 7 other MFMAs.  Operands of both blocks resident
            // (xH/xL[0..1]); the next pair's raw chunks are read during this pair and regrouped at its end.
#pragma unroll
            for (int bp = 0; bp < 8; bp += 2) {
                uint4 nc[2][2];
                if (bp + 2 < 8) {
                    nc[0][0] = *(const uint4*)(Ar + (bp + 2) * 2048 + fo0); nc[0][1] = *(const uint4*)(Ar + (bp + 2) * 2048 + fo1);
                    nc[1][0] = *(const uint4*)(Ar + (bp + 3) * 2048 + fo0); nc[1][1] = *(const uint4*)(Ar + (bp + 3) * 2048 + fo1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!is_epi) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) mm(wL[a], xH[0], acc[a][bp]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) mm(wL[a], xH[1], acc[a][bp + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + ((s + 1) & 1) * 32768 + (((bp >> 1) & 3) * 8 + wave) * 1024);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbuf + (size_t)(s & 63) * 128 + ((bp >> 1) & 3) * 64 * 1536), "s"(dst) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!is_epi) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) mm(wH[a], xL[0], acc[a][bp]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) mm(wH[a], xL[1], acc[a][bp + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + ((s + 1) & 1) * 32768 + (((bp >> 1) & 3) * 8 + wave) * 1024);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbuf + (size_t)(s & 63) * 128 + ((bp >> 1) & 3) * 64 * 1536), "s"(dst) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!is_epi) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) mm(wH[a], xH[0], acc[a][bp]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) mm(wH[a], xH[1], acc[a][bp + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (bp + 2 < 8) {
                    xH[0] = (u4v){nc[0][0].x, nc[0][0].y, nc[0][1].x, nc[0][1].y}; xL[0] = (u4v){nc[0][0].z, nc[0][0].w, nc[0][1].z, nc[0][1].w};
                    xH[1] = (u4v){nc[1][0].x, nc[1][0].y, nc[1][1].x, nc[1][1].y}; xL[1] = (u4v){nc[1][0].z, nc[1][0].w, nc[1][1].z, nc[1][1].w};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int cur = b & 1, nxt = cur ^ 1;
            if (is_epi && V != 10) {
                const int a = b & 3, bb = b >> 2;
                unsigned off;
                if (V == 8) off = ((tile * 32 + bb * 16 + a * 4 + (lane >> 4)) * 4096u + wn * 256u + (lane & 15) * 16u) & 0x1fffffffu;
                else off = ((tile * 32 + bb * 16 + lr) * 4096u + wn * 256u + a * 64u + g * 16u) & 0x1fffffffu;
                u4v pk = (u4v){__float_as_uint(acc[a][bb][0]), __float_as_uint(acc[a][bb][1]), __float_as_uint(acc[a][bb][2]), __float_as_uint(acc[a][bb][3])};
                if (V == 7) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 0);
            }
            if (!is_epi) { mm(wL[0], xH[cur], acc[0][b]); mm(wL[1], xH[cur], acc[1][b]); }
            if (V >= 2 && b + 2 < 8) xc[cur][0] = *(const uint4*)(Ar + (b + 2) * 2048 + fo0);
            __builtin_amdgcn_sched_barrier(0);
            if (!is_epi) { mm(wL[2], xH[cur], acc[2][b]); mm(wL[3], xH[cur], acc[3][b]); }
            if (V >= 2 && b + 2 < 8) xc[cur][1] = *(const uint4*)(Ar + (b + 2) * 2048 + fo1);
            __builtin_amdgcn_sched_barrier(0);
            if (!is_epi) { mm(wH[0], xL[cur], acc[0][b]); mm(wH[1], xL[cur], acc[1][b]); }
            if (V >= 1 && b + 1 < 8) {
                if (V >= 2) xH[nxt] = (u4v){xc[nxt][0].x, xc[nxt][0].y, xc[nxt][1].x, xc[nxt][1].y};
                else asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6" : "=v"(xH[nxt][0]), "=v"(xH[nxt][1]), "=v"(xH[nxt][2]) : "v"(xH[nxt][3]), "v"(xH[cur][1]), "v"(xH[cur][2]), "v"(xH[cur][3]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!is_epi) { mm(wH[2], xL[cur], acc[2][b]); mm(wH[3], xL[cur], acc[3][b]); }
            if (V >= 1 && b + 1 < 8) {
                if (V >= 2) xL[nxt] = (u4v){xc[nxt][0].z, xc[nxt][0].w, xc[nxt][1].z, xc[nxt][1].w};
                else asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6" : "=v"(xL[nxt][0]), "=v"(xL[nxt][1]), "=v"(xL[nxt][2]) : "v"(xL[nxt][3]), "v"(xL[cur][1]), "v"(xL[cur][2]), "v"(xL[cur][3]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!is_epi) { mm(wH[0], xH[cur], acc[0][b]); mm(wH[1], xH[cur], acc[1][b]); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (V >= 5) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + ((s + 1) & 1) * 32768 + ((b & 3) * 8 + wave) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbuf + (size_t)(s & 63) * 128 + (b & 3) * 64 * 1536), "s"(dst) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!is_epi) { mm(wH[2], xH[cur], acc[2][b]); mm(wH[3], xH[cur], acc[3][b]); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x == 17) ((long long*)out)[2] = clock64() - tstart;     // shader cycles of the loop (one wave)
    float sum = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 8; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    int bs = 0;
    for (int i = 0; i < 12; ++i) bs += book[i];
    if (sum == 12345.f || bs == -77) out[0] = sum;
}

template <int V, int WAVES>
static void run(const char* name, float* out, const char* gbuf, char* obuf = nullptr) {
    const int steps = 600, grid = 256;
    auto k = probe<V, WAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 160 * 1024, 0, out, gbuf, 8, obuf);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 160 * 1024, 0, out, gbuf, steps, obuf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double n = (double)grid * WAVES * steps * 96 * (V >= 6 && V != 12 ? 5.0 / 6.0 : 1.0);   // (V >= 6: a wave computes in 20 of 24 steps)
    const double tf = n * 16384.0 / (ms * 1e-3) / 1e12;
    const double cyc = (ms * 1e-3) * 2.4e9 / ((double)steps * 96 * WAVES / 4.0);
    long long hc[3] = {0, 0, 0};
    hipMemcpy(hc, out, 24, hipMemcpyDeviceToHost);
    const double real = (double)hc[2] / ((double)steps * 96 * WAVES / 4.0);
    printf("%-58s waves/CU %d: %8.3f ms  %7.1f TFLOP/s MFMA (%.0f %% of 2500; %.0f TFLOP/s of f16x2 product)  %.1f cyc/MFMA/SIMD @2.4GHz, %.1f shader cycles (clock %.2f GHz)\n",
           name, WAVES, ms, tf, tf / 25.0, tf / 3, cyc, real, 2.4 * real / cyc);
}

int main() {
    float* out;
    char* gbuf;
    hipMalloc(&out, 64);
    hipMalloc(&gbuf, 1536 * 512);
    hipMemset(gbuf, 0x3c, 1536 * 512);
    run<0, 8>("V0 MFMAs only", out, gbuf);
    run<1, 8>("V1 + regroup v_mov", out, gbuf);
    run<2, 8>("V2 + fragments from LDS (24 ds_read_b128 / step)", out, gbuf);
    run<3, 8>("V3 + s_barrier per step", out, gbuf);
    run<4, 8>("V4 + 120 scalar instructions per step", out, gbuf);
    run<5, 8>("V5 + 8 LDS-DMA pieces per step (L2-resident source)", out, gbuf);
    char* obuf;
    hipMalloc(&obuf, 512u << 20);
    run<6, 8>("V6 + epilogue steps: 8 stores of 16 rows x 64 B", out, gbuf, obuf);
    run<7, 8>("V7   the same, non-temporal", out, gbuf, obuf);
    run<8, 8>("V8   4 rows x 256 B per store", out, gbuf, obuf);
    run<9, 8>("V9   stores out of range (nothing written)", out, gbuf, obuf);
    run<10, 8>("V10  epilogue steps without the stores", out, gbuf, obuf);
    run<11, 8>("V11  as V10, MFMAs in dist-8 order (two row blocks at a time)", out, gbuf, obuf);
    run<12, 8>("V12  as V5 (no epilogue steps) in dist-8 order", out, gbuf, obuf);
    run<12, 4>("V12  dist-8 order, one wave per SIMD", out, gbuf, obuf);
    run<0, 4>("V0 MFMAs only", out, gbuf);
    run<2, 4>("V2 fragments from LDS", out, gbuf);
    run<3, 4>("V3 + s_barrier per step", out, gbuf);
    run<5, 4>("V5 everything", out, gbuf);
    return 0;
}
