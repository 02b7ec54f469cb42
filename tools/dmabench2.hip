// Companion of dmabench.hip: the same 3-slot ring protocol (48-KB stages, two in flight), but a stage's 48 pieces reach LDS either by
// LDS-DMA (global_load_lds_dwordx4), or through registers (global_load_dwordx4 -> VGPR -> ds_write_b128), or split (A-like 32 KB by
// DMA + W-like 16 KB through registers).  Source: contiguous 1-KiB pieces from an L2-resident footprint or streamed from HBM.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16v(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gload16(u32x4& dst, const void* gsrc) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(gsrc) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NREG of the wave's 6 pieces per stage go through registers, the rest by DMA
template <int NREG>
__global__ __launch_bounds__(512, 1) void ring(const char* base, size_t footprint, int nstage, unsigned long long* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int G = 6, NDMA = G - NREG, STAGE = 48 * 1024, NSLOT = 3;
    const size_t region = footprint / gridDim.x & ~(size_t)4095;
    auto src = [&](int s, int i) -> const char* {
        const size_t off = ((size_t)s * 48 + i * 8 + wave) * 1024 + lane * 16;
        return base + (size_t)blockIdx.x * region + off % region;
    };
    u32x4 r[2][NREG ? NREG : 1];
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) glds16v(src(s, i), __builtin_amdgcn_readfirstlane(lds0 + (s % NSLOT) * STAGE + (i * 8 + wave) * 1024));
#pragma unroll
        for (int i = 0; i < NREG; ++i) gload16(r[buf][i], src(s, NDMA + i));
    };
    auto commit = [&](int s, int buf) {  // registers -> LDS
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            asm volatile("" : "+v"(r[buf][i]));
            *(u32x4*)(smem + (s % NSLOT) * STAGE + ((NDMA + i) * 8 + wave) * 1024 + lane * 16) = r[buf][i];
        }
    };
    issue(0, 0);
    issue(1, 1);
    for (int s = 0; s < nstage; s += 2) {  // unrolled by two so the register buffers are static
        if (s + 1 < nstage) wait_vm<G>(); else wait_vm<0>();
        commit(s, 0);
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nstage) issue(s + 2, 0);
        if (s + 1 < nstage) {
            if (s + 2 < nstage) wait_vm<G>(); else wait_vm<0>();
            commit(s + 1, 1);
            __builtin_amdgcn_s_barrier();
            if (s + 3 < nstage) issue(s + 3, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && smem[lane * 16] == 123) sink[0] = 1;
}
template <int NREG> void run(const char* buf, size_t big, unsigned long long* sink, hipEvent_t e0, hipEvent_t e1) {
    auto k = ring<NREG>;
    const int smem = 3 * 48 * 1024, grid = 256, nstage = 400;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    for (size_t fp : {big, (size_t)8 << 20}) {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, 0, buf, fp, nstage, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it && ms < best) best = ms;
        }
        const double bytes = (double)nstage * 48 * 1024 * grid;
        printf("pieces via registers %d/6, footprint %5zu MB: %6.1f GB/s per CU  %5.2f TB/s chip\n", NREG, fp >> 20, bytes / grid / best / 1e6, bytes / best / 1e9);
    }
}
int main() {
    const size_t big = 3ull << 30;
    char* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<0>(buf, big, sink, e0, e1);
    run<2>(buf, big, sink, e0, e1);
    run<3>(buf, big, sink, e0, e1);
    run<6>(buf, big, sink, e0, e1);
    return 0;
}
