// LDS-DMA ring throughput per CU without any compute: one persistent 512-thread workgroup per CU streams "stages" of 48 KB
// (48 pieces of 1 KiB; each wave issues 6 global_load_lds_dwordx4 per stage) through a 3-slot LDS ring with the ring GEMM's
// protocol (counted vmcnt, one barrier per stage, two stages in flight).  What varies is where a piece's 64 x 16 B come from:
//   mode 0  GEMM A tile as stored today: 8 rows x 128 B, rows `lda` bytes apart (lda = 3072: K = 1536 bf16)
//   mode 1  4 rows x 256 B
//   mode 2  tile-major activations: the piece is 1 KiB contiguous
// and the footprint: big (HBM-streamed, each byte once) or small (L2-resident, re-read).
// build: hipcc -O3 --offload-arch=gfx950 tools/dmabench.hip -o tools/_bin/dmabench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds16v(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }

template <int NSLOT, int G>
__global__ __launch_bounds__(512, 1) void ring(const char* base, size_t footprint, int lda, int mode, int nstage, unsigned long long* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int STAGE = G * 8 * 1024;  // G pieces per wave x 8 waves
    // a stage = 384 "rows" of 128 B.  Workgroup w streams its own region; stage s of the region:
    auto src = [&](int s, int i) -> const char* {
        const int piece = i * 8 + wave;  // 0..8G-1
        size_t off;
        if (mode == 0) {         // rows lda apart, K-step s selects the 128-B column; new 384-row tile every lda/128 steps
            const int kst = lda / 128, tile = s / kst, ks = s - tile * kst;
            const int row = piece * 8 + (lane >> 3);
            off = ((size_t)tile * (G * 64) + row) * lda + (size_t)ks * 128 + (lane & 7) * 16;
        } else if (mode == 1) {  // 4 rows x 256 B
            const int kst = lda / 256, tile = s / kst, ks = s - tile * kst;
            const int row = piece * 4 + (lane >> 4);
            off = ((size_t)tile * (G * 32) + row) * lda + (size_t)ks * 256 + (lane & 15) * 16;
        } else {                 // contiguous
            off = ((size_t)s * (G * 8) + piece) * 1024 + lane * 16;
        }
        const size_t region = footprint / gridDim.x & ~(size_t)4095;
        return base + (size_t)blockIdx.x * region + off % region;
    };
    auto issue = [&](int s) {
#pragma unroll
        for (int i = 0; i < G; ++i) glds16v(src(s, i), __builtin_amdgcn_readfirstlane(lds0 + (s % NSLOT) * STAGE + (i * 8 + wave) * 1024));
    };
    for (int s = 0; s < NSLOT - 1; ++s) issue(s);
    for (int s = 0; s < nstage; ++s) {
        if (s + NSLOT - 2 < nstage) wait_vm_barrier<G*(NSLOT - 2)>(); else wait_vm_barrier<0>();
        if (s + NSLOT - 1 < nstage) issue(s + NSLOT - 1);
    }
    if (threadIdx.x == 0 && smem[lane * 16] == 123) sink[0] = 1;
}

template <int NSLOT, int G>
void run(const char* buf, size_t big, unsigned long long* sink, hipEvent_t e0, hipEvent_t e1) {
    auto k = ring<NSLOT, G>;
    const int smem = NSLOT * G * 8 * 1024;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    const int grid = 256, nstage = 400 * 6 / G;
    for (int mode : {0, 2})
        for (size_t fp : {big, (size_t)24 << 20, (size_t)8 << 20, (size_t)2 << 20}) {
            float best = 1e9;
            for (int it = 0; it < 4; ++it) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, 0, buf, fp, 3072, mode, nstage, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it && ms < best) best = ms;
            }
            const double bytes = (double)nstage * G * 8 * 1024 * grid;
            printf("slots %d x %2d KB (in flight %3d KB) mode %d footprint %5zu MB: %6.1f GB/s per CU  %5.2f TB/s chip\n", NSLOT, G * 8, (NSLOT - 1) * G * 8, mode,
                   fp >> 20, bytes / grid / best / 1e6, bytes / best / 1e9);
        }
}

int main() {
    const size_t big = 3ull << 30;
    char* buf;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    unsigned long long* sink;
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<3, 6>(buf, big, sink, e0, e1);
    run<6, 3>(buf, big, sink, e0, e1);
    return 0;
}
