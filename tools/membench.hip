// Attainable HBM rate for the layer1 conv3 pattern (read A[M][64], read R[M][256], write O[M][256], bf16) with no GEMM at all:
//  mode 0  flat streaming, 16 B per lane, grid-stride
//  mode 1  tile-shaped: a workgroup owns 128 rows x 128 channels (256-B row pieces at 512-B stride), like conv_gemm's epilogue
//  mode 2  tile-shaped, 8-B per lane in the MFMA C layout (16 rows x 32 B per wave instruction), like gemm_ring's epilogue
// build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o gpurun_out/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void flat(const uint4* A, const uint4* R, uint4* O, size_t nA, size_t nO) {
    size_t i = blockIdx.x * 256ull + threadIdx.x, st = gridDim.x * 256ull;
    uint4 acc = {0, 0, 0, 0};
    for (size_t j = i; j < nA; j += st) { uint4 a = A[j]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
    for (size_t j = i; j < nO; j += st) { uint4 r = R[j]; r.x ^= acc.x; r.y += acc.y; O[j] = r; }
}
// one workgroup per (128-row, 128-channel) tile; ROWS consecutive tiles when persistent
__global__ __launch_bounds__(256) void tiled16(const uint4* A, const uint4* R, uint4* O, int M, int ntile) {
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int mt = t >> 1, nt = t & 1;
        uint4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // A tile: 128 rows x 128 B = 1024 uint4
            uint4 a = A[(size_t)mt * 1024 + i * 256 + threadIdx.x];
            acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
        }
        uint4 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // 128 rows x 256 B: 16 lanes per row, 16 rows per pass
            const int row = i * 16 + (threadIdx.x >> 4), c = threadIdx.x & 15;
            r[i] = R[((size_t)mt * 128 + row) * 32 + nt * 16 + c];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 16 + (threadIdx.x >> 4), c = threadIdx.x & 15;
            r[i].x ^= acc.x; r[i].y += acc.y;
            O[((size_t)mt * 128 + row) * 32 + nt * 16 + c] = r[i];
        }
    }
}
__global__ __launch_bounds__(256) void tiled8(const uint4* A, const uint2* R, uint2* O, int M, int ntile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int mt = t >> 1, nt = t & 1;
        uint4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 a = A[(size_t)mt * 1024 + i * 256 + threadIdx.x];
            acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
        }
        // wave (wm, wn) owns 64 rows x 64 channels: 4 x 4 fragments, lane = (row lr, 4 channels g)
        const int wm = wave >> 1, wn = wave & 1;
        uint2 r[16];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = wm * 64 + b * 16 + lr, ch = nt * 128 + wn * 64 + a * 16 + 4 * g;
                r[b * 4 + a] = R[((size_t)mt * 128 + row) * 64 + ch / 4];
            }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = wm * 64 + b * 16 + lr, ch = nt * 128 + wn * 64 + a * 16 + 4 * g;
                uint2 v = r[b * 4 + a]; v.x ^= acc.x; v.y += acc.y;
                O[((size_t)mt * 128 + row) * 64 + ch / 4] = v;
            }
    }
}
int main() {
    const int M = 1228800;
    const size_t bA = (size_t)M * 64 * 2, bO = (size_t)M * 256 * 2;
    void *A, *R, *O;
    CK(hipMalloc(&A, bA)); CK(hipMalloc(&R, bO)); CK(hipMalloc(&O, bO));
    CK(hipMemset(A, 1, bA)); CK(hipMemset(R, 2, bO)); CK(hipMemset(O, 0, bO));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ntile = (M / 128) * 2;
    const double bytes = (double)bA + 2.0 * bO;
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {256 * 2, 256 * 4, 256 * 8, 256 * 16, ntile}) {
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                CK(hipEventRecord(e0));
                if (mode == 0) flat<<<grid, 256>>>((const uint4*)A, (const uint4*)R, (uint4*)O, bA / 16, bO / 16);
                else if (mode == 1) tiled16<<<grid, 256>>>((const uint4*)A, (const uint4*)R, (uint4*)O, M, ntile);
                else tiled8<<<grid, 256>>>((const uint4*)A, (const uint2*)R, (uint2*)O, M, ntile);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it && ms < best) best = ms;
            }
            printf("mode %d grid %6d: %7.1f us  %6.0f GB/s\n", mode, grid, best * 1e3, bytes / best / 1e6);
        }
    // read-only and write-only references
    return 0;
}
