// Fused ResNet stem of the detector in the f16x2 mode: conv 7x7 / 2 (+ folded BN, ReLU) and max-pool 3x3 / 2 / pad 1 in ONE kernel
// (torchvision resnet50's conv1 / bn1 / relu / maxpool as called from backbone.py:34).  Unfused, the 64-channel half-resolution map is
// the largest tensor of the whole detector (1.26 GB per 16 pages at 960 x 1280) and is written once and read once for nothing: the
// stem conv ran at 0.15 of its roof writing it and the pool at 0.57 of HBM reading it back.  Here it never leaves the CU.
//
// One persistent workgroup per CU (8 waves).  A tile is an 8 x 7 patch of POOLED pixels = a 17 x 15 patch of conv pixels (255: sixteen
// 16-pixel MFMA blocks, two per wave) = a 39 x 36 patch of the zero-padded NHWC4 input (k_nchw3_to_nhwc4_pad):
//   1. (during the previous tile's pooling pass) the tile's input patch, fetched into registers a tile earlier, goes to LDS as (hi, lo)
//      quartets (25 KB)
//   2. (same place) the loads of the patch after it are issued
//   3. conv as an implicit GEMM: K-step r = filter row r = 8 consecutive input pixels x 4 channels (pack_conv's AM_ROWS order), so a
//      lane's 8 k-slots are two adjacent 16-byte pixels of the patch; the 64 x 224 weights stay in LDS for the whole launch (57 KB, staged
//      once in the (hi, lo) quartet form the three-product MFMAs consume); 168 MFMAs per wave
//   4. bias (BN folded), ReLU -> fp32 staging [255 conv pixels][64 channels] in LDS (64 KB); conv pixels outside the map are -inf
//   5. 3 x 3 max over the staging, packed to f16x2 (range-checked), stored: 56 pixels x 256 B
// A pooled row needs conv rows 2 y - 1 .. 2 y + 1, so neighbouring tiles recompute one conv row / column of halo: 255 conv pixels per
// 224 useful ones (14 % more MFMAs, still a small kernel) -- the price of no inter-tile exchange.
#include "gemm_ring.h"
#include "kernels.h"

namespace ocrvi {
OCRVI_RANGE_FLAG_TU()

namespace {
constexpr int SP_PH = 8, SP_PW = 7;                       // pooled patch
constexpr int SP_CR = 2 * SP_PH + 1, SP_CC = 2 * SP_PW + 1;   // conv patch 17 x 15
constexpr int SP_NPX = SP_CR * SP_CC;                     // 255 conv pixels -> 16 blocks of 16 (the last pixel twice)
constexpr int SP_IR = 2 * SP_CR + 5, SP_IC = 2 * SP_CC + 6;   // input patch 39 x 36 (a conv pixel reads 7 rows x 8 pixels)
constexpr int SP_NCH = SP_IR * SP_IC;                     // 1404 16-byte pixels
constexpr int SP_PRE = (SP_NCH + 511) / 512;              // prefetch registers (uint4) per thread: 3
constexpr int SP_KS = 7;                                  // K-steps = filter rows
constexpr int SP_W_BYTES = SP_KS * 64 * 128;              // 57344
// LDS image of the patch, in 16-byte units: input row rho at (rho >> 1) * 79 + (rho & 1) * 36; inside a row 18 units of hi quartets (unit q =
// the hi halves of pixels 2 q and 2 q + 1: what a lane's MFMA operand takes, a lane always starts at an even pixel) and 18 of lo quartets.
// The row-pair stride of 79 = 15 (mod 16) makes the 16 lanes of a read conflict-free although every 16-pixel block spans two conv rows
// (15 pixels each): the pixels after the row break continue in the next bank group (simulated with the gfx950 lane-group rule: 4.25
// LDS cycles per ds_read_b128 against 11 for plain [row][36 pixels] rows).
constexpr int SP_PAIR = 79, SP_ODD = 36, SP_LO = 18;
constexpr int SP_P_UNITS = ((SP_IR - 1) >> 1) * SP_PAIR + SP_ODD + 2 * SP_LO;   // 1573 >= the last unit + 1
constexpr int SP_P_BYTES = ((SP_P_UNITS * 16 + 255) / 256) * 256;
constexpr int SP_S_BYTES = 256 * 64 * 4;                  // 65536
constexpr int SP_SMEM = SP_W_BYTES + SP_P_BYTES + SP_S_BYTES;
static_assert(SP_NPX <= 256 && SP_NPX > 240, "sixteen MFMA blocks");
}  // namespace

// xpad [N][Hp][Wp][4] f16x2 (zero border: 3 rows / columns before the image), w [64][Kp = 224] f16x2 in AM_ROWS order (chunk format),
// y [N][OH][OW][64] f16x2 with OH = H / 4, OW = W / 4 (H, W = the image; the conv map is H / 2 x W / 2)
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const char* __restrict__ xpad, const char* __restrict__ w, const float* __restrict__ bias,
                                                          float wscale, char* __restrict__ y, int N, int CH, int CW, int OH, int OW, int Hp,
                                                          int Wp, int tyN, int txN, int dbg, unsigned long long* __restrict__ prof) {
    typedef f16x2_t T;
    typedef Mma<T>::u4v U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ws = smem;
    char* const Ps = smem + SP_W_BYTES;
    char* const Ss = smem + SP_W_BYTES + SP_P_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int total = N * tyN * txN;

    // ---- weights -> LDS once, in quartet form: k-slots 8 q .. 8 q + 7 of (filter row r, channel n) are memory chunks 2 q, 2 q + 1
    // ([4 hi | 4 lo] each); LDS chunk 2 q takes the two hi halves, chunk 2 q + 1 the two lo halves (XOR-swizzled by n as every [row][128 B] image)
    for (int i = tid; i < SP_KS * 64 * 4; i += 512) {
        const int q = i & 3, n = (i >> 2) & 63, r = i >> 8;
        const uint4* src = (const uint4*)(w + ((size_t)n * (SP_KS * 32) + r * 32 + 8 * q) * 4);
        const uint4 c0 = src[0], c1 = src[1];
        char* row = Ws + (r * 64 + n) * 128;
        const int sw = swz128(n);
        *(uint4*)(row + (((2 * q) ^ sw) << 4)) = make_uint4(c0.x, c0.y, c1.x, c1.y);
        *(uint4*)(row + (((2 * q + 1) ^ sw) << 4)) = make_uint4(c0.z, c0.w, c1.z, c1.w);
    }

    auto tile_of = [&](int t, int& img, int& py0, int& px0) {
        img = t / (tyN * txN);
        const int rem = t - img * tyN * txN, ty = rem / txN;
        py0 = ty * SP_PH;
        px0 = (rem - ty * txN) * SP_PW;
    };
    static_assert(SP_PRE == 3, "three prefetch registers");
    uint4 pre0, pre1, pre2;
    // pixel tid + 512 k of the tile's input patch (rows / columns outside the padded image are clamped: only conv pixels outside the map,
    // whose results are discarded, read them; no branch and no array: a conditionally written array element stays in scratch memory)
    auto fetch1 = [&](int t, int k) -> uint4 {
        int img, py0, px0;
        tile_of(t, img, py0, px0);
        const int iy0 = 4 * py0 - 2, ix0 = 4 * px0 - 2;
        const int idx = min(tid + 512 * k, SP_NCH - 1);
        const int pr = idx / SP_IC, pc = idx - pr * SP_IC;
        const int iy = min(max(iy0 + pr, 0), Hp - 1), ix = min(max(ix0 + pc, 0), Wp - 1);
        return *(const uint4*)(xpad + (((size_t)img * Hp + iy) * Wp + ix) * 16);
    };

    auto put = [&](int k, const uint4& v) {   // pixel tid + 512 k of the patch -> its hi / lo quartet halves
        const int idx = tid + 512 * k;
        if (idx < SP_NCH) {
            const int pr = idx / SP_IC, pc = idx - pr * SP_IC;
            char* d = Ps + ((pr >> 1) * SP_PAIR + (pr & 1) * SP_ODD + (pc >> 1)) * 16 + (pc & 1) * 8;
            *(uint2*)d = make_uint2(v.x, v.y);
            *(uint2*)(d + SP_LO * 16) = make_uint2(v.z, v.w);
        }
    };
    // this wave's two MFMA blocks: conv pixel p = 16 blk + lr -> (i, j) of the 17 x 15 patch; patch byte offset of its first input pixel
    int a_off[2];
    bool dup[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int p0 = (wave * 2 + b) * 16 + lr, p = min(p0, SP_NPX - 1);
        dup[b] = p0 >= SP_NPX;
        const int i = p / SP_CC, j = p - i * SP_CC;
        a_off[b] = (i * SP_PAIR + j + g) * 16;          // hi quartet of input pixels 2 (j + g), 2 (j + g) + 1 of input row 2 i
    }
    const int swb = swz128(lr);
    const int fob0 = ((2 * g) ^ swb) << 4, fob1 = ((2 * g + 1) ^ swb) << 4;
    float4 bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) bv[a] = *(const float4*)(bias + 16 * a + 4 * g);
    unsigned long long range_mask = 0;

    // Two barriers per tile: [MFMAs of tile t from the patch, results -> staging] | A | [patch of tile t + 1 -> LDS, loads of tile t + 2's
    // patch issued, max-pool of tile t from the staging -> global] | B.  (A: every wave is done reading the patch and writing the staging;
    // B: the next patch is complete and the staging is free.)
    long long tk[6] = {0, 0, 0, 0, 0, 0}, tk0 = 0;   // (OCRVI_STEM_DBG & 8: shader-clock cycles per phase, summed over the waves into prof[])
    auto tick = [&](int k) {
        if (dbg & 8) {
            const long long c = clock64();
            tk[k] += c - tk0;
            tk0 = c;
        }
    };
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int G = (int)gridDim.x;
    {
        const int tf = min(t, total - 1);
        pre0 = fetch1(tf, 0); pre1 = fetch1(tf, 1); pre2 = fetch1(tf, 2);
        put(0, pre0);
        put(1, pre1);
        put(2, pre2);
        const int tn = min(t + G, total - 1);
        pre0 = fetch1(tn, 0); pre1 = fetch1(tn, 1); pre2 = fetch1(tn, 2);
    }
    __syncthreads();   // (also: the weights are in LDS)
    if (dbg & 8) tk0 = clock64();
    for (; t < total; t += G) {
        int img, py0, px0;
        tile_of(t, img, py0, px0);

        // ---- conv: 7 K-steps x (2 pixel blocks x 4 channel blocks) x 3 products
        f32x4 acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][0] = acc[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (software-pipelined by hand: the 12 fragment reads of filter row r + 1 are issued before the 24 MFMAs of row r -- left to itself
        // hipcc reads every weight fragment right in front of its first MFMA and waits for it, ~5 exposed LDS round trips per row)
        if (!(dbg & 2)) {    // (OCRVI_STEM_DBG, development, wrong results: 1 no pooling pass, 2 no MFMA phase, 4 no staging writes)
            U xH[2][2], xL[2][2], wH[2][4], wL[2][4];
            auto load_row = [&](auto R, auto S) {
                constexpr int r = decltype(R)::value, sl = decltype(S)::value;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const char* ap = Ps + a_off[b] + ((r >> 1) * SP_PAIR + (r & 1) * SP_ODD) * 16;
                    const uint4 h = *(const uint4*)ap, l = *(const uint4*)(ap + SP_LO * 16);
                    xH[sl][b] = (U){h.x, h.y, h.z, h.w};
                    xL[sl][b] = (U){l.x, l.y, l.z, l.w};
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const char* wr = Ws + (r * 64 + 16 * a + lr) * 128;
                    const uint4 h = *(const uint4*)(wr + fob0), l = *(const uint4*)(wr + fob1);
                    wH[sl][a] = (U){h.x, h.y, h.z, h.w};
                    wL[sl][a] = (U){l.x, l.y, l.z, l.w};
                }
            };
            auto mma_row = [&](auto S) {
                constexpr int sl = decltype(S)::value;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    Mma<T>::three(wH[sl][a], wL[sl][a], xH[sl][0], xL[sl][0], acc[a][0]);
                    Mma<T>::three(wH[sl][a], wL[sl][a], xH[sl][1], xL[sl][1], acc[a][1]);
                }
            };
            auto row = [&](auto R) {
                constexpr int r = decltype(R)::value;
                if constexpr (r + 1 < SP_KS) load_row(IC<r + 1>{}, IC<(r + 1) & 1>{});
                __builtin_amdgcn_sched_barrier(0);
                mma_row(IC<r & 1>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            load_row(IC<0>{}, IC<0>{});
            row(IC<0>{}); row(IC<1>{}); row(IC<2>{}); row(IC<3>{}); row(IC<4>{}); row(IC<5>{}); row(IC<6>{});
        }
        tick(0);
        // ---- bias + ReLU -> staging (fp32, 16-byte chunk c of pixel p at chunk c ^ (p & 15): conflict-free writes and pool reads);
        // conv pixels outside the map are -inf, the max-pool's padding value
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int p = (wave * 2 + b) * 16 + lr;
            if (!dup[b] && !(dbg & 4)) {
                const int i = p / SP_CC, j = p - i * SP_CC;
                const int cr = 2 * py0 - 1 + i, cc = 2 * px0 - 1 + j;
                const bool ok = (unsigned)cr < (unsigned)CH && (unsigned)cc < (unsigned)CW;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float4 v;
                    v.x = ok ? fmaxf(acc[a][b][0] * wscale + bv[a].x, 0.f) : -INFINITY;
                    v.y = ok ? fmaxf(acc[a][b][1] * wscale + bv[a].y, 0.f) : -INFINITY;
                    v.z = ok ? fmaxf(acc[a][b][2] * wscale + bv[a].z, 0.f) : -INFINITY;
                    v.w = ok ? fmaxf(acc[a][b][3] * wscale + bv[a].w, 0.f) : -INFINITY;
                    *(float4*)(Ss + p * 256 + (((4 * a + g) ^ (p & 15)) << 4)) = v;
                }
            }
        }
        tick(1);
        __syncthreads();   // A
        tick(2);
        put(0, pre0);
        put(1, pre1);
        put(2, pre2);
        {   // the patch after next (the last tiles fetch the last tile again: no branch around the loads)
            const int tn = min(t + 2 * G, total - 1);
            pre0 = fetch1(tn, 0); pre1 = fetch1(tn, 1); pre2 = fetch1(tn, 2);
        }
        tick(3);
        // ---- 3 x 3 / 2 max-pool over the staging -> f16x2, 56 pixels x 16 chunks of 4 channels
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 512 * k;
            if (item < SP_PH * SP_PW * 16 && !(dbg & 1)) {
                const int q = item >> 4, c = item & 15;
                const int qy = q / SP_PW, qx = q - qy * SP_PW;
                const int py = py0 + qy, px = px0 + qx;
                if (py < OH && px < OW) {
                    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                    for (int di = 0; di < 3; ++di)
#pragma unroll
                        for (int dj = 0; dj < 3; ++dj) {
                            const int p = (2 * qy + di) * SP_CC + 2 * qx + dj;
                            const float4 v = *(const float4*)(Ss + p * 256 + ((c ^ (p & 15)) << 4));
                            m[0] = fmaxf(m[0], v.x); m[1] = fmaxf(m[1], v.y); m[2] = fmaxf(m[2], v.z); m[3] = fmaxf(m[3], v.w);
                        }
                    range_mask |= f16x2_out_of_range(m);
                    *(uint4*)(y + ((((size_t)img * OH + py) * OW + px) * 64 + 4 * c) * 4) = Chunk<T>::pack(m);
                }
            }
        }
        tick(4);
        __syncthreads();   // B
        tick(5);
    }
    f16x2_raise(range_mask);
    if ((dbg & 8) && prof && lane == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(prof + k, (unsigned long long)tk[k]);
}

bool stem_pool_eligible(int dtype, int cout, int Kp, int KH, int H, int W) {
    static const bool off = getenv("OCRVI_STEM_FUSED") && atoi(getenv("OCRVI_STEM_FUSED")) == 0;
    return !off && dtype == OCRVI_F16X2 && cout == 64 && Kp == SP_KS * 32 && KH == 7 && H % 4 == 0 && W % 4 == 0;
}

// xpad: the padded NHWC4 input of the stem conv (Hp >= H + 6, Wp >= W + 8); y: [N][H / 4][W / 4][64]
int k_stem_pool(int dtype, const void* xpad, const void* w, const float* bias, float wscale, void* y, int N, int H, int W, int Hp, int Wp,
                hipStream_t s) {
    OCRVI_CHECK(dtype == OCRVI_F16X2 && xpad && w && bias && y && N > 0 && H % 4 == 0 && W % 4 == 0 && Hp >= H + 6 && Wp >= W + 8, OCRVI_EINVAL,
                "stem_pool: bad arguments (N=%d, %dx%d, padded %dx%d)", N, H, W, Hp, Wp);
    const int CH = H / 2, CW = W / 2, OH = H / 4, OW = W / 4;
    const int tyN = cdiv(OH, SP_PH), txN = cdiv(OW, SP_PW);
    OCRVI_CHECK((size_t)N * tyN * txN < ((size_t)1 << 30), OCRVI_EINVAL, "stem_pool: too many tiles");
    ProfScope ps_("stem_pool_f16x2", 2.0 * N * CH * CW * 64 * 147, (double)N * (H * (double)W * 16 + OH * (double)OW * 256), s);
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const int total = N * tyN * txN;
    const int grid = cdiv(total, cdiv(total, std::min(total, n_cu)));   // one persistent workgroup per CU, equal tile counts
    static const int dbg = getenv("OCRVI_STEM_DBG") ? atoi(getenv("OCRVI_STEM_DBG")) : 0;
    OCRVI_TRY(ensure_max_smem((const void*)stem_pool_kernel, SP_SMEM));
    unsigned long long* prof = nullptr;
    if (dbg & 8) {   // development: phase cycles, printed per launch (synchronises)
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) OCRVI_HIP(hipMalloc((void**)&dbuf, 64));
        OCRVI_HIP(hipMemsetAsync(dbuf, 0, 64, s));
        prof = dbuf;
    }
    hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(512), SP_SMEM, s, (const char*)xpad, (const char*)w, bias, wscale, (char*)y, N, CH, CW, OH,
                       OW, Hp, Wp, tyN, txN, dbg, prof);
    OCRVI_HIP(hipGetLastError());
    if (prof) {
        unsigned long long h[6];
        OCRVI_HIP(hipMemcpyAsync(h, prof, 48, hipMemcpyDeviceToHost, s));
        OCRVI_HIP(hipStreamSynchronize(s));
        const double wv = 8.0 * grid, tiles = (double)total / grid;
        fprintf(stderr, "stem_pool grid %d tiles/wg %.1f: cycles per wave and tile: mfma %.0f staging %.0f barrier A %.0f patch write + fetch issue %.0f pool %.0f barrier B %.0f\n",
                grid, tiles, h[0] / wv / tiles, h[1] / wv / tiles, h[2] / wv / tiles, h[3] / wv / tiles, h[4] / wv / tiles, h[5] / wv / tiles);
    }
    return OCRVI_OK;
}

}  // namespace ocrvi
