// Direct grouped 3x3 convolution with 32 channels per group (stride 1, pad 1): SVTRv2's LocalMixing convs (svtrv2.py:47-63),
// 16-bit types.  As an implicit GEMM this layer re-reads every input pixel's 64-byte group slice nine times from L2 (one K-step per
// tap pair) for a GEMM of only N = 32 columns; here a workgroup stages a (TH+2) x (TW+2) halo tile of ONE group once in LDS and takes
// all nine taps' MFMA operands from it: fragment rows are 16 consecutive pixels of an image row (shifted by the tap), K = the group's
// 32 input channels = one v_mfma 16x16x32.  Pixel and weight rows are 96 bytes apart in LDS (64 used): conflict-free ds_read_b128
// under the gfx950 lane-group rule.  Epilogue: folded-BN bias, GELU / ReLU, optional fp32 residual (the recogniser's residual stream),
// 16-byte stores (output channels permuted in the weight-fragment read for 16-bit outputs, as in gemm_ring.h).
#pragma once
#include "conv_gemm.h"

namespace ocrvi {

constexpr int GC_PS = 96;        // LDS bytes per halo pixel / weight row
constexpr int GC_TH = 4;         // output rows per tile
constexpr int GC_NBX = 5;        // 16-pixel blocks per tile row (tile width 80)

template <typename T, bool F32O>
__global__ __launch_bounds__(256, 2) void gconv32_kernel(const ConvParams p, int TH, int nbx, int bands_y, int bands_x, unsigned hw_magic) {
    constexpr int PS = GC_PS, MB = (GC_TH * GC_NBX + 3) / 4;  // M-blocks per wave (TH <= GC_TH rows per tile)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g4 = lane >> 4;
    const int TW = 16 * nbx, HW = TW + 2, HH = TH + 2;
    char* const halo = smem;
    char* const wl = smem + HH * HW * PS;
    const int gp = blockIdx.y;
    const int ntile = p.n_img * bands_y * bands_x;
    const T* const Xg = (const T*)p.x + p.cin_off + gp * 32;
    // Persistent over tiles (the group, hence the weights, is fixed per workgroup): the next tile's halo is loaded into registers
    // before the current tile's MFMAs and written to LDS after them.  All loads are unconditional (clamped addresses, value masked at
    // the LDS store): a branch around each load would cost one L2 round trip per pass.
    constexpr int HPASS = ((GC_TH + 2) * (16 * GC_NBX + 2) * 4 + 255) / 256, WPASS = (32 * 36 + 255) / 256;
    uint4 hv[HPASS];
    const int htotal = HH * HW * 4;
    auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
        const int bx = t % bands_x;
        t /= bands_x;
        const int by = t % bands_y;
        img = t / bands_y;
        y0 = by * TH;
        x0 = bx * TW;
    };
    auto fetch = [&](int t) {
        int img, y0, x0;
        tile_origin(t, img, y0, x0);
        const T* X = Xg + (size_t)img * p.H * p.W * p.Cin;
#pragma unroll
        for (int k = 0; k < HPASS; ++k) {
            const int i = min(tid + 256 * k, htotal - 1);
            const int pix = i >> 2, c = i & 3;
            const int hy = (int)(((unsigned)pix * hw_magic) >> 16), hx = pix - hy * HW;
            const int iy = min(max(y0 + hy - 1, 0), p.H - 1), ix = min(max(x0 + hx - 1, 0), p.W - 1);
            hv[k] = *(const uint4*)(X + ((size_t)iy * p.W + ix) * p.Cin + c * 8);
        }
    };
    auto stage = [&](int t) {  // registers -> LDS, zeros outside the image
        int img, y0, x0;
        tile_origin(t, img, y0, x0);
#pragma unroll
        for (int k = 0; k < HPASS; ++k) {
            const int i = tid + 256 * k;
            if (i < htotal) {
                const int pix = i >> 2, c = i & 3;
                const int hy = (int)(((unsigned)pix * hw_magic) >> 16), hx = pix - hy * HW;
                const int iy = y0 + hy - 1, ix = x0 + hx - 1;
                const bool in = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                *(uint4*)(halo + pix * PS + c * 16) = in ? hv[k] : make_uint4(0, 0, 0, 0);
            }
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntile) return;
    fetch(tile);
    {   // the group's 9 x 32 x 32 weights, once
        const T* Wg = (const T*)p.w + (size_t)gp * p.Np * p.Kp;  // [n][tap * 32 + c]
        uint4 wv[WPASS];
#pragma unroll
        for (int k = 0; k < WPASS; ++k) {
            const int i = min(tid + 256 * k, 32 * 36 - 1);
            const int n = i / 36, rem = i - n * 36, tap = rem >> 2, c = rem & 3;
            wv[k] = *(const uint4*)(Wg + (size_t)n * p.Kp + tap * 32 + c * 8);
        }
#pragma unroll
        for (int k = 0; k < WPASS; ++k) {
            const int i = tid + 256 * k;
            if (i < 32 * 36) {
                const int n = i / 36, rem = i - n * 36, tap = rem >> 2, c = rem & 3;
                *(uint4*)(wl + (tap * 32 + n) * PS + c * 16) = wv[k];
            }
        }
    }
    stage(tile);
    __syncthreads();
    const int nblk = TH * nbx;
    // output channel (inside the group) of MFMA row lr of N-block nb: fp32 output: 16 nb + lr; 16-bit: 8 (lr >> 2) + 4 nb + (lr & 3)
    int wrow[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wrow[nb] = F32O ? nb * 16 + lr : 8 * (lr >> 2) + 4 * nb + (lr & 3);
    int aoff[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int mb = wave + 4 * m, ty = mb / nbx, tb = mb - ty * nbx;
        aoff[m] = (ty * HW + tb * 16 + lr) * PS + g4 * 16;  // tap (0, 0) = halo pixel (ty, tb * 16 + lr)
    }
    float4 bias_r[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ch = gp * 32 + (F32O ? nb * 16 + 4 * g4 : 8 * g4 + 4 * nb);
        bias_r[nb] = p.bias ? *(const float4*)(p.bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto activate = [&](float (&v)[4]) {
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
    };
    for (;;) {
        const int next = tile + gridDim.x;
        if (next < ntile) fetch(next);
        // ---- MFMA: wave w owns M-blocks w, w + 4, ... (block mb = tile row mb / nbx, 16 pixels from column 16 * (mb % nbx))
        f32x4 acc[MB][2];
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m][0] = acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            uint4 wf[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wf[nb] = *(const uint4*)(wl + (tap * 32 + wrow[nb]) * PS + g4 * 16);
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (wave + 4 * m < nblk) {
                    const uint4 xf = *(const uint4*)(halo + aoff[m] + (r * HW + s) * PS);
                    Mma<T>::half(wf[0], xf, acc[m][0]);
                    Mma<T>::half(wf[1], xf, acc[m][1]);
                }
            }
        }
        __syncthreads();  // every wave is done reading the halo tile
        if (next < ntile) stage(next);
        // ---- epilogue of the current tile (registers only; overlaps the other waves' LDS stores)
        int img, y0, x0;
        tile_origin(tile, img, y0, x0);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int mb = wave + 4 * m, ty = mb / nbx, tb = mb - ty * nbx;
            const int y = y0 + ty, x = x0 + tb * 16 + lr;
            if (mb >= nblk || y >= p.H || x >= p.W) continue;
            const size_t pix = ((size_t)img * p.H + y) * p.W + x;
            float v[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int ch = gp * 32 + (F32O ? nb * 16 + 4 * g4 : 8 * g4 + 4 * nb);
                v[nb][0] = acc[m][nb][0] + bias_r[nb].x; v[nb][1] = acc[m][nb][1] + bias_r[nb].y;
                v[nb][2] = acc[m][nb][2] + bias_r[nb].z; v[nb][3] = acc[m][nb][3] + bias_r[nb].w;
                if (p.res_post) activate(v[nb]);
                if (F32O && p.res_mode == RES_SAME) {
                    const float4 rv = *(const float4*)((const float*)p.res + pix * p.ldr + ch);
                    v[nb][0] += rv.x; v[nb][1] += rv.y; v[nb][2] += rv.z; v[nb][3] += rv.w;
                }
                if (!p.res_post) activate(v[nb]);
                if (F32O) *(float4*)((float*)p.out + pix * p.ldo + p.out_coff + ch) = make_float4(v[nb][0], v[nb][1], v[nb][2], v[nb][3]);
            }
            if (!F32O) {
                union { T e[8]; uint4 u; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pk.e[e] = from_f32<T>(v[0][e]);
                    pk.e[4 + e] = from_f32<T>(v[1][e]);
                }
                *(uint4*)((T*)p.out + pix * p.ldo + p.out_coff + gp * 32 + 8 * g4) = pk.u;
            }
        }
        if (next >= ntile) break;
        tile = next;
        __syncthreads();  // the next tile's halo is in LDS
    }
}

// true when (p, amode) is a LocalMixing-shaped grouped convolution this kernel handles
static inline bool gconv32_eligible(const ConvParams& p, int amode, int esz) {
    static const bool on = !(getenv("OCRVI_GCONV32") && atoi(getenv("OCRVI_GCONV32")) == 0);
    if (!on || amode != AM_CONV3 || esz != 2 || p.groups < 2 || p.Cin_g != 32 || p.N_g != 32 || p.KH != 3) return false;
    if (p.SH != 1 || p.SW != 1 || p.PH != 1 || p.PW != 1 || p.H != p.OH || p.W != p.OW || p.store_mode != ST_NHWC) return false;
    if (p.Kp < 288 || p.Cin % 8 != 0 || p.cin_off % 8 != 0 || ((uintptr_t)p.x & 15) != 0 || ((uintptr_t)p.w & 15) != 0 || p.Kp % 8 != 0) return false;
    const int og = p.out_f32 ? 4 : 8;
    if (p.ldo % og != 0 || p.out_coff % og != 0 || ((uintptr_t)p.out & 15) != 0) return false;
    if (p.res_mode == RES_SAME) return p.out_f32 && p.res_f32 && p.ldr % 4 == 0 && ((uintptr_t)p.res & 15) == 0;
    return p.res_mode == RES_NONE;
}

template <typename T>
static int launch_gconv32(const ConvParams& p, hipStream_t stream) {
    const int nbx = std::min(GC_NBX, cdiv(p.W, 16));
    const int th = p.H % 4 == 0 ? 4 : (p.H % 3 == 0 ? 3 : (p.H < 4 ? p.H : 4));   // rows per tile: avoid a mostly empty last band
    const int bands_x = cdiv(p.W, 16 * nbx), bands_y = cdiv(p.H, th);
    const int smem = (th + 2) * (16 * nbx + 2) * GC_PS + 9 * 32 * GC_PS;
    const unsigned hw_magic = 65536u / (unsigned)(16 * nbx + 2) + 1;  // pix / HW == (pix * magic) >> 16 for pix < 2^11
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const int ntile = p.n_img * bands_y * bands_x;
    const int per_group = std::max(1, 2 * n_cu / p.groups);                  // two persistent workgroups per CU over all groups
    const dim3 grid(cdiv(ntile, cdiv(ntile, per_group)), p.groups);          // equal tile counts
    if (p.out_f32) {
        auto k = gconv32_kernel<T, true>;
        OCRVI_TRY(ensure_max_smem((const void*)k, 80 * 1024));
        hipLaunchKernelGGL(k, grid, dim3(256), smem, stream, p, th, nbx, bands_y, bands_x, hw_magic);
    } else {
        auto k = gconv32_kernel<T, false>;
        OCRVI_TRY(ensure_max_smem((const void*)k, 80 * 1024));
        hipLaunchKernelGGL(k, grid, dim3(256), smem, stream, p, th, nbx, bands_y, bands_x, hw_magic);
    }
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

}  // namespace ocrvi
