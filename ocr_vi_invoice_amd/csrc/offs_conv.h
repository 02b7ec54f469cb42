// Direct 3x3 convolution for the 27-channel offset / mask conv of DeformableConv2d (model/det/dcn.py:42-46: conv -> 18 offsets +
// sigmoid(9 mask logits)), stride 1 or 2, pad 1.  As an implicit GEMM (conv_gemm_kernel, 128x32 tile) this layer pays two barriers
// and a register-staged global -> LDS hop per 128-byte K-step for 8 (16-bit) or 64 (fp32) MFMAs per wave: it ran at 6-22 % (16-bit) and
// 38-46 % (fp32) of the matrix peak.  Here a persistent workgroup of 8 waves owns a TH x 16 tile of output pixels; per channel block of
// 128 bytes it DMAs (global_load_lds) the tile's input patch -- (TH-1) S + 3 rows of 15 S + 3 pixels, every pixel once -- and the nine
// taps' 32 x 128-byte weight rows into one of two LDS stages, and takes all nine taps' MFMA operands from the patch: the A fragment of tap
// (dy, dx) for output row y is the 16 patch pixels ((y S + dy) PW + x S + dx), x = 0..15.  One wait + barrier per channel block, the next
// block's (or the next tile's first) stage in flight meanwhile.  Patch rows are XOR-swizzled by patch-pixel index (swz128); fragment reads
// that start at an arbitrary pixel are 2-way bank-conflicted (simulated with the gfx950 ds_read_b128 lane groups), which the fp32 build
// hides behind its 32-cycle MFMAs.  Accumulation order is (channel block, tap, channel) instead of the implicit GEMM's (tap, channel):
// results differ from it by fp32 rounding only.  Measured (MI355X, 16 pages 960x1280, all 13 instances of a detector forward): f16 0.91 ->
// 0.58 ms, fp32 2.47 -> 2.03 ms; the stride-1 tile height (16 / 8 / 4 rows) is picked per shape for the fewest rounds of workgroups over
// the CUs (60x80 maps: 320 tiles of 16 rows are 1.25 rounds, 640 of 8 rows 2.5).  OCRVI_OFFS_DIRECT=0 falls back to the implicit GEMM.
#pragma once
#include "gemm_ring.h"

namespace ocrvi {

template <typename T, int S, int TH_> struct OffsCfg {
    static constexpr int EPC = TypeInfo<T>::EPC, CBE = 8 * EPC;     // elements per 128-byte channel block
    static constexpr int TW = 16, TH = TH_;                         // output tile: TH rows (16, 8 or 4) of one 16-pixel MFMA block each
    static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, NPIX = PH * PW;
    static constexpr int NI_P = (NPIX * 8 + 63) / 64;               // 1-KiB DMA instructions per patch stage
    static constexpr int NI_W = 9 * 32 * 8 / 64;                    // 36: nine taps x 32 weight rows
    static constexpr int PATCH = NI_P * 1024, STAGE = PATCH + NI_W * 1024;
    static constexpr int SLOTS = (NI_P + NI_W + 7) / 8;             // DMA instructions per wave per stage (8 waves)
    static constexpr int SMEM = 2 * STAGE;
    static constexpr int MBW = TH == 16 ? 2 : 1, NBW = TH >= 8 ? 2 : 1;   // tile rows and 16-channel blocks per wave (8 waves)
    static_assert(TH == 16 || TH == 8 || TH == 4, "tile height");
};

template <typename T, int S, int TH>
__global__ __launch_bounds__(512, 1) void offs_conv_kernel(const ConvParams p, int tiles_x, int tiles_y) {
    using C = OffsCfg<T, S, TH>;
    constexpr int PW = C::PW, NPIX = C::NPIX, NI_P = C::NI_P, NI_W = C::NI_W, SLOTS = C::SLOTS, MBW = C::MBW, NBW = C::NBW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int esz = (int)sizeof(T);
    const int ncb = p.Cin_g / C::CBE;
    const int ntile = p.n_img * tiles_y * tiles_x;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int my_tiles = first < ntile ? (ntile - first + G - 1) / G : 0;
    const int nstage = my_tiles * ncb;
    const char* const X = (const char*)p.x + (size_t)p.cin_off * esz;
    const char* const Wp = (const char*)p.w;
    const char* const zero = (const char*)p.zero_page + (lane & 7) * 16;
    const int pix_b = p.Cin * esz, wrow_b = p.Kp * esz;

    // ---- DMA issue cursor: stage q = (my tile q / ncb, channel block q % ncb); instruction i = wave + 8 j of a stage is patch piece i
    // (i < NI_P: 8 patch pixels x 128 B) or weight piece i - NI_P (8 rows of one tap).  Per-lane sources for channel block 0 are
    // rebuilt once per tile; every further block adds 128 bytes.
    const char* src[SLOTS];
    int i_tile = 0, i_cb = 0, i_q = 0;
    auto setup_tile = [&](int t) {
        const int tile = first + t * G;
        const int tx = tile % tiles_x, r = tile / tiles_x, ty = r % tiles_y, img = r / tiles_y;
        const int y0 = ty * C::TH * S - 1, x0 = tx * C::TW * S - 1;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int i = wave + 8 * j;
            const char* s = zero;
            if (i < NI_P) {
                const int pp = i * 8 + (lane >> 3), cq = lane & 7;
                const int py = pp / PW, px = pp - py * PW;
                const int iy = y0 + py, ix = x0 + px;
                if (pp < NPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    s = X + ((size_t)(img * p.H + iy) * p.W + ix) * pix_b + ((cq ^ swz128(pp)) << 4);
            } else if (i < NI_P + NI_W) {
                const int iw = i - NI_P, tap = iw >> 2, n = (iw & 3) * 8 + (lane >> 3), cq = lane & 7;
                s = Wp + (size_t)n * wrow_b + (size_t)tap * p.Cin_g * esz + ((cq ^ swz128(n)) << 4);
            }
            src[j] = s;
        }
    };
    auto issue_stage = [&]() {
        if (i_cb == 0) setup_tile(i_tile);
        const unsigned base = lds0 + (i_q & 1) * C::STAGE;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int i = wave + 8 * j;
            if (i < NI_P + NI_W) {                                     // (wave-uniform)
                const bool is_zero = src[j] == zero;
                glds16v(is_zero ? src[j] : src[j] + (size_t)i_cb * 128, __builtin_amdgcn_readfirstlane(base + i * 1024));
            }
        }
        ++i_q;
        if (++i_cb == ncb) { i_cb = 0; ++i_tile; }
    };

    if (nstage > 0) issue_stage();
    f32x4 acc[NBW][MBW];
    const int mb0 = TH == 16 ? 2 * wave : (TH == 8 ? wave : (wave >> 1)), nb0 = TH == 4 ? (wave & 1) : 0;
    for (int q = 0; q < nstage; ++q) {
        const int cb = q % ncb, t = q / ncb;
        wait_vm_barrier<0>();                    // stage q has landed (own pieces) and every wave is past its reads of the other buffer
        if (q + 1 < nstage) issue_stage();       // ... which the next stage now overwrites
        if (cb == 0) {
#pragma unroll
            for (int a = 0; a < NBW; ++a)
#pragma unroll
                for (int b = 0; b < MBW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const char* const Ps = smem + (q & 1) * C::STAGE;
        const char* const Ws = Ps + C::PATCH;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            if constexpr (IsSplit<T>::value) {   // f16x2: three MFMAs per fragment pair (Mma<f16x2_t>::regroup / three)
                typedef typename Mma<T>::u4v U;
                U xH[MBW], xL[MBW];
#pragma unroll
                for (int b = 0; b < MBW; ++b) {
                    const int pp = ((mb0 + b) * S + dy) * PW + lr * S + dx;
                    const char* r = Ps + pp * 128;
                    Mma<T>::regroup(*(const uint4*)(r + (((2 * g) ^ swz128(pp)) << 4)), *(const uint4*)(r + (((2 * g + 1) ^ swz128(pp)) << 4)), xH[b], xL[b]);
                }
#pragma unroll
                for (int a = 0; a < NBW; ++a) {
                    const int n = (nb0 + a) * 16 + lr;
                    const char* r = Ws + (tap * 32 + n) * 128;
                    U wH, wL;
                    Mma<T>::regroup(*(const uint4*)(r + (((2 * g) ^ swz128(n)) << 4)), *(const uint4*)(r + (((2 * g + 1) ^ swz128(n)) << 4)), wH, wL);
#pragma unroll
                    for (int b = 0; b < MBW; ++b) Mma<T>::three(wH, wL, xH[b], xL[b], acc[a][b]);
                }
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 wf[NBW], xf[MBW];
#pragma unroll
                for (int a = 0; a < NBW; ++a) {
                    const int n = (nb0 + a) * 16 + lr;
                    wf[a] = *(const uint4*)(Ws + (tap * 32 + n) * 128 + (((2 * g + h) ^ swz128(n)) << 4));
                }
#pragma unroll
                for (int b = 0; b < MBW; ++b) {
                    const int pp = ((mb0 + b) * S + dy) * PW + lr * S + dx;
                    xf[b] = *(const uint4*)(Ps + pp * 128 + (((2 * g + h) ^ swz128(pp)) << 4));
                }
#pragma unroll
                for (int a = 0; a < NBW; ++a)
#pragma unroll
                    for (int b = 0; b < MBW; ++b) Mma<T>::half(wf[a], xf[b], acc[a][b]);
            }
        }
        if (cb == ncb - 1) {   // the tile is complete: bias, sigmoid on the mask channels (dcn.py:46), fp32 [pixel][32] rows
            const int tile = first + t * G;
            const int tx = tile % tiles_x, r = tile / tiles_x, ty = r % tiles_y, img = r / tiles_y;
#pragma unroll
            for (int b = 0; b < MBW; ++b) {
                const int oy = ty * C::TH + mb0 + b, ox = tx * C::TW + lr;
                if (oy >= p.OH || ox >= p.OW) continue;
                float* o = (float*)p.out + ((size_t)(img * p.OH + oy) * p.OW + ox) * 32;
#pragma unroll
                for (int a = 0; a < NBW; ++a) {
                    const int n = (nb0 + a) * 16 + 4 * g;
                    float v[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int nn = n + rr;
                        float tv = nn < p.N_g ? unscale<T>(acc[a][b][rr], p.wscale) + p.bias[nn] : 0.f;
                        if (nn >= 18) tv = nn < p.N_g ? 1.0f / (1.0f + expf(-tv)) : 0.f;
                        v[rr] = tv;
                    }
                    *(float4*)(o + n) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    wait_vm_only<0>();
}

inline bool offs_conv_eligible(const ConvParams& p, int amode, int dtype) {
    static const bool on = !(getenv("OCRVI_OFFS_DIRECT") && atoi(getenv("OCRVI_OFFS_DIRECT")) == 0);   // A/B switch
    const int cbe = 128 / (int)dtype_size(dtype);
    return on && amode == AM_CONV3 && p.store_mode == ST_DCN_OFFS && p.groups == 1 && p.N_g <= 32 && p.Np == 32 && p.bias &&
           p.Cin_g % cbe == 0 && p.Cin % (16 / (int)dtype_size(dtype)) == 0 && p.cin_off % (16 / (int)dtype_size(dtype)) == 0 &&
           p.Kp == 9 * p.Cin_g && p.PH == 1 && p.PW == 1 && p.SH == p.SW && (p.SH == 1 || p.SH == 2) &&
           p.OH == (p.H - 1) / p.SH + 1 && p.OW == (p.W - 1) / p.SW + 1 && ((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0 &&
           ((uintptr_t)p.out & 15) == 0;
}

template <typename T, int S, int TH>
static int launch_offs_conv_t(const ConvParams& p_in, int n_cu, hipStream_t stream) {
    using C = OffsCfg<T, S, TH>;
    static_assert(C::SMEM <= 160 * 1024, "two stages must fit the LDS");
    ConvParams p = p_in;
    void* dump = nullptr;
    OCRVI_TRY(ring_pages(&p.zero_page, &dump));
    const int tiles_x = cdiv(p.OW, C::TW), tiles_y = cdiv(p.OH, C::TH);
    const int ntile = p.n_img * tiles_y * tiles_x;
    int grid = std::min(ntile, n_cu);
    grid = cdiv(ntile, cdiv(ntile, grid));   // equal tile counts
    auto kern = offs_conv_kernel<T, S, TH>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, C::SMEM));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C::SMEM, stream, p, tiles_x, tiles_y);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T>
static int launch_offs_conv(const ConvParams& p, hipStream_t stream) {
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    if (p.SH == 2) return launch_offs_conv_t<T, 2, 4>(p, n_cu, stream);   // (a taller stride-2 patch does not fit two LDS stages)
    // stride 1: the tile height that minimises (rounds of workgroups over the CUs) x (rows per tile + the per-stage fixed cost)
    int best = 16, best_cost = 1 << 30;
    for (int th = 16; th >= 4; th >>= 1) {
        const int tiles = p.n_img * cdiv(p.OH, th) * cdiv(p.OW, 16);
        const int cost = cdiv(tiles, n_cu) * (th + 2);
        if (cost < best_cost) { best = th; best_cost = cost; }
    }
    const char* fe = getenv("OCRVI_OFFS_TH");           // test / experiment knob (read per launch so that a test can sweep it)
    const int force = fe ? atoi(fe) : 0;
    if (force == 16 || force == 8 || force == 4) best = force;
    if (best == 16) return launch_offs_conv_t<T, 1, 16>(p, n_cu, stream);
    if (best == 8) return launch_offs_conv_t<T, 1, 8>(p, n_cu, stream);
    return launch_offs_conv_t<T, 1, 4>(p, n_cu, stream);
}

}  // namespace ocrvi
