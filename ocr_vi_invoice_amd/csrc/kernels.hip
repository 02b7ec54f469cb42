// Memory-bound kernels of the hot path: layout conversion, max-pool, LayerNorm, FRM column attention, ASF,
// DB head tail, CTC log-softmax / argmax / collapse.  wave64 reductions, 8..16-byte vector accesses.
#include "kernels.h"

namespace ocrvi {
OCRVI_RANGE_FLAG_TU()   // binds this unit's f16x2 range-flag pointer (common.h)


#define DISPATCH_DT(dt, CALL)                                        \
    switch (dt) {                                                    \
        case OCRVI_F32: { using T = float; CALL; break; }            \
        case OCRVI_BF16: { using T = bf16_t; CALL; break; }          \
        case OCRVI_F16: { using T = f16_t; CALL; break; }            \
        case OCRVI_F16X2: { using T = f16x2_t; CALL; break; }        \
        default: set_error("unknown dtype %d", dt); return OCRVI_EINVAL; \
    }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// 64-lane sum without the LDS crossbar: four DPP adds leave every 16-lane row's total in all of its lanes, the four row totals are read
// into SGPRs and added (the result is wave-uniform).  ~12 full-rate VALU instructions against 6 dependent ds_bpermute round trips.
template <int CTRL> __device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x124>(v);     // row_ror:4
    v = dpp_add<0x128>(v);     // row_ror:8
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------ NCHW3 f32 -> padded NHWC4 T
template <typename T>
__global__ void nchw3_to_nhwc4_pad_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int H, int W, int pt, int pl, int Hp, int Wp) {
    const size_t total = (size_t)N * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const size_t t = i / Wp;
        const int yp = (int)(t % Hp), n = (int)(t / Hp);
        const int yy = yp - pt, xx = xp - pl;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const size_t plane = (size_t)H * W, o = (size_t)n * 3 * plane + (size_t)yy * W + xx;
            f[0] = x[o]; f[1] = x[o + plane]; f[2] = x[o + 2 * plane];
        }
        store4<T>(y + i * 4, f);
    }
}
int k_nchw3_to_nhwc4_pad(int dtype, const float* x, void* y, int N, int H, int W, int pad_t, int pad_l, int Hp, int Wp, hipStream_t s) {
    OCRVI_CHECK(x && y && N > 0 && Hp >= H + pad_t && Wp >= W + pad_l, OCRVI_EINVAL, "nhwc4 pad: bad shape");
    ProfScope ps_("nchw_to_nhwc4", 0.0, (double)N*(H*W*12.0+(double)Hp*Wp*4*dtype_size(dtype)), s);
    const size_t total = (size_t)N * Hp * Wp;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(nchw3_to_nhwc4_pad_kernel<T>, dim3(grid), dim3(256), 0, s, x, (T*)y, N, H, W, pad_t, pad_l, Hp, Wp));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// ------------------------------------------------------------------ maxpool 3x3 s2 p1, NHWC
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH, int OW) {
    constexpr int EPC = TypeInfo<T>::EPC;
    const int cch = C / EPC;
    const size_t total = (size_t)N * OH * OW * cch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cch);
        size_t t = i / cch;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int n = (int)(t / OH);
        float m[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 - 1 + r;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int iw = ow * 2 - 1 + q;
                if ((unsigned)iw >= (unsigned)W) continue;
                float f[EPC];
                Chunk<T>::unpack(*(const uint4*)(x + (((size_t)n * H + ih) * W + iw) * C + cc * EPC), f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) m[e] = fmaxf(m[e], f[e]);
            }
        }
        *(uint4*)(y + (((size_t)n * OH + oh) * OW + ow) * C + cc * EPC) = Chunk<T>::pack(m);
    }
}
int k_maxpool3x3s2(int dtype, const void* x, void* y, int N, int H, int W, int C, hipStream_t s) {
    OCRVI_CHECK(x && y && C % 8 == 0 && H >= 2 && W >= 2, OCRVI_EINVAL, "maxpool: bad shape C=%d", C);
    ProfScope ps_("maxpool3x3s2", 0.0, (double)N*H*W*C*dtype_size(dtype)*1.25, s);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const size_t total = (size_t)N * OH * OW * (C / (dtype_size(dtype) == 4 ? 4 : 8));
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(maxpool3x3s2_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)x, (T*)y, N, H, W, C, OH, OW));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// ------------------------------------------------------------------ LayerNorm
// A row of D <= 128 elements takes half a wave (two rows per wave), longer rows a whole wave with up to 4 float4 per lane.  Every
// wave walks rows with a grid stride and loads its next row before it reduces the current one, so a CU keeps two rows per wave in flight.
template <int W> __device__ __forceinline__ float seg_sum(float v) {  // sum over aligned groups of W lanes
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <typename TI, typename TO, int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, TO* __restrict__ out, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows, int D) {
    constexpr int RPW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = gridDim.x * (blockDim.x >> 6);
    float g[NV][4], b[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * LPR + l) * 4;
        if (c < D) {
            load4<float>(gamma + c, g[i]);
            load4<float>(beta + c, b[i]);
        }
    }
    float cur[NV][4], nxt[NV][4];
    auto fetch = [&](int row, float (&v)[NV][4]) {
        if (row >= rows) return;
        const TI* xr = x + (size_t)row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * LPR + l) * 4;
            if (c < D) load4<TI>(xr + c, v[i]);
        }
    };
    int row = wave * RPW + sub;
    fetch(row, cur);
    for (; row - sub < rows; row += nwave * RPW) {     // (wave-uniform trip count: the shuffles need every lane)
        fetch(row + nwave * RPW, nxt);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((i * LPR + l) * 4 < D) sum += cur[i][0] + cur[i][1] + cur[i][2] + cur[i][3];
        const float mean = seg_sum<LPR>(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((i * LPR + l) * 4 < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = cur[i][e] - mean; sq += d * d; }
            }
        const float rstd = rsqrtf(seg_sum<LPR>(sq) / (float)D + 1e-5f);
        if (row < rows) {
            TO* orow = out + (size_t)row * D;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * LPR + l) * 4;
                if (c < D) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (cur[i][e] - mean) * rstd * g[i][e] + b[i][e];
                    store4<TO>(orow + c, o);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[i][e] = nxt[i][e];
    }
}
template <typename TI, typename TO>
static void launch_layernorm(const TI* x, TO* out, const float* gamma, const float* beta, int rows, int D, hipStream_t s) {
    const int grid = std::min(cdiv(rows, 4), 256 * 8);  // 8 workgroups of 4 waves per CU, each wave walking rows
    if (D <= 128) hipLaunchKernelGGL((layernorm_kernel<TI, TO, 32, 1>), dim3(std::min(cdiv(rows, 8), 256 * 8)), dim3(256), 0, s, x, out, gamma, beta, rows, D);
    else if (D <= 256) hipLaunchKernelGGL((layernorm_kernel<TI, TO, 64, 1>), dim3(grid), dim3(256), 0, s, x, out, gamma, beta, rows, D);
    else if (D <= 512) hipLaunchKernelGGL((layernorm_kernel<TI, TO, 64, 2>), dim3(grid), dim3(256), 0, s, x, out, gamma, beta, rows, D);
    else hipLaunchKernelGGL((layernorm_kernel<TI, TO, 64, 4>), dim3(grid), dim3(256), 0, s, x, out, gamma, beta, rows, D);
}
template <typename T>
static void layernorm_dt(const void* x, int x_f32, void* out, int out_f32, const float* gamma, const float* beta, int rows, int D, hipStream_t s) {
    if (x_f32 && out_f32) launch_layernorm<float, float>((const float*)x, (float*)out, gamma, beta, rows, D, s);
    else if (x_f32) launch_layernorm<float, T>((const float*)x, (T*)out, gamma, beta, rows, D, s);
    else if (out_f32) launch_layernorm<T, float>((const T*)x, (float*)out, gamma, beta, rows, D, s);
    else launch_layernorm<T, T>((const T*)x, (T*)out, gamma, beta, rows, D, s);
}
int k_layernorm(int dtype, const void* x, int x_f32, void* out, int out_f32, const float* gamma, const float* beta, int rows, int D,
                hipStream_t s) {
    OCRVI_CHECK(x && out && gamma && beta && rows > 0 && D % 4 == 0 && D <= 1024, OCRVI_EINVAL, "layernorm: bad shape rows=%d D=%d", rows, D);
    ProfScope ps_("layernorm", 0.0, (double)rows*D*((x_f32?4:dtype_size(dtype))+(out_f32?4:dtype_size(dtype))), s);
    DISPATCH_DT(dtype, (layernorm_dt<T>(x, x_f32, out, out_f32, gamma, beta, rows, D, s)));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// ------------------------------------------------------------------ casts / layout taps
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float f[4];
        load4<float>(x + i * 4, f);
        store4<T>(y + i * 4, f);
    }
}
int k_cast_from_f32(int dtype, const float* x, void* y, size_t n, hipStream_t s) {
    OCRVI_CHECK(x && y && n % 4 == 0, OCRVI_EINVAL, "cast: n %% 4 != 0");
    const int grid = (int)std::min<size_t>((n / 4 + 255) / 256, 8192);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<T>, dim3(std::max(grid, 1)), dim3(256), 0, s, x, (T*)y, n / 4));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T>
__global__ void nhwc_to_nchw_f32_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int HW, int C, int ld, int coff) {
    // 32x32 tile transpose through LDS so both sides are coalesced
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? load_elem<T>(x, ((size_t)n * HW + p) * ld + coff + c) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) y[((size_t)n * C + c) * HW + p] = tile[tx][r];
    }
}
int k_nhwc_to_nchw_f32(int dtype, const void* x, float* y, int N, int H, int W, int C, int ld, int coff, hipStream_t s) {
    OCRVI_CHECK(x && y && N > 0 && N < 65536, OCRVI_EINVAL, "nhwc->nchw: bad shape");
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<T>, grid, dim3(256), 0, s, (const T*)x, y, N, HW, C, ld, coff));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// ------------------------------------------------------------------ FRM vertical cross-attention (H keys per column, head_dim 32)
// One half-wave (32 lanes = head_dim) per (column, head): lane d holds q[d], k_h[d], v_h[d].
template <typename T>
__global__ void frm_vertical_kernel(const T* __restrict__ kv, const float* __restrict__ vq, T* __restrict__ out, int B, int H, int W, int D) {
    const int heads = D >> 5;
    const size_t total = (size_t)B * W * heads;
    const int d = threadIdx.x & 31;
    const size_t item = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (item >= total) return;  // whole 32-lane group exits together
    const int hd = (int)(item % heads);
    const size_t col = item / heads;  // b*W + w
    const int b = (int)(col / W), w = (int)(col % W);
    const float q = vq[hd * 32 + d];
    const float scale = 0.17677669529663687f;  // 32^-0.5
    float mx = -INFINITY, sc[8], vv[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        if (h >= H) break;
        const size_t tok = ((size_t)b * H + h) * W + w;
        const float kk = load_elem<T>(kv, tok * 2 * D + hd * 32 + d);
        vv[h] = load_elem<T>(kv, tok * 2 * D + D + hd * 32 + d);
        float s = q * kk;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
        sc[h] = s * scale;
        mx = fmaxf(mx, sc[h]);
    }
    float den = 0.f, acc = 0.f;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        if (h >= H) break;
        const float p = __expf(sc[h] - mx);
        den += p;
        acc += p * vv[h];
    }
    store_elem<T>(out, col * D + hd * 32 + d, acc / den);
}
int k_frm_vertical(int dtype, const void* kv, const float* vq, void* out, int B, int H, int W, int D, hipStream_t s) {
    OCRVI_CHECK(kv && vq && out && D % 32 == 0 && H >= 1 && H <= 8, OCRVI_EINVAL, "frm vertical: D=%d H=%d unsupported", D, H);
    const size_t threads = (size_t)B * W * (D / 32) * 32;
    DISPATCH_DT(dtype, hipLaunchKernelGGL(frm_vertical_kernel<T>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const T*)kv, vq, (T*)out, B, H, W, D));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// ------------------------------------------------------------------ ASF (adaptive scale fusion, neck.py:57-79)
// score = conv1x1(concat_l up(p_l)) = sum_l up(W_l . p_l): bilinear upsampling is linear and acts per channel, so it commutes
// with the 1x1 attention conv.  (1) asf_scores_kernel: 4-channel fp32 score maps s_l = W_l . p_l for the three coarse levels at
// their NATIVE resolution (1/4 + 1/16 + 1/64 of the pixels).  (2) asf_blend_kernel at p2 resolution: 16 lanes share a pixel
// (16 channels = two 16-byte chunks per lane, 4 pixels per wave): own-level score by a 16-lane butterfly, coarse scores by
// bilinear taps of the tiny maps, softmax over 4, then out = a0*p2 + sum_{l,tap} (a_l*w_tap) * p_l[tap].
template <typename T>
__global__ __launch_bounds__(256) void asf_scores_kernel(const T* __restrict__ p, const float* __restrict__ w, float* __restrict__ sc, int level,
                                                         size_t npix) {
    const int lane = threadIdx.x & 63;
    float wr[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = *(const float4*)(w + i * 1024 + level * 256 + lane * 4);
        wr[i][0] = t.x; wr[i][1] = t.y; wr[i][2] = t.z; wr[i][3] = t.w;
    }
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    for (size_t pix = wave; pix < npix; pix += nwaves) {
        float f[4];
        load4<T>(p + pix * 256 + lane * 4, f);
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = wave_sum(f[0] * wr[i][0] + f[1] * wr[i][1] + f[2] * wr[i][2] + f[3] * wr[i][3]);
        if (lane == 0) *(float4*)(sc + pix * 4) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// waves_per_eu(2, 2): left to itself hipcc allocates ~120 VGPRs for the fp16 / fp32 builds to reach 4 waves per SIMD and pays for it by
// consuming the 25 loads of a pixel in small batches; this kernel lives on loads in flight per wave, not on waves (fp16 1374 -> 830 us,
// fp32 1710 -> 1486 us for 16 pages; the bf16 build needs 239 VGPRs either way).
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void asf_blend_kernel(const T* __restrict__ p2, const T* __restrict__ p3, const T* __restrict__ p4,
                                                        const T* __restrict__ p5, const float* __restrict__ s3, const float* __restrict__ s4,
                                                        const float* __restrict__ s5, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, int N, int H, int W) {
    constexpr int EPC = TypeInfo<T>::EPC;   // channels per lane = one 16-byte chunk
    constexpr int LPP = 256 / EPC;          // lanes per pixel (32 for bf16/fp16, 64 for fp32)
    constexpr int PPW = 64 / LPP;           // pixels per wave pass
    const int lane = threadIdx.x & 63, j = lane % LPP, sub = lane / LPP;
    float w0[4][EPC];                       // level-0 attention weights of this lane's channels
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
            const float4 t = *(const float4*)(w + i * 1024 + j * EPC + e);
            w0[i][e] = t.x; w0[i][e + 1] = t.y; w0[i][e + 2] = t.z; w0[i][e + 3] = t.w;
        }
    const float4 bv = *(const float4*)bias;
    const T* lev[3] = {p3, p4, p5};
    const float* slev[3] = {s3, s4, s5};
    // A workgroup sweeps compact 8x8 pixel tiles (one tile row of 8 pixels per pass: 4 waves x PPW pixels x 8/(4*PPW) passes) so the
    // coarse-level taps (a 5x5 / 3x3 / 2x2 neighbourhood per tile) are re-used from L1 instead of being fetched from L2 per pixel.
    const int tiles_x = W >> 3, tiles_y = H >> 3;
    const int ntile = N * tiles_y * tiles_x;
    const int wv = threadIdx.x >> 6;
    constexpr int XPP = 4 * PPW;       // pixels of a tile row handled per pass
    float shl[3], swl[3];  // align_corners=True scales (in-1)/(out-1) per coarse level
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        shl[l] = H > 1 ? (float)((H >> (l + 1)) - 1) / (float)(H - 1) : 0.f;
        swl[l] = W > 1 ? (float)((W >> (l + 1)) - 1) / (float)(W - 1) : 0.f;
    }
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int tx = tile % tiles_x, tt = tile / tiles_x, ty = tt % tiles_y, n = tt / tiles_y;
    for (int it = 0; it < 8 * (8 / XPP); ++it) {
        const int y = ty * 8 + it / (8 / XPP);
        const int x = tx * 8 + (it % (8 / XPP)) * XPP + wv * PPW + sub;
        const size_t pix = ((size_t)n * H + y) * W + x;
        const bool ok = true;
        const size_t pp = pix;
        float f0[EPC];
        Chunk<T>::unpack(*(const uint4*)(p2 + pp * 256 + j * EPC), f0);
        float sc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) a = fmaf(w0[i][e], f0[e], a);
#pragma unroll
            for (int o = 1; o < LPP; o <<= 1) a += __shfl_xor(a, o);
            sc[i] = a;
        }
        sc[0] += bv.x; sc[1] += bv.y; sc[2] += bv.z; sc[3] += bv.w;
        // coarse levels: tap geometry (F.interpolate bilinear, align_corners=True: src = dst * (in-1)/(out-1), neck.py:65)
        int toff[3][4];
        float tw[3][4];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int Hl = H >> (l + 1), Wl = W >> (l + 1);
            const float fy = shl[l] * (float)y, fx = swl[l] * (float)x;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < Hl - 1 ? 1 : 0), x1 = x0 + (x0 < Wl - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            const int b0 = n * Hl * Wl;
            toff[l][0] = b0 + y0 * Wl + x0; toff[l][1] = b0 + y0 * Wl + x1;
            toff[l][2] = b0 + y1 * Wl + x0; toff[l][3] = b0 + y1 * Wl + x1;
            tw[l][0] = hy * hx; tw[l][1] = hy * lx; tw[l][2] = ly * hx; tw[l][3] = ly * lx;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 sv = *(const float4*)(slev[l] + (size_t)toff[l][q] * 4);
                sc[0] = fmaf(tw[l][q], sv.x, sc[0]); sc[1] = fmaf(tw[l][q], sv.y, sc[1]);
                sc[2] = fmaf(tw[l][q], sv.z, sc[2]); sc[3] = fmaf(tw[l][q], sv.w, sc[3]);
            }
        }
        const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = __expf(sc[i] - mx); den += sc[i]; }
        const float inv = 1.f / den;
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = f0[e] * (sc[0] * inv);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const float al = sc[l + 1] * inv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float cf = al * tw[l][q];
                float f[EPC];
                Chunk<T>::unpack(*(const uint4*)(lev[l] + (size_t)toff[l][q] * 256 + j * EPC), f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = fmaf(cf, f[e], o[e]);
            }
        }
        if (ok) *(uint4*)(out + pix * 256 + j * EPC) = Chunk<T>::pack(o);
    }
    }
}

// 4-byte element types (fp32, f16x2): all 64 lanes of a wave share ONE pixel (64 x 4 channels), so a pixel's twelve coarse taps are
// wave-uniform.  The per-pixel form above, run that way, is bound by LATENCY: one pixel in flight per wave, 13 dependent loads each
// (6300 cycles per pixel and wave, 1583 us for 16 pages against ~500 us of HBM time).  Here a wave walks DOWN one column of ASF_TALL
// pixels: the taps' columns never change and their rows advance by at most one per pixel, so every level keeps three tap rows in
// registers -- (top, bottom) in use and the row after them already requested -- and p2 runs four pixels ahead; in the steady state no
// load is waited for.  The four waves of a workgroup take four adjacent columns (their coarse taps overlap in L1).  The arithmetic is
// asf_blend_kernel's except for the order of the 64-lane score sums (DPP row sums + four row totals instead of a butterfly).
constexpr int ASF_TALL = 48;
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void asf_blend_col_kernel(const T* __restrict__ p2, const T* __restrict__ p3, const T* __restrict__ p4,
                                                        const T* __restrict__ p5, const float* __restrict__ s3, const float* __restrict__ s4,
                                                        const float* __restrict__ s5, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, int N, int H, int W) {
    static_assert(TypeInfo<T>::EPC == 4, "one pixel per wave");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float w0[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = *(const float4*)(w + i * 1024 + lane * 4);
        w0[i][0] = t.x; w0[i][1] = t.y; w0[i][2] = t.z; w0[i][3] = t.w;
    }
    const float4 bv = *(const float4*)bias;
    const T* lev[3] = {p3, p4, p5};
    const float* slev[3] = {s3, s4, s5};
    const int strips = W >> 2, segs = (H + ASF_TALL - 1) / ASF_TALL;
    const int ntask = N * segs * strips;
    float shl[3], swl[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        shl[l] = H > 1 ? (float)((H >> (l + 1)) - 1) / (float)(H - 1) : 0.f;
        swl[l] = W > 1 ? (float)((W >> (l + 1)) - 1) / (float)(W - 1) : 0.f;
    }
    for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
        const int sx = task % strips, tt = task / strips, sg = tt % segs, n = tt / segs;
        const int x = sx * 4 + wv;
        const int ya = sg * ASF_TALL, yb = min(ya + ASF_TALL, H);      // rows [ya, yb) of column x
        // ---- per level: the taps' column pair and its weights, and three tap rows: slots 0,1 = (top, x0 / x1), 2,3 = bottom, 4,5 = next
        int cx0[3], cx1[3], cy0[3], base[3];
        float clx[3];
        uint4 tap[3][6];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int Hl = H >> (l + 1), Wl = W >> (l + 1);
            const float fx = swl[l] * (float)x;
            const int x0 = (int)fx;
            cx0[l] = x0;
            cx1[l] = x0 + (x0 < Wl - 1 ? 1 : 0);
            clx[l] = fx - (float)x0;
            base[l] = n * Hl * Wl;
            const int y0 = __builtin_amdgcn_readfirstlane((int)(shl[l] * (float)ya));
            cy0[l] = y0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int row = min(y0 + r, Hl - 1);
                tap[l][2 * r] = *(const uint4*)(lev[l] + (size_t)(base[l] + row * Wl + cx0[l]) * 256 + lane * 4);
                tap[l][2 * r + 1] = *(const uint4*)(lev[l] + (size_t)(base[l] + row * Wl + cx1[l]) * 256 + lane * 4);
            }
        }
        const T* const pcol = p2 + (((size_t)n * H + ya) * W + x) * 256 + lane * 4;
        T* const ocol = out + (((size_t)n * H + ya) * W + x) * 256 + lane * 4;
        const size_t rowstep = (size_t)W * 256;
        const int rows = yb - ya;
        uint4 ring[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ring[k] = *(const uint4*)(pcol + (size_t)min(k, rows - 1) * rowstep);
        auto pixel = [&](int yy, uint4& slot) {
            const int y = ya + yy;
            float f0[4];
            Chunk<T>::unpack(slot, f0);
            slot = *(const uint4*)(pcol + (size_t)min(yy + 4, rows - 1) * rowstep);     // four pixels ahead
            // tap rows of this pixel; the coarse score vectors are wave-uniform loads (scalar), requested before the reductions below
            float tw[3][4];
            float4 sv[3][4];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const int Hl = H >> (l + 1), Wl = W >> (l + 1);
                const float fy = shl[l] * (float)y;
                const int y0 = __builtin_amdgcn_readfirstlane((int)fy);
                if (y0 != cy0[l]) {      // wave-uniform: the rows advance by one; request the row after the new bottom row
                    tap[l][0] = tap[l][2]; tap[l][1] = tap[l][3];
                    tap[l][2] = tap[l][4]; tap[l][3] = tap[l][5];
                    const int row = min(y0 + 2, Hl - 1);
                    tap[l][4] = *(const uint4*)(lev[l] + (size_t)(base[l] + row * Wl + cx0[l]) * 256 + lane * 4);
                    tap[l][5] = *(const uint4*)(lev[l] + (size_t)(base[l] + row * Wl + cx1[l]) * 256 + lane * 4);
                    cy0[l] = y0;
                }
                const int y1 = y0 + (y0 < Hl - 1 ? 1 : 0);
                const float ly = fy - (float)y0, lx = clx[l], hy = 1.f - ly, hx = 1.f - lx;
                tw[l][0] = hy * hx; tw[l][1] = hy * lx; tw[l][2] = ly * hx; tw[l][3] = ly * lx;
                const int o00 = __builtin_amdgcn_readfirstlane(base[l] + y0 * Wl + cx0[l]), o01 = __builtin_amdgcn_readfirstlane(base[l] + y0 * Wl + cx1[l]);
                const int o10 = __builtin_amdgcn_readfirstlane(base[l] + y1 * Wl + cx0[l]), o11 = __builtin_amdgcn_readfirstlane(base[l] + y1 * Wl + cx1[l]);
                sv[l][0] = *(const float4*)(slev[l] + (size_t)o00 * 4); sv[l][1] = *(const float4*)(slev[l] + (size_t)o01 * 4);
                sv[l][2] = *(const float4*)(slev[l] + (size_t)o10 * 4); sv[l][3] = *(const float4*)(slev[l] + (size_t)o11 * 4);
            }
            float sc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) a = fmaf(w0[i][e], f0[e], a);
                sc[i] = wave_sum_uniform(a);
            }
            sc[0] += bv.x; sc[1] += bv.y; sc[2] += bv.z; sc[3] += bv.w;
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sc[0] = fmaf(tw[l][q], sv[l][q].x, sc[0]); sc[1] = fmaf(tw[l][q], sv[l][q].y, sc[1]);
                    sc[2] = fmaf(tw[l][q], sv[l][q].z, sc[2]); sc[3] = fmaf(tw[l][q], sv[l][q].w, sc[3]);
                }
            const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { sc[i] = __expf(sc[i] - mx); den += sc[i]; }
            const float inv = 1.f / den;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f0[e] * (sc[0] * inv);
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float al = sc[l + 1] * inv;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float cf = al * tw[l][q];
                    if constexpr (IsSplit<T>::value) {
                        Chunk<T>::fma(tap[l][q], cf, o);      // both halves of every channel in mixed-precision fmas, no conversions
                    } else {
                        float f[4];
                        Chunk<T>::unpack(tap[l][q], f);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaf(cf, f[e], o[e]);
                    }
                }
            }
            *(uint4*)(ocol + (size_t)yy * rowstep) = Chunk<T>::pack(o);
        };
#pragma unroll 1
        for (int yy = 0; yy < rows; yy += 4) {      // unrolled by four so that the p2 ring has static slots
            pixel(yy, ring[0]);
            if (yy + 1 < rows) pixel(yy + 1, ring[1]);
            if (yy + 2 < rows) pixel(yy + 2, ring[2]);
            if (yy + 3 < rows) pixel(yy + 3, ring[3]);
        }
    }
}

template <typename T>
static void asf_launch(const void* p2, const void* p3, const void* p4, const void* p5, const float* w, const float* b, float* s3, float* s4,
                       float* s5, void* out, int N, int H, int W, hipStream_t s) {
    const void* lv[3] = {p3, p4, p5};
    float* sv[3] = {s3, s4, s5};
    for (int l = 0; l < 3; ++l) {
        const size_t np = (size_t)N * (H >> (l + 1)) * (W >> (l + 1));
        const int grid = (int)std::min<size_t>((np + 3) / 4, 2048);
        hipLaunchKernelGGL(asf_scores_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)lv[l], w, sv[l], l + 1, np);
    }
    const size_t total = (size_t)N * H * W;
    const int grid = (int)std::min<size_t>(total / 64, 256 * 8);
    if constexpr (TypeInfo<T>::EPC == 4) {
        const int ntask = N * ((H + ASF_TALL - 1) / ASF_TALL) * (W / 4);       // (column strip of four, ASF_TALL rows) per workgroup pass
        hipLaunchKernelGGL(asf_blend_col_kernel<T>, dim3(std::min(ntask, 512)), dim3(256), 0, s, (const T*)p2, (const T*)p3, (const T*)p4, (const T*)p5, s3, s4, s5, w, b,
                           (T*)out, N, H, W);
    } else {
        hipLaunchKernelGGL(asf_blend_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)p2, (const T*)p3, (const T*)p4, (const T*)p5, s3, s4, s5, w, b,
                           (T*)out, N, H, W);
    }
}

int k_asf(int dtype, const void* p2, const void* p3, const void* p4, const void* p5, const float* w, const float* b, float* scratch, void* out,
          int N, int H, int W, hipStream_t s) {
    OCRVI_CHECK(p2 && p3 && p4 && p5 && w && b && out && scratch && H % 8 == 0 && W % 8 == 0, OCRVI_EINVAL, "asf: bad shape %dx%d", H, W);
    const size_t n3 = (size_t)N * (H / 2) * (W / 2), n4 = (size_t)N * (H / 4) * (W / 4);
    float *s3 = scratch, *s4 = s3 + n3 * 4, *s5 = s4 + n4 * 4;
    ProfScope ps_("asf_fused", 0.0, (double)N * H * W * 256 * dtype_size(dtype) * (2.0 + 1.0 / 4 + 1.0 / 16 + 1.0 / 64), s);
    DISPATCH_DT(dtype, asf_launch<T>(p2, p3, p4, p5, w, b, s3, s4, s5, out, N, H, W, s));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}
// bytes of fp32 scratch k_asf needs for the coarse score maps
size_t asf_scratch_bytes(int N, int H, int W) { return ((size_t)N * (H / 2) * (W / 2) + (size_t)N * (H / 4) * (W / 4) + (size_t)N * (H / 8) * (W / 8)) * 16; }

// ------------------------------------------------------------------ DB head tail: deconv2 (64->1, 2x2 s2) x2 branches + sigmoid + step
// 128/EPC lanes share one pixel's 128-channel row (one 16-byte chunk per lane -> the wave reads whole contiguous rows):
// lanes of the first half own the binarise branch (ch 0..63), of the second the threshold branch.  Each lane keeps the
// deconv weights of its own channels in registers, partial sums are butterflied inside the branch, and lanes 0..3 of a
// pixel write the four 2x2 output positions.
template <typename T>
__global__ __launch_bounds__(256) void db_tail_kernel(const T* __restrict__ y, const float* __restrict__ w2, const float* __restrict__ b2, float k,
                                                      float* __restrict__ binary, float* __restrict__ thresh, float* __restrict__ tbin,
                                                      float* __restrict__ blog, float* __restrict__ tlog, int N, int H2, int W2) {
    constexpr int EPC = TypeInfo<T>::EPC;
    constexpr int LPP = 128 / EPC;      // lanes per pixel (16 or 32)
    constexpr int PPW = 64 / LPP;       // pixels per wave pass
    const int lane = threadIdx.x & 63;
    const int j = lane % LPP, sub = lane / LPP;
    const int br = j >= LPP / 2 ? 1 : 0;
    float wr[EPC][4];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float4 t = *(const float4*)(w2 + (size_t)(j * EPC + e) * 4);  // w2 is [2][64][4] = [128][4] in channel order
        wr[e][0] = t.x; wr[e][1] = t.y; wr[e][2] = t.z; wr[e][3] = t.w;
    }
    const float bias = b2[br];
    const size_t total = (size_t)N * H2 * W2;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    const int OW = 2 * W2;
    for (size_t p0 = wave * PPW; p0 < total; p0 += nwaves * PPW) {
        const size_t pix = p0 + sub;
        const bool ok = pix < total;
        float f[EPC];
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (ok) raw = *(const uint4*)(y + pix * 128 + j * EPC);
        Chunk<T>::unpack(raw, f);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = fmaf(f[e], wr[e][q], a[q]);
        }
#pragma unroll
        for (int o = 1; o < LPP / 2; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += __shfl_xor(a[q], o);
        }
        // every lane of a branch now holds that branch's 4 logits; fetch the other branch's from the partner half
        float mine[4], other[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mine[q] = a[q] + bias;
            other[q] = __shfl_xor(mine[q], LPP / 2);
        }
        if (ok && j < 4) {  // lane j of the binarise half writes output position q = j
            const float lb = j == 0 ? mine[0] : (j == 1 ? mine[1] : (j == 2 ? mine[2] : mine[3]));
            const float lt = j == 0 ? other[0] : (j == 1 ? other[1] : (j == 2 ? other[2] : other[3]));
            const int x = (int)(pix % W2);
            const size_t t = pix / W2;
            const int yy = (int)(t % H2), n = (int)(t / H2);
            const size_t o = ((size_t)n * (2 * H2) + 2 * yy + (j >> 1)) * OW + 2 * x + (j & 1);
            const float pb = 1.f / (1.f + expf(-lb)), pt = 1.f / (1.f + expf(-lt));
            binary[o] = pb;
            if (thresh) thresh[o] = pt;
            if (tbin) tbin[o] = 1.f / (1.f + expf(-k * (pb - pt)));
            if (blog) blog[o] = lb;
            if (tlog) tlog[o] = lt;
        }
    }
}
int k_db_tail(int dtype, const void* y, const float* w2, const float* b2, float k, float* binary, float* thresh, float* thresh_binary,
              float* bin_logits, float* thresh_logits, int N, int H2, int W2, hipStream_t s) {
    OCRVI_CHECK(y && w2 && b2 && binary && N > 0, OCRVI_EINVAL, "db tail: null operand");
    ProfScope ps_("db_head_tail", 0.0, (double)N*H2*W2*(128.0*dtype_size(dtype)+16.0*((thresh?1:0)+(thresh_binary?1:0)+(bin_logits?1:0)+(thresh_logits?1:0)+1)), s);
    const size_t total = (size_t)N * H2 * W2;
    const int grid = (int)std::min<size_t>((total + 15) / 16, 256 * 8);
    DISPATCH_DT(dtype, hipLaunchKernelGGL(db_tail_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)y, w2, b2, k, binary, thresh, thresh_binary, bin_logits, thresh_logits, N, H2, W2));
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// logits -> binary = sigmoid(bin), thresh = sigmoid(thr), thresh_binary = 1/(1+exp(-k(binary-thresh)))  (head.py:28-40)
__global__ void db_maps_kernel(const float* __restrict__ bl, const float* __restrict__ tl, float k, float* __restrict__ binary,
                               float* __restrict__ thresh, float* __restrict__ tbin, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 b = ((const float4*)bl)[i], t = ((const float4*)tl)[i];
        float4 pb, pt, st;
        pb.x = 1.f / (1.f + expf(-b.x)); pb.y = 1.f / (1.f + expf(-b.y)); pb.z = 1.f / (1.f + expf(-b.z)); pb.w = 1.f / (1.f + expf(-b.w));
        pt.x = 1.f / (1.f + expf(-t.x)); pt.y = 1.f / (1.f + expf(-t.y)); pt.z = 1.f / (1.f + expf(-t.z)); pt.w = 1.f / (1.f + expf(-t.w));
        ((float4*)binary)[i] = pb;
        if (thresh) ((float4*)thresh)[i] = pt;
        if (tbin) {
            st.x = 1.f / (1.f + expf(-k * (pb.x - pt.x))); st.y = 1.f / (1.f + expf(-k * (pb.y - pt.y)));
            st.z = 1.f / (1.f + expf(-k * (pb.z - pt.z))); st.w = 1.f / (1.f + expf(-k * (pb.w - pt.w)));
            ((float4*)tbin)[i] = st;
        }
    }
}
int k_db_maps(const float* bin_logits, const float* thresh_logits, float k, float* binary, float* thresh, float* thresh_binary, size_t n,
              hipStream_t s) {
    OCRVI_CHECK(bin_logits && thresh_logits && binary && n % 4 == 0, OCRVI_EINVAL, "db maps: bad argument");
    ProfScope ps_("db_maps", 0.0, (double)n * 4 * (3 + (thresh ? 1 : 0) + (thresh_binary ? 1 : 0)), s);
    const int grid = (int)std::min<size_t>((n / 4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(db_maps_kernel, dim3(grid), dim3(256), 0, s, bin_logits, thresh_logits, k, binary, thresh, thresh_binary, n / 4);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// ------------------------------------------------------------------ CTC: log-softmax + argmax (one wave per (b,t) row), collapse
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

__global__ void ctc_logsoftmax_argmax_kernel(const float* __restrict__ logits, int ld, float* __restrict__ log_probs,
                                             int32_t* __restrict__ argmax_ids, int B, int T, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // b*T + t
    if (row >= B * T) return;
    const int b = row / T, t = row % T;
    const float* x = logits + (size_t)row * ld;
    constexpr int MAXV = 16;  // C <= 1024
    float v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < C ? x[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i * 64 + lane < C) sum += expf(v[i] - mx);
    const float lse = logf(wave_sum(sum));
    // argmax over the log-probabilities themselves (svtrv2.py:555 takes argmax of log_softmax output); first index wins ties
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < C) {
            const float lp = (v[i] - mx) - lse;
            if (log_probs) log_probs[((size_t)t * B + b) * C + c] = lp;
            if (lp > best) { best = lp; bi = c; }
        }
    }
    wave_argmax(best, bi);
    if (argmax_ids && lane == 0) argmax_ids[(size_t)b * T + t] = bi;
}
int k_ctc_logsoftmax_argmax(const float* logits, int ld, float* log_probs, int32_t* argmax_ids, int B, int T, int C, hipStream_t s) {
    OCRVI_CHECK(logits && B > 0 && T > 0 && C > 0 && C <= 1024 && ld >= C, OCRVI_EINVAL, "ctc: bad shape B=%d T=%d C=%d", B, T, C);
    ProfScope ps_("ctc_logsoftmax_argmax", 0.0, (double)B*T*C*(log_probs?8.0:4.0), s);
    hipLaunchKernelGGL(ctc_logsoftmax_argmax_kernel, dim3(cdiv(B * T, 4)), dim3(256), 0, s, logits, ld, log_probs, argmax_ids, B, T, C);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

__global__ void ctc_argmax_tbc_kernel(const float* __restrict__ lp, int32_t* __restrict__ argmax_ids, int B, int T, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // t*B + b
    if (row >= B * T) return;
    const int t = row / B, b = row % B;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    bool any_nan = false;
    for (int c = lane; c < C; c += 64) {
        const float v = lp[(size_t)row * C + c];
        any_nan |= (v != v);
        if (v > best) { best = v; bi = c; }
    }
    if (bi == 0x7fffffff && !any_nan) bi = lane < C ? lane : 0x7fffffff;  // all -inf: first index
    wave_argmax(best, bi);
    if (lane == 0) argmax_ids[(size_t)b * T + t] = bi == 0x7fffffff ? 0 : bi;
}
int k_ctc_argmax_tbc(const float* log_probs, int32_t* argmax_ids, int B, int T, int C, hipStream_t s) {
    OCRVI_CHECK(log_probs && argmax_ids && B > 0 && T > 0 && C > 0, OCRVI_EINVAL, "ctc argmax: bad shape");
    hipLaunchKernelGGL(ctc_argmax_tbc_kernel, dim3(cdiv(B * T, 4)), dim3(256), 0, s, log_probs, argmax_ids, B, T, C);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// One wave per sequence; 64 time steps per pass, keep[t] = id != blank && id != previous id; compaction by ballot prefix count.
__global__ void ctc_collapse_kernel(const int32_t* __restrict__ am, int32_t* __restrict__ ids, int32_t* __restrict__ lens, int B, int T, int blank) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int32_t* a = am + (size_t)b * T;
    int32_t* o = ids ? ids + (size_t)b * T : nullptr;
    int count = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int cur = t < T ? a[t] : blank;
        const int prev = (t > 0 && t < T) ? a[t - 1] : -1;
        const bool keep = t < T && cur != blank && cur != prev;
        const unsigned long long mask = __ballot(keep);
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep && o) o[pos] = cur;
        count += __popcll(mask);
    }
    if (o)
        for (int t = count + lane; t < T; t += 64) o[t] = -1;
    if (lens && lane == 0) lens[b] = count;
}
int k_ctc_collapse(const int32_t* argmax_ids, int32_t* ids, int32_t* lens, int B, int T, int blank, hipStream_t s) {
    OCRVI_CHECK(argmax_ids && B > 0 && T > 0, OCRVI_EINVAL, "ctc collapse: bad shape");
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, argmax_ids, ids, lens, B, T, blank);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

}  // namespace ocrvi
