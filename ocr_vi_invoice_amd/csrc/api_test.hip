// Kernel-level hooks of the C ABI (include/ocrvi.h, "test/bench hooks"): run ONE production kernel on
// caller-supplied float32 NCHW device tensors.  They allocate scratch, convert layouts, time with HIP events on
// their own stream, and synchronise -- test infrastructure around the same kernels the model graphs launch.
#include <memory>

#include "model.h"

using namespace ocrvi;
namespace ocrvi {
OCRVI_RANGE_FLAG_TU()   // binds this unit's f16x2 range-flag pointer (common.h)
}

namespace {
template <typename T>
__global__ void nchw_f32_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW) {
    const size_t total = (size_t)N * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t t = i / C;
        const int p = (int)(t % HW), n = (int)(t / HW);
        store_elem<T>(y, i, x[((size_t)n * C + c) * HW + p]);
    }
}
int to_nhwc(int dt, const float* x, void* y, int N, int C, int HW, hipStream_t s) {
    const size_t total = (size_t)N * C * HW;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    switch (dt) {
        case OCRVI_F32: hipLaunchKernelGGL(nchw_f32_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, s, x, (float*)y, N, C, HW); break;
        case OCRVI_BF16: hipLaunchKernelGGL(nchw_f32_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, (bf16_t*)y, N, C, HW); break;
        case OCRVI_F16: hipLaunchKernelGGL(nchw_f32_to_nhwc_kernel<f16_t>, dim3(grid), dim3(256), 0, s, x, (f16_t*)y, N, C, HW); break;
        case OCRVI_F16X2: hipLaunchKernelGGL(nchw_f32_to_nhwc_kernel<f16x2_t>, dim3(grid), dim3(256), 0, s, x, (f16x2_t*)y, N, C, HW); break;
        default: set_error("unknown dtype %d", dt); return OCRVI_EINVAL;
    }
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}
// [N][18][P] offsets + [N][9][P] masks -> [N*P][32] rows (the layout the offset conv's epilogue writes)
__global__ void pack_offsets_kernel(const float* __restrict__ off, const float* __restrict__ mask, float* __restrict__ o, int N, int P) {
    const size_t total = (size_t)N * P * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31);
        const size_t m = i >> 5;
        const int p = (int)(m % P), n = (int)(m / P);
        float v = 0.f;
        if (c < 18) v = off[((size_t)n * 18 + c) * P + p];
        else if (c < 27) v = mask[((size_t)n * 9 + (c - 18)) * P + p];
        o[i] = v;
    }
}
struct Scratch {
    std::vector<void*> p;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
        for (void* q : p) (void)hipFree(q);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
    }
    int alloc(size_t bytes, void** out) {
        OCRVI_HIP(hipMalloc(out, bytes ? bytes : 16));
        p.push_back(*out);
        return OCRVI_OK;
    }
    int init() {
        unsigned* flag = nullptr;
        OCRVI_TRY(range_flag_bind(&flag));   // the hooks run f16x2 kernels too: ocrvi_range_flag / ocrvi_range_reset see them
        OCRVI_HIP(hipStreamCreate(&s));
        OCRVI_HIP(hipEventCreate(&e0));
        OCRVI_HIP(hipEventCreate(&e1));
        return OCRVI_OK;
    }
};
template <typename F>
int timed(Scratch& sc, int iters, float* avg_ms, F&& launch) {
    OCRVI_TRY(launch());  // warm-up / the run whose output is checked
    if (iters > 0 && avg_ms) {
        OCRVI_HIP(hipEventRecord(sc.e0, sc.s));
        for (int i = 0; i < iters; ++i) OCRVI_TRY(launch());
        OCRVI_HIP(hipEventRecord(sc.e1, sc.s));
        OCRVI_HIP(hipEventSynchronize(sc.e1));
        float ms = 0.f;
        OCRVI_HIP(hipEventElapsedTime(&ms, sc.e0, sc.e1));
        *avg_ms = ms / iters;
    }
    return OCRVI_OK;
}
}  // namespace

extern "C" int ocrvi_test_deform_conv(int device, int dtype, const float* x, const float* offset, const float* mask,
                                      const float* weight_host, const float* bias_host, int N, int C, int H, int W, int Co, int stride,
                                      int relu, float* out, int iters, float* avg_ms) {
    OCRVI_CHECK(x && offset && mask && weight_host && out && (stride == 1 || stride == 2), OCRVI_EINVAL, "test_deform_conv: bad argument");
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    DeviceStore st;
    ConvLayer L;
    PackedConv pc = pack_conv(weight_host, bias_host, Co, C, 3, 3, 1, AM_DCN, dtype);
    OCRVI_TRY(upload_packed(st, pc, AM_DCN, &L));
    void *xn = nullptr, *yn = nullptr, *offs = nullptr;
    OCRVI_TRY(sc.alloc((size_t)N * H * W * C * dtype_size(dtype), &xn));
    OCRVI_TRY(sc.alloc((size_t)N * Ho * Wo * Co * dtype_size(dtype), &yn));
    OCRVI_TRY(sc.alloc((size_t)N * Ho * Wo * 32 * 4, &offs));
    OCRVI_TRY(to_nhwc(dtype, x, xn, N, C, H * W, sc.s));
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1024), dim3(256), 0, sc.s, offset, mask, (float*)offs, N, Ho * Wo);
    OCRVI_HIP(hipGetLastError());
    Runner r(dtype, sc.s, (void*)256, 0);
    Tensor tx; tx.p = xn; tx.n = N; tx.h = H; tx.w = W; tx.c = C;
    Tensor ty; ty.p = yn; ty.n = N; ty.h = Ho; ty.w = Wo; ty.c = Co;
    ConvOpts o;
    o.sh = o.sw = stride; o.pad = 1; o.act = relu ? ACT_RELU : ACT_NONE; o.offs = (const float*)offs;
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() { return conv(r, L, tx, ty, o); }));
    OCRVI_TRY(k_nhwc_to_nchw_f32(dtype, yn, out, N, Ho, Wo, Co, Co, 0, sc.s));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}

extern "C" int ocrvi_test_stem_pool(int device, int dtype, const float* x, const float* weight_host, const float* bias_host, int N, int H, int W,
                                    int fused, float* out, int iters, float* avg_ms) {
    OCRVI_CHECK(x && weight_host && bias_host && out && N > 0 && H >= 8 && W >= 8 && H % 4 == 0 && W % 4 == 0, OCRVI_EINVAL, "test_stem_pool: bad argument");
    OCRVI_CHECK(!fused || dtype == OCRVI_F16X2, OCRVI_EINVAL, "test_stem_pool: the fused kernel is the f16x2 form");
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    DeviceStore st;
    ConvLayer L;
    PackedConv pc = pack_conv(weight_host, bias_host, 64, 3, 7, 7, 1, AM_ROWS, dtype);
    OCRVI_TRY(upload_packed(st, pc, AM_ROWS, &L));
    const int Hp = H + 6, Wp = W + 8;
    void *xp = nullptr, *sn = nullptr, *yn = nullptr;
    OCRVI_TRY(sc.alloc((size_t)N * Hp * Wp * 4 * dtype_size(dtype), &xp));
    OCRVI_TRY(sc.alloc((size_t)N * (H / 2) * (W / 2) * 64 * dtype_size(dtype), &sn));
    OCRVI_TRY(sc.alloc((size_t)N * (H / 4) * (W / 4) * 64 * dtype_size(dtype), &yn));
    OCRVI_TRY(k_nchw3_to_nhwc4_pad(dtype, x, xp, N, H, W, 3, 3, Hp, Wp, sc.s));
    Runner r(dtype, sc.s, (void*)256, 0);
    Tensor tx; tx.p = xp; tx.n = N; tx.h = Hp; tx.w = Wp; tx.c = 4;
    Tensor ts; ts.p = sn; ts.n = N; ts.h = H / 2; ts.w = W / 2; ts.c = 64;
    ConvOpts o;
    o.sh = o.sw = 2; o.pad = 3; o.act = ACT_RELU; o.Hp = Hp; o.Wp = Wp;
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() -> int {
        if (fused) return k_stem_pool(dtype, xp, L.w, L.bias, L.wscale, yn, N, H, W, Hp, Wp, sc.s);
        OCRVI_TRY(conv(r, L, tx, ts, o));
        return k_maxpool3x3s2(dtype, sn, yn, N, H / 2, W / 2, 64, sc.s);
    }));
    OCRVI_TRY(k_nhwc_to_nchw_f32(dtype, yn, out, N, H / 4, W / 4, 64, 64, 0, sc.s));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}

extern "C" int ocrvi_test_conv(int device, int dtype, const float* x, const float* weight_host, const float* bias_host, int N, int C,
                               int H, int W, int Co, int ksize, int sh, int sw, int groups, int act, float* out, int iters, float* avg_ms) {
    OCRVI_CHECK(x && weight_host && out && (ksize == 1 || ksize == 3) && groups >= 1 && C % groups == 0 && Co % groups == 0, OCRVI_EINVAL,
                "test_conv: bad argument");
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / sh + 1, Wo = (W + 2 * pad - ksize) / sw + 1;
    const int amode = ksize == 1 ? AM_CONV1 : AM_CONV3;
    DeviceStore st;
    ConvLayer L;
    PackedConv pc = pack_conv(weight_host, bias_host, Co, C / groups, ksize, ksize, groups, amode, dtype);
    OCRVI_TRY(upload_packed(st, pc, amode, &L));
    void *xn = nullptr, *yn = nullptr;
    // OCRVI_TEST_PADC=n (timing experiments only; the result is then meaningless): give the activation rows a channel stride of C + n
    const int padc = getenv("OCRVI_TEST_PADC") ? atoi(getenv("OCRVI_TEST_PADC")) : 0;
    OCRVI_TRY(sc.alloc((size_t)N * H * W * (C + padc) * dtype_size(dtype), &xn));
    OCRVI_TRY(sc.alloc((size_t)N * Ho * Wo * Co * dtype_size(dtype), &yn));
    OCRVI_TRY(to_nhwc(dtype, x, xn, N, C, H * W, sc.s));
    Runner r(dtype, sc.s, (void*)256, 0);
    Tensor tx; tx.p = xn; tx.n = N; tx.h = H; tx.w = W; tx.c = C + padc;
    Tensor ty; ty.p = yn; ty.n = N; ty.h = Ho; ty.w = Wo; ty.c = Co;
    ConvOpts o;
    o.sh = sh; o.sw = sw; o.pad = pad; o.act = act;
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() { return conv(r, L, tx, ty, o); }));
    OCRVI_TRY(k_nhwc_to_nchw_f32(dtype, yn, out, N, Ho, Wo, Co, Co, 0, sc.s));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}

// The 27-channel offset / mask conv of DeformableConv2d on its own (dcn.py:42-46): 3x3, pad 1, stride 1 or 2; out = device float32
// [N, Ho, Wo, 32]: channels 0..17 offsets, 18..26 sigmoid(mask logits), 27..31 zero -- the layout the deformable conv kernels consume.
extern "C" int ocrvi_test_offset_conv(int device, int dtype, const float* x, const float* weight_host, const float* bias_host, int N, int C,
                                      int H, int W, int stride, float* out, int iters, float* avg_ms) {
    OCRVI_CHECK(x && weight_host && bias_host && out && (stride == 1 || stride == 2) && N > 0 && C > 0, OCRVI_EINVAL, "test_offset_conv: bad argument");
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    DeviceStore st;
    ConvLayer L;
    PackedConv pc = pack_conv(weight_host, bias_host, 27, C, 3, 3, 1, AM_CONV3, dtype);
    OCRVI_TRY(upload_packed(st, pc, AM_CONV3, &L));
    void* xn = nullptr;
    OCRVI_TRY(sc.alloc((size_t)N * H * W * C * dtype_size(dtype), &xn));
    OCRVI_TRY(to_nhwc(dtype, x, xn, N, C, H * W, sc.s));
    Runner r(dtype, sc.s, (void*)256, 0);
    Tensor tx; tx.p = xn; tx.n = N; tx.h = H; tx.w = W; tx.c = C;
    Tensor ty; ty.p = out; ty.n = N; ty.h = Ho; ty.w = Wo; ty.c = 32; ty.f32 = true;
    ConvOpts o;
    o.sh = o.sw = stride; o.pad = 1; o.store_mode = ST_DCN_OFFS;
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() { return conv(r, L, tx, ty, o); }));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}

extern "C" int ocrvi_test_gemm(int device, int dtype, const float* a, const float* weight_host, const float* bias_host, const float* res, int M,
                               int K, int N, int act, int res_post, int out_f32, float* out, int iters, float* avg_ms) {
    OCRVI_CHECK(a && weight_host && out && M > 0 && K > 0 && N > 0, OCRVI_EINVAL, "test_gemm: bad argument");
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    const bool f32o = out_f32 || dtype == OCRVI_F32;   // (f16x2: its own 4-byte format unless out_f32)
    DeviceStore st;
    ConvLayer L;
    PackedConv pc = pack_conv(weight_host, bias_host, N, K, 1, 1, 1, AM_CONV1, dtype);
    OCRVI_TRY(upload_packed(st, pc, AM_CONV1, &L));
    void *an = nullptr, *yn = nullptr, *rn = nullptr;
    OCRVI_TRY(sc.alloc((size_t)M * K * dtype_size(dtype), &an));
    OCRVI_TRY(sc.alloc((size_t)M * N * (f32o ? 4 : dtype_size(dtype)), &yn));
    OCRVI_TRY(k_cast_from_f32(dtype, a, an, (size_t)M * K, sc.s));
    Runner r(dtype, sc.s, (void*)256, 0);
    Tensor tx; tx.p = an; tx.n = 1; tx.h = 1; tx.w = M; tx.c = K;
    Tensor ty; ty.p = yn; ty.n = 1; ty.h = 1; ty.w = M; ty.c = N; ty.f32 = f32o;
    Tensor tr = ty;
    ConvOpts o;
    o.act = act;
    if (res) {  // the residual has the output's element type
        if (f32o) {
            rn = (void*)res;
        } else {
            OCRVI_TRY(sc.alloc((size_t)M * N * dtype_size(dtype), &rn));
            OCRVI_TRY(k_cast_from_f32(dtype, res, rn, (size_t)M * N, sc.s));
        }
        tr.p = rn;
        o.res = &tr; o.res_mode = RES_SAME; o.res_post = res_post;
    }
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() { return conv(r, L, tx, ty, o); }));
    if (f32o) OCRVI_HIP(hipMemcpyAsync(out, yn, (size_t)M * N * 4, hipMemcpyDeviceToDevice, sc.s));
    else OCRVI_TRY(k_nhwc_to_nchw_f32(dtype, yn, out, 1, 1, M * N, 1, 1, 0, sc.s));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}

extern "C" int ocrvi_test_attention(int device, int dtype, const float* qkv, int B, int N, int heads, float* out, int iters, float* avg_ms) {
    OCRVI_CHECK(qkv && out && B > 0 && N > 0 && heads > 0, OCRVI_EINVAL, "test_attention: bad argument");
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    const int D = heads * 32;
    const size_t nin = (size_t)B * N * 3 * D, nout = (size_t)B * N * D;
    void *q = nullptr, *o = nullptr;
    OCRVI_TRY(sc.alloc(nin * dtype_size(dtype), &q));
    OCRVI_TRY(sc.alloc(nout * dtype_size(dtype), &o));
    OCRVI_TRY(k_cast_from_f32(dtype, qkv, q, nin, sc.s));
    void* asc = nullptr;
    if (const size_t sb = attention_scratch_bytes(dtype, B, N, heads)) OCRVI_TRY(sc.alloc(sb, &asc));
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() { return k_attention(dtype, q, o, B, N, heads, sc.s, asc); }));
    // [B*N][D] T -> float32 (same layout): a C=1 "NHWC -> NCHW" copy is a plain cast
    OCRVI_TRY(k_nhwc_to_nchw_f32(dtype, o, out, 1, 1, (int)nout, 1, 1, 0, sc.s));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}

extern "C" int ocrvi_test_mlp(int device, int dtype, float* x, const float* ln_g_host, const float* ln_b_host, const float* w1_host, const float* b1_host,
                              const float* w2_host, const float* b2_host, const float* next_g_host, const float* next_b_host, int want_xn, int M, int D,
                              float* xn_out, int iters, float* avg_ms) {
    OCRVI_CHECK(x && ln_g_host && ln_b_host && w1_host && b1_host && w2_host && b2_host && M > 0, OCRVI_EINVAL, "test_mlp: bad argument");
    OCRVI_CHECK(mlp_fused_eligible(dtype, D), OCRVI_EINVAL, "test_mlp: the fused MLP needs a 16-bit dtype and D in {128, 256, 384}, or f16x2 and D in {128, 256} (got dtype %d, D %d)", dtype, D);
    OCRVI_HIP(hipSetDevice(device));
    Scratch sc;
    OCRVI_TRY(sc.init());
    DeviceStore st;
    std::vector<char> packed;
    pack_mlp_stream(w1_host, w2_host, D, dtype, packed);
    void* ws = nullptr;
    float *g = nullptr, *b = nullptr, *b1 = nullptr, *b2 = nullptr, *ng = nullptr, *nb = nullptr;
    OCRVI_TRY(st.upload(packed.data(), packed.size(), &ws));
    OCRVI_TRY(st.upload(ln_g_host, (size_t)D * 4, (void**)&g));
    OCRVI_TRY(st.upload(ln_b_host, (size_t)D * 4, (void**)&b));
    OCRVI_TRY(st.upload(b1_host, (size_t)4 * D * 4, (void**)&b1));
    OCRVI_TRY(st.upload(b2_host, (size_t)D * 4, (void**)&b2));
    if (next_g_host) {
        OCRVI_TRY(st.upload(next_g_host, (size_t)D * 4, (void**)&ng));
        OCRVI_TRY(st.upload(next_b_host, (size_t)D * 4, (void**)&nb));
    }
    void *xn = nullptr, *x0 = nullptr;
    if (want_xn) OCRVI_TRY(sc.alloc((size_t)M * D * dtype_size(dtype), &xn));
    OCRVI_TRY(sc.alloc((size_t)M * D * 4, &x0));   // the kernel updates x in place: every timed repeat starts from the caller's x
    OCRVI_HIP(hipMemcpyAsync(x0, x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, sc.s));
    OCRVI_TRY(timed(sc, iters, avg_ms, [&]() {
        return k_mlp_fused(dtype, (float*)x0, xn, g, b, ng, nb, ws, b1, b2, M, D, sc.s);
    }));
    // one clean application for the result
    OCRVI_TRY(k_mlp_fused(dtype, x, xn, g, b, ng, nb, ws, b1, b2, M, D, sc.s));
    if (want_xn && xn_out) OCRVI_TRY(k_nhwc_to_nchw_f32(dtype, xn, xn_out, 1, 1, M * D, 1, 1, 0, sc.s));
    OCRVI_HIP(hipStreamSynchronize(sc.s));
    return OCRVI_OK;
}
