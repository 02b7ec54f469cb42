// Host-side plumbing shared by the detection and recognition graphs: device-resident packed weights,
// NHWC tensor views over the caller's workspace, and a conv() wrapper that fills ConvParams.
#pragma once
#include "conv_gemm.h"
#include "kernels.h"

namespace ocrvi {

struct DeviceStore {  // owns every hipMalloc'd weight buffer of a handle
    std::vector<void*> ptrs;
    ~DeviceStore() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    int upload(const void* host, size_t bytes, void** out) {
        void* d = nullptr;
        OCRVI_HIP(hipMalloc(&d, bytes ? bytes : 16));
        ptrs.push_back(d);
        if (bytes) OCRVI_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
        *out = d;
        return OCRVI_OK;
    }
    int upload_f32(const std::vector<float>& v, float** out) { return upload(v.data(), v.size() * 4, (void**)out); }
};

struct ConvLayer {
    void* w = nullptr;
    float* bias = nullptr;
    int Np = 0, Kp = 0, N_g = 0, Cin_g = 0, groups = 1, KH = 1, amode = AM_CONV1;
    int shuffle_co = 0;
    float wscale = 1.f;   // f16x2 weight scale (PackedConv::wscale)
};

int upload_packed(DeviceStore& st, const PackedConv& pc, int amode, ConvLayer* L);
// conv weight `name`.w [cout][cin_g][k][k] (+ `name`.b [cout] if has_bias) from the blob
int load_conv(DeviceStore& st, const Blob& blob, const std::string& name, int cout, int cin_g, int k, int groups, int amode, int dtype,
              bool has_bias, ConvLayer* L, const float* extra_bias = nullptr);
int load_vec(DeviceStore& st, const Blob& blob, const std::string& name, int n, float** out);

struct Tensor {  // NHWC view
    void* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    bool f32 = false;
    size_t pixels() const { return (size_t)n * h * w; }
};

struct Runner {
    int dtype;
    hipStream_t stream;
    Arena arena;
    Runner(int dt, hipStream_t s, void* ws, size_t bytes) : dtype(dt), stream(s), arena(ws, bytes) {}
    bool dry() const { return arena.planning(); }
    size_t esz(bool f32) const { return f32 ? 4 : dtype_size(dtype); }
    Tensor alloc(int n, int h, int w, int c, bool f32 = false) {
        Tensor t;
        t.n = n; t.h = h; t.w = w; t.c = c; t.f32 = f32;
        t.p = arena.alloc((size_t)n * h * w * c * esz(f32));
        return t;
    }
};

struct ConvOpts {
    int sh = 1, sw = 1, pad = 0;
    int act = ACT_NONE;
    const Tensor* res = nullptr;
    int res_mode = RES_NONE;
    int res_post = 0;
    int store_mode = ST_NHWC;
    const float* offs = nullptr;  // AM_DCN offsets, or ST_DB_TAIL second-deconv weights [2][64][4] + biases [2]
    void* out2 = nullptr;         // ST_DB_TAIL: threshold-branch logit map
    int out_coff = 0;
    int cin_off = 0;
    int Hp = 0, Wp = 0;  // AM_ROWS padded input dims
};
// y must be pre-shaped (n,h,w,c, f32) and allocated; x likewise.
int conv(Runner& r, const ConvLayer& L, const Tensor& x, const Tensor& y, const ConvOpts& o);

}  // namespace ocrvi
