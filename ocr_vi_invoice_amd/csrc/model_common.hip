#include "model.h"

namespace ocrvi {

int upload_packed(DeviceStore& st, const PackedConv& pc, int amode, ConvLayer* L) {
    OCRVI_TRY(st.upload(pc.bytes.data(), pc.bytes.size(), &L->w));
    L->bias = nullptr;
    if (!pc.bias.empty()) OCRVI_TRY(st.upload_f32(pc.bias, &L->bias));
    L->Np = pc.Np; L->Kp = pc.Kp; L->N_g = pc.N_g; L->Cin_g = pc.Cin_g; L->groups = pc.groups; L->KH = pc.KH;
    L->amode = amode;
    L->wscale = pc.wscale;
    return OCRVI_OK;
}

int load_conv(DeviceStore& st, const Blob& blob, const std::string& name, int cout, int cin_g, int k, int groups, int amode, int dtype,
              bool has_bias, ConvLayer* L, const float* extra_bias) {
    const BlobTensor *w = nullptr, *b = nullptr;
    OCRVI_TRY(blob.get(name + ".w", cout, cin_g, k, k, &w));
    if (has_bias) OCRVI_TRY(blob.get(name + ".b", cout, 0, 0, 0, &b));
    std::vector<float> bias;
    if (b) bias.assign(b->data, b->data + cout);
    if (extra_bias) {
        if (bias.empty()) bias.assign(cout, 0.f);
        for (int i = 0; i < cout; ++i) bias[i] += extra_bias[i];
    }
    const int kk = k == 0 ? 1 : k;  // k == 0: a 2-D nn.Linear weight (out, in)
    PackedConv pc = pack_conv(w->data, bias.empty() ? nullptr : bias.data(), cout, cin_g, kk, kk, groups, amode, dtype);
    return upload_packed(st, pc, amode, L);
}

int load_vec(DeviceStore& st, const Blob& blob, const std::string& name, int n, float** out) {
    const BlobTensor* t = nullptr;
    OCRVI_TRY(blob.get(name, n, 0, 0, 0, &t));
    return st.upload(t->data, (size_t)n * 4, (void**)out);
}

int conv(Runner& r, const ConvLayer& L, const Tensor& x, const Tensor& y, const ConvOpts& o) {
    ConvParams p;
    p.x = x.p; p.w = L.w; p.bias = L.bias; p.out = y.p;
    p.n_img = x.n; p.H = x.h; p.W = x.w; p.Cin = x.c;
    p.KH = L.KH; p.SH = o.sh; p.SW = o.sw; p.PH = o.pad; p.PW = o.pad;
    p.Cin_g = L.Cin_g; p.cin_off = o.cin_off;
    p.N_g = L.N_g; p.Np = L.Np; p.Kp = L.Kp; p.groups = L.groups;
    p.store_mode = o.store_mode;
    p.wscale = L.wscale;
    if (o.store_mode == ST_SHUFFLE2) {
        p.OH = y.h / 2; p.OW = y.w / 2; p.shuffle_co = L.shuffle_co;
    } else if (o.store_mode == ST_DB_TAIL) {  // y is the fp32 logit map [n, 4*OH, 4*OW, 1]
        p.OH = y.h / 4; p.OW = y.w / 4; p.shuffle_co = L.shuffle_co; p.out2 = o.out2;
    } else {
        p.OH = y.h; p.OW = y.w;
    }
    p.M = x.n * p.OH * p.OW;
    p.ldo = o.store_mode == ST_DCN_OFFS ? 32 : y.c;
    p.out_coff = o.out_coff;
    p.out_f32 = y.f32 ? 1 : 0;
    p.act = o.act;
    p.res_post = o.res_post;
    if (o.res) {
        p.res = o.res->p; p.res_mode = o.res_mode; p.ldr = o.res->c; p.res_f32 = o.res->f32 ? 1 : 0;
    }
    p.offs = o.offs;
    p.Hp = o.Hp; p.Wp = o.Wp;
    OCRVI_CHECK(!x.f32 || r.dtype == OCRVI_F32, OCRVI_EINVAL, "conv: f32 activations fed to a %d-typed GEMM", r.dtype);
    if (L.amode != AM_ROWS) {
        const int k = L.amode == AM_CONV1 ? 1 : 3;
        const int eh = (x.h + 2 * o.pad - k) / o.sh + 1, ew = (x.w + 2 * o.pad - k) / o.sw + 1;
        OCRVI_CHECK(eh == p.OH && ew == p.OW, OCRVI_EINVAL, "conv: output %dx%d but expected %dx%d", p.OH, p.OW, eh, ew);
    }
    if (r.dry()) return OCRVI_OK;
    return launch_conv_dt(r.dtype, p, L.amode, r.stream);
}

}  // namespace ocrvi
