// SVTRv2 inference graph (model/rec2/svtrv2.py:475-536) + device half of greedy CTC decode (:555-566).
// The residual stream is kept in fp32; GEMM operands are the handle's compute dtype.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>

#include "gemm_ring.h"
#include "model.h"

using namespace ocrvi;

namespace {
struct LNw { float *g = nullptr, *b = nullptr; };
struct MlpW {
    ConvLayer fc1, fc2;
    void* stream = nullptr;      // 16-bit modes, D % 128 == 0: fc1 / fc2 packed for the fused MLP kernel (mlp_fused.hip)
    float *b1 = nullptr, *b2 = nullptr;
};
struct BlockW {
    bool local = false;
    LNw n1, n2;
    ConvLayer conv1, conv2;  // local
    ConvLayer qkv, proj;     // global
    MlpW mlp;
};
}  // namespace

struct ocrvi_rec {
    int device = 0;
    ocrvi_rec_cfg cfg{};
    DeviceStore store;
    ConvLayer stem1, stem2;
    std::vector<BlockW> blocks[3];
    ConvLayer merge[2];
    LNw backbone_norm;
    LNw h_norm, h_norm2, v_norm_kv, v_norm2;
    ConvLayer h_qkv, h_proj, v_kv, v_proj, head;
    MlpW h_mlp, v_mlp;
    float* vq = nullptr;
    // pointers into the last forward's workspace (debug taps)
    void *tap_bn = nullptr, *tap_frm = nullptr;
    RangeWatch range;     // f16x2 only: the device's range flag as of the end of the last forward
};

static int load_ln(DeviceStore& st, const Blob& b, const std::string& name, int d, LNw* ln) {
    OCRVI_TRY(load_vec(st, b, name + ".w", d, &ln->g));
    return load_vec(st, b, name + ".b", d, &ln->b);
}
static int load_lin(DeviceStore& st, const Blob& b, const std::string& name, int cout, int cin, int dt, ConvLayer* L, const float* extra = nullptr) {
    return load_conv(st, b, name, cout, cin, 0, 1, AM_CONV1, dt, true, L, extra);
}
static int load_mlp(DeviceStore& st, const Blob& b, const std::string& name, int d, int dt, MlpW* m) {
    OCRVI_TRY(load_lin(st, b, name + ".fc1", 4 * d, d, dt, &m->fc1));
    OCRVI_TRY(load_lin(st, b, name + ".fc2", d, 4 * d, dt, &m->fc2));
    static const bool fuse = !(getenv("OCRVI_MLP_FUSED") && atoi(getenv("OCRVI_MLP_FUSED")) == 0);   // A/B switch
    if (fuse && mlp_fused_eligible(dt, d)) {
        const BlobTensor *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
        OCRVI_TRY(b.get(name + ".fc1.w", 4 * d, d, 0, 0, &w1));
        OCRVI_TRY(b.get(name + ".fc1.b", 4 * d, 0, 0, 0, &b1));
        OCRVI_TRY(b.get(name + ".fc2.w", d, 4 * d, 0, 0, &w2));
        OCRVI_TRY(b.get(name + ".fc2.b", d, 0, 0, 0, &b2));
        std::vector<char> packed;
        pack_mlp_stream(w1->data, w2->data, d, dt, packed);
        OCRVI_TRY(st.upload(packed.data(), packed.size(), &m->stream));
        OCRVI_TRY(st.upload(b1->data, (size_t)4 * d * 4, (void**)&m->b1));
        OCRVI_TRY(st.upload(b2->data, (size_t)d * 4, (void**)&m->b2));
    }
    return OCRVI_OK;
}

extern "C" int ocrvi_rec_create(int device, const void* blob_p, size_t blob_bytes, const ocrvi_rec_cfg* cfg, ocrvi_rec** out) {
    OCRVI_CHECK(cfg && out, OCRVI_EINVAL, "rec_create: null argument");
    OCRVI_CHECK(dtype_valid(cfg->dtype), OCRVI_EINVAL, "rec_create: bad dtype %d", cfg->dtype);
    for (int s = 0; s < 3; ++s)
        OCRVI_CHECK(cfg->dims[s] > 0 && cfg->dims[s] % 32 == 0 && cfg->num_blocks[s] > 0 && cfg->num_local[s] >= 0 &&
                        cfg->num_local[s] <= cfg->num_blocks[s],
                    OCRVI_EINVAL, "rec_create: bad stage %d config (dims must be multiples of 32)", s);
    OCRVI_CHECK(cfg->num_classes > 0 && cfg->num_classes % 4 == 0 && cfg->num_classes <= 1024, OCRVI_EINVAL,
                "rec_create: num_classes=%d must be a multiple of 4 and <= 1024", cfg->num_classes);
    DeviceGuard dg(device);  // the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    Blob blob;
    OCRVI_TRY(blob.parse(blob_p, blob_bytes));
    std::unique_ptr<ocrvi_rec> h(new ocrvi_rec);
    h->device = device;
    h->cfg = *cfg;
    const int dt = cfg->dtype;
    DeviceStore& st = h->store;
    const int d0 = cfg->dims[0], mid = d0 / 2;
    OCRVI_TRY(load_conv(st, blob, "stem.conv1", mid, 3, 3, 1, AM_ROWS, dt, true, &h->stem1));
    OCRVI_TRY(load_conv(st, blob, "stem.conv2", d0, mid, 3, 1, AM_CONV3, dt, true, &h->stem2));
    for (int s = 0; s < 3; ++s) {
        const int d = cfg->dims[s];
        h->blocks[s].resize(cfg->num_blocks[s]);
        for (int b = 0; b < cfg->num_blocks[s]; ++b) {
            BlockW& bw = h->blocks[s][b];
            const std::string p = "stages." + std::to_string(s) + ".blocks." + std::to_string(b);
            bw.local = b < cfg->num_local[s];
            OCRVI_TRY(load_ln(st, blob, p + ".norm1", d, &bw.n1));
            OCRVI_TRY(load_ln(st, blob, p + ".norm2", d, &bw.n2));
            if (bw.local) {
                const int groups = d / 32;  // LocalMixing: groups = dim // 32 (svtrv2.py:47)
                OCRVI_TRY(load_conv(st, blob, p + ".mixer.conv1", d, 32, 3, groups, AM_CONV3, dt, true, &bw.conv1));
                OCRVI_TRY(load_conv(st, blob, p + ".mixer.conv2", d, 32, 3, groups, AM_CONV3, dt, true, &bw.conv2));
            } else {
                OCRVI_TRY(load_lin(st, blob, p + ".mixer.qkv", 3 * d, d, dt, &bw.qkv));
                OCRVI_TRY(load_lin(st, blob, p + ".mixer.proj", d, d, dt, &bw.proj));
            }
            OCRVI_TRY(load_mlp(st, blob, p + ".mlp", d, dt, &bw.mlp));
        }
        if (s < 2) OCRVI_TRY(load_conv(st, blob, "merges." + std::to_string(s), cfg->dims[s + 1], d, 3, 1, AM_CONV3, dt, true, &h->merge[s]));
    }
    const int d = cfg->dims[2];
    OCRVI_TRY(load_ln(st, blob, "backbone_norm", d, &h->backbone_norm));
    OCRVI_TRY(load_ln(st, blob, "frm.h_norm", d, &h->h_norm));
    OCRVI_TRY(load_ln(st, blob, "frm.h_norm2", d, &h->h_norm2));
    OCRVI_TRY(load_ln(st, blob, "frm.v_norm_kv", d, &h->v_norm_kv));
    OCRVI_TRY(load_ln(st, blob, "frm.v_norm2", d, &h->v_norm2));
    OCRVI_TRY(load_lin(st, blob, "frm.h_qkv", 3 * d, d, dt, &h->h_qkv));
    OCRVI_TRY(load_lin(st, blob, "frm.h_proj", d, d, dt, &h->h_proj));
    OCRVI_TRY(load_lin(st, blob, "frm.v_kv", 2 * d, d, dt, &h->v_kv));
    // t_q + v_proj(out): the residual t_q is the (input-independent) select token -> fold it into the bias (svtrv2.py:226,244)
    const BlobTensor* tok = nullptr;
    OCRVI_TRY(blob.get("frm.select_token", d, 0, 0, 0, &tok));
    OCRVI_TRY(load_lin(st, blob, "frm.v_proj", d, d, dt, &h->v_proj, tok->data));
    OCRVI_TRY(load_mlp(st, blob, "frm.h_mlp", d, dt, &h->h_mlp));
    OCRVI_TRY(load_mlp(st, blob, "frm.v_mlp", d, dt, &h->v_mlp));
    OCRVI_TRY(load_vec(st, blob, "frm.vq", d, &h->vq));
    OCRVI_TRY(load_lin(st, blob, "head", cfg->num_classes, d, dt, &h->head));
    {   // scratch pages of the ring GEMM: allocate now so no forward (possibly under graph capture) ever allocates
        const void* z; void* d;
        OCRVI_TRY(ring_pages(&z, &d));
    }
    if (dt == OCRVI_F16X2) OCRVI_TRY(h->range.init());
    *out = h.release();
    return OCRVI_OK;
}

extern "C" void ocrvi_rec_destroy(ocrvi_rec* h) {
    if (!h) return;
    DeviceGuard dg(h->device);
    delete h;
}

static int check_rec_shape(const ocrvi_rec* h, int B, int H, int W) {
    OCRVI_CHECK(h, OCRVI_EINVAL, "rec: null handle");
    OCRVI_CHECK(B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 4 == 0, OCRVI_EINVAL,
                "rec: input (B=%d,3,%d,%d) needs H %% 16 == 0 and W %% 4 == 0", B, H, W);
    // svtrv2.py:503-536 takes any size; here the first global-attention stage sees (H/8)(W/4) tokens: the 16-bit kernels stage every key
    // in one workgroup's LDS (<= 1024), the 4-byte modes split the keys into chunks of <= 512 and merge (attention.hip), FRM's vertical
    // attention takes up to 8 rows
    const int tok = (H / 4) * (W / 4) / 2, tok_max = dtype_size(h->cfg.dtype) == 2 ? 1024 : 4096;
    OCRVI_CHECK(tok <= tok_max && H / 16 <= 8, OCRVI_EINVAL,
                "rec: %dx%d gives %d tokens in the first global-attention stage (at most %d in this mode) or more than 8 rows after the backbone", H, W, tok, tok_max);
    OCRVI_CHECK((size_t)B * (H / 2) * (W / 2) < ((size_t)1 << 23), OCRVI_EINVAL, "rec: batch of %d %dx%d crops too large for one call (chunk it)", B, H, W);
    return OCRVI_OK;
}

static int ln(Runner& r, const Tensor& x, const Tensor& y, const LNw& w) {
    if (r.dry()) return OCRVI_OK;
    return k_layernorm(r.dtype, x.p, x.f32, y.p, y.f32, w.g, w.b, (int)x.pixels(), x.c, r.stream);
}
static Tensor view(const Tensor& t, int n, int h, int w, int c) {
    Tensor v = t;
    v.n = n; v.h = h; v.w = w; v.c = c;
    return v;
}

// y = x + fc2(gelu(fc1(LN(x))))   (svtrv2.py:38-39,100).  `next`: what the caller needs in xn afterwards -- nullptr: nothing; a norm:
// LN(y) with it; &kCast: T(y).  Returns through *xn_done whether xn was produced (the fused kernel does it in its epilogue).
static const LNw kCast{};
static int mlp_block(Runner& r, const Tensor& x, const Tensor& xn, const Tensor& hb, const LNw& n, const MlpW& m, int d, const LNw* next = nullptr,
                     bool* xn_done = nullptr) {
    const int rows = (int)x.pixels();
    if (xn_done) *xn_done = false;
    if (m.stream) {
        if (xn_done) *xn_done = next != nullptr;
        if (r.dry()) return OCRVI_OK;
        return k_mlp_fused(r.dtype, (float*)x.p, next ? xn.p : nullptr, n.g, n.b, next ? next->g : nullptr, next ? next->b : nullptr, m.stream, m.b1, m.b2,
                           rows, d, r.stream);
    }
    OCRVI_TRY(ln(r, x, xn, n));
    ConvOpts o1;
    o1.act = ACT_GELU;
    OCRVI_TRY(conv(r, m.fc1, view(xn, rows, 1, 1, d), view(hb, rows, 1, 1, 4 * d), o1));
    ConvOpts o2;
    o2.res = &x;
    o2.res_mode = RES_SAME;
    return conv(r, m.fc2, view(hb, rows, 1, 1, 4 * d), view(x, rows, 1, 1, d), o2);
}

static int rec_run(ocrvi_rec* h, Runner& r, const float* x, int B, int H, int W, float* log_probs, int32_t* argmax_ids, int32_t* ids,
                   int32_t* lens) {
    const ocrvi_rec_cfg& c = h->cfg;
    const int dt = c.dtype;
    const bool lowp = dt != OCRVI_F32;
    int Hs = H / 4, Ws = W / 4;
    const int T = Ws;
    // ---- buffers (sized for the widest stage)
    size_t max_tok_d = 0, max_tok = 0;
    {
        int hh = Hs;
        for (int s = 0; s < 3; ++s) {
            const size_t tok = (size_t)B * hh * Ws;
            max_tok = std::max(max_tok, tok);
            max_tok_d = std::max(max_tok_d, tok * c.dims[s]);
            if (s < 2) hh /= 2;
        }
    }
    const int Hp = H + 2, Wp = W + 8;
    Tensor xpad = r.alloc(B, Hp, Wp, 4);
    Tensor s1 = r.alloc(B, H / 2, W / 2, c.dims[0] / 2);
    void* X[2] = {r.arena.alloc(max_tok_d * 4), r.arena.alloc(max_tok_d * 4)};  // fp32 residual stream (ping-pong at merges)
    void* XN = r.arena.alloc(max_tok_d * dtype_size(dt));
    void* T1 = r.arena.alloc(max_tok_d * dtype_size(dt));
    void* BIG = r.arena.alloc(max_tok_d * 4 * dtype_size(dt));  // qkv (3d) or MLP hidden (4d)
    const int d2 = c.dims[2];
    Tensor logits = r.alloc(B * T, 1, 1, c.num_classes, true);
    int32_t* am_buf = (int32_t*)r.arena.alloc((size_t)B * T * 4);

    if (!r.dry()) OCRVI_TRY(k_nchw3_to_nhwc4_pad(dt, x, xpad.p, B, H, W, 1, 1, Hp, Wp, r.stream));
    {   // ConvStem (svtrv2.py:118-122): conv3x3/2 + BN + GELU, twice
        ConvOpts o;
        o.sh = o.sw = 2; o.pad = 1; o.act = ACT_GELU; o.Hp = Hp; o.Wp = Wp;
        OCRVI_TRY(conv(r, h->stem1, xpad, s1, o));
    }
    int cur = 0;
    Tensor xs;
    xs.p = X[cur]; xs.n = B; xs.h = Hs; xs.w = Ws; xs.c = c.dims[0]; xs.f32 = true;
    {
        ConvOpts o;
        o.sh = o.sw = 2; o.pad = 1; o.act = ACT_GELU;
        OCRVI_TRY(conv(r, h->stem2, s1, xs, o));
    }
    for (int s = 0; s < 3; ++s) {
        const int d = c.dims[s];
        const int rows = B * Hs * Ws;
        Tensor xn; xn.p = XN; xn.n = B; xn.h = Hs; xn.w = Ws; xn.c = d; xn.f32 = false;
        Tensor t1 = xn; t1.p = T1;
        Tensor big; big.p = BIG; big.n = rows; big.h = big.w = 1; big.c = 4 * d; big.f32 = false;
        bool xn_ready = false;   // the previous block's fused MLP already wrote LN(x; norm1) (or the cast the merge needs) into xn
        const int nblk = (int)h->blocks[s].size();
        for (int bi = 0; bi < nblk; ++bi) {
            const BlockW& bw = h->blocks[s][bi];
            if (!xn_ready) OCRVI_TRY(ln(r, xs, xn, bw.n1));
            if (bw.local) {  // x + gelu(bn(conv(gelu(bn(conv(LN x))))))  (svtrv2.py:57-63,98)
                ConvOpts o;
                o.pad = 1; o.act = ACT_GELU;
                OCRVI_TRY(conv(r, bw.conv1, xn, t1, o));
                o.res = &xs; o.res_mode = RES_SAME; o.res_post = 1;
                OCRVI_TRY(conv(r, bw.conv2, t1, xs, o));
            } else {  // x + proj(MHSA(qkv(LN x)))  (svtrv2.py:77-86,98)
                ConvOpts o;
                OCRVI_TRY(conv(r, bw.qkv, view(xn, rows, 1, 1, d), view(big, rows, 1, 1, 3 * d), o));
                {   // (sequences beyond 512 keys in the 4-byte modes: per-chunk partial rows, merged; scratch from the arena)
                    const size_t mk = r.arena.mark(), sb = attention_scratch_bytes(dt, B, Hs * Ws, d / 32);
                    void* asc = sb ? r.arena.alloc(sb) : nullptr;
                    if (!r.dry()) OCRVI_TRY(k_attention(dt, big.p, t1.p, B, Hs * Ws, d / 32, r.stream, asc));
                    r.arena.release(mk);
                }
                o.res = &xs; o.res_mode = RES_SAME;
                OCRVI_TRY(conv(r, bw.proj, view(t1, rows, 1, 1, d), view(xs, rows, 1, 1, d), o));
            }
            const LNw* next = bi + 1 < nblk ? &h->blocks[s][bi + 1].n1 : ((s < 2 && lowp) ? &kCast : nullptr);
            OCRVI_TRY(mlp_block(r, view(xs, rows, 1, 1, d), xn, big, bw.n2, bw.mlp, d, next, &xn_ready));
        }
        if (s < 2) {  // PatchMerging (svtrv2.py:131-138): conv3x3 stride (2,1) + BN, no activation
            Tensor src = xn;
            if (lowp) {
                if (!xn_ready && !r.dry()) OCRVI_TRY(k_cast_from_f32(dt, (const float*)xs.p, xn.p, (size_t)rows * d, r.stream));
            } else {
                src = xs; src.f32 = false;
            }
            Tensor nx;
            nx.p = X[cur ^ 1]; nx.n = B; nx.h = Hs / 2; nx.w = Ws; nx.c = c.dims[s + 1]; nx.f32 = true;
            ConvOpts o;
            o.sh = 2; o.sw = 1; o.pad = 1;
            OCRVI_TRY(conv(r, h->merge[s], src, nx, o));
            xs = nx;
            cur ^= 1;
            Hs /= 2;
        }
    }
    // ---- backbone_norm (svtrv2.py:500) -> FRM (svtrv2.py:192-247)
    const int rows = B * Hs * Ws;
    Tensor bn; bn.p = X[cur ^ 1]; bn.n = rows; bn.h = bn.w = 1; bn.c = d2; bn.f32 = true;
    OCRVI_TRY(ln(r, view(xs, rows, 1, 1, d2), bn, h->backbone_norm));
    h->tap_bn = bn.p;
    Tensor xr; xr.p = X[cur]; xr.n = rows; xr.h = xr.w = 1; xr.c = d2; xr.f32 = true;  // stage output no longer needed
    Tensor xn; xn.p = XN; xn.n = rows; xn.h = xn.w = 1; xn.c = d2; xn.f32 = false;
    Tensor t1 = xn; t1.p = T1;
    Tensor big; big.p = BIG; big.n = rows; big.h = big.w = 1; big.c = 4 * d2; big.f32 = false;
    OCRVI_TRY(ln(r, bn, xn, h->h_norm));
    {
        ConvOpts o;
        OCRVI_TRY(conv(r, h->h_qkv, xn, view(big, rows, 1, 1, 3 * d2), o));
        {   // one sequence per image row
            const size_t mk = r.arena.mark(), sb = attention_scratch_bytes(dt, B * Hs, Ws, d2 / 32);
            void* asc = sb ? r.arena.alloc(sb) : nullptr;
            if (!r.dry()) OCRVI_TRY(k_attention(dt, big.p, t1.p, B * Hs, Ws, d2 / 32, r.stream, asc));
            r.arena.release(mk);
        }
        o.res = &bn; o.res_mode = RES_SAME;
        OCRVI_TRY(conv(r, h->h_proj, t1, xr, o));
    }
    bool xn_ready = false;
    OCRVI_TRY(mlp_block(r, xr, xn, big, h->h_norm2, h->h_mlp, d2, &h->v_norm_kv, &xn_ready));
    if (!xn_ready) OCRVI_TRY(ln(r, xr, xn, h->v_norm_kv));
    const int cols = B * Ws;
    Tensor tq; tq.p = bn.p; tq.n = cols; tq.h = tq.w = 1; tq.c = d2; tq.f32 = true;
    // note: tq would alias the backbone_norm tap; use a fresh buffer so the tap survives
    tq.p = r.arena.alloc((size_t)cols * d2 * 4);
    {
        ConvOpts o;
        OCRVI_TRY(conv(r, h->v_kv, xn, view(big, rows, 1, 1, 2 * d2), o));
        if (!r.dry()) OCRVI_TRY(k_frm_vertical(dt, big.p, h->vq, t1.p, B, Hs, Ws, d2, r.stream));
        OCRVI_TRY(conv(r, h->v_proj, view(t1, cols, 1, 1, d2), tq, o));  // bias carries + select_token
    }
    OCRVI_TRY(mlp_block(r, tq, view(xn, cols, 1, 1, d2), big, h->v_norm2, h->v_mlp, d2, lowp ? &kCast : nullptr, &xn_ready));
    h->tap_frm = tq.p;
    // ---- CTC head (svtrv2.py:528-532) + greedy decode (svtrv2.py:555-566)
    Tensor hin = view(xn, cols, 1, 1, d2);
    if (lowp) {
        if (!xn_ready && !r.dry()) OCRVI_TRY(k_cast_from_f32(dt, (const float*)tq.p, xn.p, (size_t)cols * d2, r.stream));
    } else {
        hin = tq; hin.f32 = false;
    }
    {
        ConvOpts o;
        OCRVI_TRY(conv(r, h->head, hin, logits, o));
    }
    if (!r.dry()) {
        int32_t* am = argmax_ids ? argmax_ids : am_buf;
        OCRVI_TRY(k_ctc_logsoftmax_argmax((const float*)logits.p, c.num_classes, log_probs, am, B, T, c.num_classes, r.stream));
        if (ids || lens) OCRVI_TRY(k_ctc_collapse(am, ids, lens, B, T, c.blank_id, r.stream));
    }
    return OCRVI_OK;
}

extern "C" int ocrvi_rec_workspace_bytes(const ocrvi_rec* h, int B, int H, int W, size_t* bytes) {
    OCRVI_CHECK(bytes, OCRVI_EINVAL, "rec_workspace_bytes: null out");
    OCRVI_TRY(check_rec_shape(h, B, H, W));
    Runner r(h->cfg.dtype, nullptr, nullptr, 0);
    OCRVI_TRY(rec_run(const_cast<ocrvi_rec*>(h), r, nullptr, B, H, W, nullptr, nullptr, nullptr, nullptr));
    *bytes = r.arena.peak + 256;
    return OCRVI_OK;
}

extern "C" int ocrvi_rec_forward(ocrvi_rec* h, const float* x, int B, int H, int W, float* log_probs, int32_t* argmax_ids, int32_t* ids,
                                 int32_t* lens, void* workspace, size_t workspace_bytes, void* stream) {
    OCRVI_TRY(check_rec_shape(h, B, H, W));
    OCRVI_CHECK(x && workspace, OCRVI_EINVAL, "rec_forward: null input/workspace");
    DeviceGuard dg(h->device);  // launches go to the handle's device whatever the caller's current device is
    OCRVI_HIP(dg.err);
    size_t need = 0;
    OCRVI_TRY(ocrvi_rec_workspace_bytes(h, B, H, W, &need));
    OCRVI_CHECK(workspace_bytes >= need, OCRVI_ENOMEM, "rec_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    OCRVI_CHECK(((uintptr_t)workspace & 255) == 0, OCRVI_EINVAL, "rec_forward: workspace must be 256-byte aligned");
    Runner r(h->cfg.dtype, (hipStream_t)stream, workspace, workspace_bytes);
    OCRVI_TRY(rec_run(h, r, x, B, H, W, log_probs, argmax_ids, ids, lens));
    OCRVI_CHECK(!r.arena.overflow, OCRVI_ENOMEM, "rec_forward: workspace overflow");
    return h->range.snapshot((hipStream_t)stream);
}

extern "C" int ocrvi_rec_status(const ocrvi_rec* h) {
    OCRVI_CHECK(h, OCRVI_EINVAL, "rec_status: null handle");
    return h->range.status("rec");
}

extern "C" int ocrvi_rec_debug_features(ocrvi_rec* h, int B, int H, int W, float* backbone_norm, float* frm, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    OCRVI_TRY(check_rec_shape(h, B, H, W));
    OCRVI_CHECK(h->tap_bn && h->tap_frm, OCRVI_EINVAL, "rec_debug_features: no forward has run");
    DeviceGuard dg(h->device);
    OCRVI_HIP(dg.err);
    (void)workspace; (void)workspace_bytes;
    const int d = h->cfg.dims[2];
    const size_t ntok = (size_t)B * (H / 16) * (W / 4), ncol = (size_t)B * (W / 4);
    if (backbone_norm) OCRVI_HIP(hipMemcpyAsync(backbone_norm, h->tap_bn, ntok * d * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (frm) OCRVI_HIP(hipMemcpyAsync(frm, h->tap_frm, ncol * d * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return OCRVI_OK;
}

extern "C" int ocrvi_ctc_greedy(int device, const float* log_probs, int T, int B, int C, int blank_id, int32_t* argmax_ids, int32_t* ids,
                                int32_t* lens, void* stream) {
    OCRVI_CHECK(log_probs && argmax_ids, OCRVI_EINVAL, "ctc_greedy: log_probs and argmax_ids are required");
    DeviceGuard dg(device);  // the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    OCRVI_TRY(k_ctc_argmax_tbc(log_probs, argmax_ids, B, T, C, (hipStream_t)stream));
    if (ids || lens) OCRVI_TRY(k_ctc_collapse(argmax_ids, ids, lens, B, T, blank_id, (hipStream_t)stream));
    return OCRVI_OK;
}
