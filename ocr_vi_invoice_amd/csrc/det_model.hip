// DBNet++ inference graph (model/det/dbnet.py:13-17): ResNet-50 with DCNv2 in layers 2-4 (backbone.py:39-60,
// dcn.py:41-59), FPN + adaptive scale fusion (neck.py:26-79), DB head (head.py:32-48).  NHWC activations in the
// handle's compute dtype; eval-mode BatchNorm is pre-folded into the conv weights by the Python packer.
#include <algorithm>
#include <memory>

#include "gemm_ring.h"
#include "model.h"

using namespace ocrvi;

namespace {
const int kBlocks[4] = {3, 4, 6, 3};
const int kWidth[4] = {64, 128, 256, 512};
struct Bottleneck {
    ConvLayer conv1, conv2, off, conv3, down;
    bool dcn = false, has_down = false;
    int stride = 1;
};
}  // namespace

struct ocrvi_det {
    int device = 0;
    ocrvi_det_cfg cfg{};
    DeviceStore store;
    ConvLayer stem;
    std::vector<Bottleneck> layers[4];
    ConvLayer lat[4], fpn[4];
    float *asf_w = nullptr, *asf_b = nullptr;
    ConvLayer head_conv, head_dc1;
    float* dc2_wb = nullptr;  // [2][64][4] second-deconv weights followed by the 2 biases
    Tensor tap_c[4], tap_fused;
    RangeWatch range;     // f16x2 only: the device's range flag as of the end of the last forward
};

extern "C" int ocrvi_det_create(int device, const void* blob_p, size_t blob_bytes, const ocrvi_det_cfg* cfg, ocrvi_det** out) {
    OCRVI_CHECK(cfg && out, OCRVI_EINVAL, "det_create: null argument");
    OCRVI_CHECK(dtype_valid(cfg->dtype), OCRVI_EINVAL, "det_create: bad dtype %d", cfg->dtype);
    DeviceGuard dg(device);  // the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    Blob blob;
    OCRVI_TRY(blob.parse(blob_p, blob_bytes));
    std::unique_ptr<ocrvi_det> h(new ocrvi_det);
    h->device = device;
    h->cfg = *cfg;
    const int dt = cfg->dtype;
    DeviceStore& st = h->store;
    OCRVI_TRY(load_conv(st, blob, "stem", 64, 3, 7, 1, AM_ROWS, dt, true, &h->stem));
    int inpl = 64;
    for (int li = 0; li < 4; ++li) {
        const int w = kWidth[li];
        h->layers[li].resize(kBlocks[li]);
        for (int b = 0; b < kBlocks[li]; ++b) {
            Bottleneck& bk = h->layers[li][b];
            const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
            bk.dcn = li >= 1;                          // backbone.py:28-31: DCN in layer2..4
            bk.stride = (b == 0 && li >= 1) ? 2 : 1;   // torchvision Bottleneck v1.5: stride on the 3x3
            bk.has_down = b == 0;
            OCRVI_TRY(load_conv(st, blob, p + ".conv1", w, inpl, 1, 1, AM_CONV1, dt, true, &bk.conv1));
            OCRVI_TRY(load_conv(st, blob, p + ".conv2", w, w, 3, 1, bk.dcn ? AM_DCN : AM_CONV3, dt, true, &bk.conv2));
            if (bk.dcn) OCRVI_TRY(load_conv(st, blob, p + ".conv2.off", 27, w, 3, 1, AM_CONV3, dt, true, &bk.off));
            OCRVI_TRY(load_conv(st, blob, p + ".conv3", 4 * w, w, 1, 1, AM_CONV1, dt, true, &bk.conv3));
            if (bk.has_down) OCRVI_TRY(load_conv(st, blob, p + ".down", 4 * w, inpl, 1, 1, AM_CONV1, dt, true, &bk.down));
            inpl = 4 * w;
        }
    }
    const int cin[4] = {256, 512, 1024, 2048};
    for (int i = 0; i < 4; ++i) {
        OCRVI_TRY(load_conv(st, blob, "neck.lat" + std::to_string(i), 256, cin[i], 1, 1, AM_CONV1, dt, true, &h->lat[i]));
        OCRVI_TRY(load_conv(st, blob, "neck.fpn" + std::to_string(i), 256, 256, 3, 1, AM_CONV3, dt, true, &h->fpn[i]));
    }
    {
        const BlobTensor *w = nullptr, *b = nullptr;
        OCRVI_TRY(blob.get("neck.asf.w", 4, 1024, 0, 0, &w));
        OCRVI_TRY(blob.get("neck.asf.b", 4, 0, 0, 0, &b));
        OCRVI_TRY(st.upload(w->data, 4096 * 4, (void**)&h->asf_w));
        OCRVI_TRY(st.upload(b->data, 16, (void**)&h->asf_b));
    }
    OCRVI_TRY(load_conv(st, blob, "head.conv", 128, 256, 3, 1, AM_CONV3, dt, true, &h->head_conv));
    {   // two ConvTranspose2d(64,64,2,2)+BN+ReLU as ONE grouped pixel-shuffle GEMM (group 0 = binarise, 1 = threshold branch)
        std::vector<float> w2(2 * 64 * 4), b2(2);
        const char* names[2] = {"head.bin", "head.thr"};
        const float *dw[2] = {nullptr, nullptr}, *db[2] = {nullptr, nullptr};
        for (int br = 0; br < 2; ++br) {
            const BlobTensor *w = nullptr, *b = nullptr, *ww = nullptr, *bb = nullptr;
            OCRVI_TRY(blob.get(std::string(names[br]) + ".dc1.w", 64, 64, 2, 2, &w));
            OCRVI_TRY(blob.get(std::string(names[br]) + ".dc1.b", 64, 0, 0, 0, &b));
            dw[br] = w->data;
            db[br] = b->data;
            OCRVI_TRY(blob.get(std::string(names[br]) + ".dc2.w", 64, 1, 2, 2, &ww));
            OCRVI_TRY(blob.get(std::string(names[br]) + ".dc2.b", 1, 0, 0, 0, &bb));
            std::copy(ww->data, ww->data + 256, w2.begin() + br * 256);
            b2[br] = bb->data[0];
        }
        PackedConv both = pack_deconv2(dw, db, 2, 64, 64, dt);
        OCRVI_TRY(upload_packed(st, both, AM_CONV1, &h->head_dc1));
        h->head_dc1.shuffle_co = 64;
        w2.insert(w2.end(), b2.begin(), b2.end());
        OCRVI_TRY(st.upload_f32(w2, &h->dc2_wb));
    }
    {   // scratch pages of the ring GEMM: allocate now so no forward (possibly under graph capture) ever allocates
        const void* z; void* d;
        OCRVI_TRY(ring_pages(&z, &d));
    }
    if (dt == OCRVI_F16X2) OCRVI_TRY(h->range.init());
    *out = h.release();
    return OCRVI_OK;
}

extern "C" void ocrvi_det_destroy(ocrvi_det* h) {
    if (!h) return;
    DeviceGuard dg(h->device);
    delete h;
}

static int check_det_shape(const ocrvi_det* h, int N, int H, int W) {
    OCRVI_CHECK(h, OCRVI_EINVAL, "det: null handle");
    OCRVI_CHECK(N > 0 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, OCRVI_EINVAL,
                "det: input (N=%d,3,%d,%d) needs H and W to be multiples of 32", N, H, W);
    // the stem's GEMM has N*(H/2)*(W/2) rows and the conv kernels index rows with 23 bits (magic-number division)
    OCRVI_CHECK((size_t)N * (H / 2) * (W / 2) < ((size_t)1 << 23), OCRVI_EINVAL,
                "det: batch of %d %dx%d images too large for one call (at most %zu pixels per call: chunk it)", N, H, W, ((size_t)1 << 25) - 1);
    return OCRVI_OK;
}

static int det_run(ocrvi_det* h, Runner& r, const float* x, int N, int H, int W, float* binary, float* thresh, float* tbin, float* blog,
                   float* tlog) {
    const int dt = h->cfg.dtype;
    const int Hp = H + 6, Wp = W + 8;
    Tensor xpad = r.alloc(N, Hp, Wp, 4);
    if (!r.dry()) OCRVI_TRY(k_nchw3_to_nhwc4_pad(dt, x, xpad.p, N, H, W, 3, 3, Hp, Wp, r.stream));
    // ---- stem: conv 7x7/2 + BN + ReLU, maxpool 3x3/2 (torchvision resnet50 via backbone.py:34)
    Tensor cur;
    if (stem_pool_eligible(dt, h->stem.N_g, h->stem.Kp, h->stem.KH, H, W)) {
        // f16x2: one kernel; the half-resolution 64-channel map (the detector's largest tensor) is never written (stem_pool.hip)
        cur = r.alloc(N, H / 4, W / 4, 64);
        if (!r.dry()) OCRVI_TRY(k_stem_pool(dt, xpad.p, h->stem.w, h->stem.bias, h->stem.wscale, cur.p, N, H, W, Hp, Wp, r.stream));
    } else {
        Tensor s = r.alloc(N, H / 2, W / 2, 64);
        {
            ConvOpts o;
            o.sh = o.sw = 2; o.pad = 3; o.act = ACT_RELU; o.Hp = Hp; o.Wp = Wp;
            OCRVI_TRY(conv(r, h->stem, xpad, s, o));
        }
        cur = r.alloc(N, H / 4, W / 4, 64);
        if (!r.dry()) OCRVI_TRY(k_maxpool3x3s2(dt, s.p, cur.p, N, H / 2, W / 2, 64, r.stream));
    }
    // ---- layers 1..4.  Block outputs ping-pong between two slots per layer (sized for the layer's output); the last block of a
    // layer writes a fresh buffer that stays live as the c2..c5 tap (backbone.py:56-60).  Temporaries of a block are released
    // when it ends, so a chunk of 16 pages keeps ~4 GB less live than with one buffer per block.
    Tensor feats[4];
    for (int li = 0; li < 4; ++li) {
        const int w = kWidth[li];
        const int nb = (int)h->layers[li].size();
        const int lh = cur.h / h->layers[li][0].stride, lw = cur.w / h->layers[li][0].stride;
        Tensor tap = r.alloc(N, lh, lw, 4 * w);                       // the layer's final output
        const size_t slot_mark = r.arena.mark();
        Tensor slot[2] = {r.alloc(N, lh, lw, 4 * w), r.alloc(N, lh, lw, 4 * w)};
        for (int b = 0; b < nb; ++b) {
            const Bottleneck& bk = h->layers[li][b];
            const int oh = cur.h / bk.stride, ow = cur.w / bk.stride;
            Tensor y = b == nb - 1 ? tap : slot[b & 1];               // never the buffer `cur` lives in (slot[(b-1)&1] or the previous tap)
            const size_t mark2 = r.arena.mark();
            Tensor t1 = r.alloc(N, cur.h, cur.w, w);
            Tensor t2 = r.alloc(N, oh, ow, w);
            {
                ConvOpts o;
                o.act = ACT_RELU;
                OCRVI_TRY(conv(r, bk.conv1, cur, t1, o));
            }
            if (bk.dcn) {
                // DeformableConv2d.forward (dcn.py:41-59): 27-ch conv -> offsets (ch 0..17) + sigmoid mask (ch 18..26) -> deform_conv2d
                float* offs = (float*)r.arena.alloc((size_t)N * oh * ow * 32 * 4);
                Tensor ot; ot.p = offs; ot.n = N; ot.h = oh; ot.w = ow; ot.c = 32; ot.f32 = true;
                ConvOpts oo;
                oo.sh = oo.sw = bk.stride; oo.pad = 1; oo.store_mode = ST_DCN_OFFS;
                OCRVI_TRY(conv(r, bk.off, t1, ot, oo));
                ConvOpts o;
                o.sh = o.sw = bk.stride; o.pad = 1; o.act = ACT_RELU; o.offs = offs;
                OCRVI_TRY(conv(r, bk.conv2, t1, t2, o));
            } else {
                ConvOpts o;
                o.sh = o.sw = bk.stride; o.pad = 1; o.act = ACT_RELU;
                OCRVI_TRY(conv(r, bk.conv2, t1, t2, o));
            }
            Tensor idn = cur;
            if (bk.has_down) {
                idn = r.alloc(N, oh, ow, 4 * w);
                ConvOpts o;
                o.sh = o.sw = bk.stride;
                OCRVI_TRY(conv(r, bk.down, cur, idn, o));
            }
            {
                ConvOpts o;
                o.act = ACT_RELU; o.res = &idn; o.res_mode = RES_SAME;
                OCRVI_TRY(conv(r, bk.conv3, t2, y, o));
            }
            r.arena.release(mark2);  // t1, t2, offsets, downsample are dead
            cur = y;
        }
        r.arena.release(slot_mark);  // both slots are dead once the layer's last block has written the tap
        feats[li] = cur;
        h->tap_c[li] = cur;
    }
    // ---- FPN top-down (neck.py:26-41)
    Tensor inner[4], p[4];
    for (int i = 3; i >= 0; --i) {
        inner[i] = r.alloc(N, feats[i].h, feats[i].w, 256);
        ConvOpts o;
        if (i < 3) { o.res = &inner[i + 1]; o.res_mode = RES_UP2; }  // lateral + nearest-2x(last_inner)
        OCRVI_TRY(conv(r, h->lat[i], feats[i], inner[i], o));
        p[i] = r.alloc(N, feats[i].h, feats[i].w, 256);
        ConvOpts o3;
        o3.pad = 1; o3.act = ACT_RELU;
        OCRVI_TRY(conv(r, h->fpn[i], inner[i], p[i], o3));
    }
    // ---- adaptive scale fusion (neck.py:57-79)
    Tensor fused = r.alloc(N, p[0].h, p[0].w, 256);
    float* asf_scratch = (float*)r.arena.alloc(asf_scratch_bytes(N, p[0].h, p[0].w));
    if (!r.dry()) OCRVI_TRY(k_asf(dt, p[0].p, p[1].p, p[2].p, p[3].p, h->asf_w, h->asf_b, asf_scratch, fused.p, N, p[0].h, p[0].w, r.stream));
    h->tap_fused = fused;
    // ---- DB head (head.py:32-48): both branches' 3x3 convs as one 256->128 conv, both deconv1 as one grouped pixel-shuffle GEMM
    Tensor hc = r.alloc(N, fused.h, fused.w, 128);
    {
        ConvOpts o;
        o.pad = 1; o.act = ACT_RELU;
        OCRVI_TRY(conv(r, h->head_conv, fused, hc, o));
    }
    // deconv1 (+BN+ReLU) and deconv2 (64 -> 1) of both branches in ONE GEMM: the 1.26 GB deconv1 activation is never materialised
    const size_t map_elems = (size_t)N * H * W;
    float* bl = blog ? blog : (float*)r.arena.alloc(map_elems * 4);
    float* tl = tlog ? tlog : (float*)r.arena.alloc(map_elems * 4);
    {
        Tensor lm; lm.p = bl; lm.n = N; lm.h = H; lm.w = W; lm.c = 1; lm.f32 = true;
        ConvOpts o;
        o.store_mode = ST_DB_TAIL; o.out2 = tl; o.offs = h->dc2_wb;
        OCRVI_TRY(conv(r, h->head_dc1, hc, lm, o));
    }
    if (!r.dry()) OCRVI_TRY(k_db_maps(bl, tl, h->cfg.k, binary, thresh, tbin, map_elems, r.stream));
    return OCRVI_OK;
}

extern "C" int ocrvi_det_workspace_bytes(const ocrvi_det* h, int N, int H, int W, size_t* bytes) {
    OCRVI_CHECK(bytes, OCRVI_EINVAL, "det_workspace_bytes: null out");
    OCRVI_TRY(check_det_shape(h, N, H, W));
    Runner r(h->cfg.dtype, nullptr, nullptr, 0);
    OCRVI_TRY(det_run(const_cast<ocrvi_det*>(h), r, nullptr, N, H, W, nullptr, nullptr, nullptr, nullptr, nullptr));
    *bytes = r.arena.peak + 256;
    return OCRVI_OK;
}

extern "C" int ocrvi_det_forward(ocrvi_det* h, const float* x, int N, int H, int W, float* binary, float* thresh, float* thresh_binary,
                                 float* bin_logits, float* thresh_logits, void* workspace, size_t workspace_bytes, void* stream) {
    OCRVI_TRY(check_det_shape(h, N, H, W));
    OCRVI_CHECK(x && binary && workspace, OCRVI_EINVAL, "det_forward: x, binary and workspace are required");
    DeviceGuard dg(h->device);  // launches go to the handle's device whatever the caller's current device is
    OCRVI_HIP(dg.err);
    size_t need = 0;
    OCRVI_TRY(ocrvi_det_workspace_bytes(h, N, H, W, &need));
    OCRVI_CHECK(workspace_bytes >= need, OCRVI_ENOMEM, "det_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    OCRVI_CHECK(((uintptr_t)workspace & 255) == 0, OCRVI_EINVAL, "det_forward: workspace must be 256-byte aligned");
    Runner r(h->cfg.dtype, (hipStream_t)stream, workspace, workspace_bytes);
    OCRVI_TRY(det_run(h, r, x, N, H, W, binary, thresh, thresh_binary, bin_logits, thresh_logits));
    OCRVI_CHECK(!r.arena.overflow, OCRVI_ENOMEM, "det_forward: workspace overflow");
    return h->range.snapshot((hipStream_t)stream);
}

extern "C" int ocrvi_det_status(const ocrvi_det* h) {
    OCRVI_CHECK(h, OCRVI_EINVAL, "det_status: null handle");
    return h->range.status("det");
}

extern "C" int ocrvi_det_debug_features(ocrvi_det* h, int N, int H, int W, float* c2, float* c3, float* c4, float* c5, float* fused,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    OCRVI_TRY(check_det_shape(h, N, H, W));
    OCRVI_CHECK(h->tap_fused.p, OCRVI_EINVAL, "det_debug_features: no forward has run");
    DeviceGuard dg(h->device);
    OCRVI_HIP(dg.err);
    (void)workspace; (void)workspace_bytes;
    float* outs[4] = {c2, c3, c4, c5};
    for (int i = 0; i < 4; ++i) {
        const Tensor& t = h->tap_c[i];
        if (outs[i]) OCRVI_TRY(k_nhwc_to_nchw_f32(h->cfg.dtype, t.p, outs[i], t.n, t.h, t.w, t.c, t.c, 0, (hipStream_t)stream));
    }
    const Tensor& f = h->tap_fused;
    if (fused) OCRVI_TRY(k_nhwc_to_nchw_f32(h->cfg.dtype, f.p, fused, f.n, f.h, f.w, f.c, f.c, 0, (hipStream_t)stream));
    return OCRVI_OK;
}
