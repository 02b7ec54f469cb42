// Implicit-GEMM convolution / linear / deformable-conv kernel for gfx950 (MFMA 16x16, wave64, LDS-tiled).
//
//   out[m, n] = act( sum_k A[m, k] * Wt[n, k] + bias[n] (+ res) ),   m = (img, oh, ow),  n = out channel
//
// Activations are NHWC so a K-chunk of one filter tap is a contiguous run of channels.  Both operands are
// staged as [rows][128 B] LDS tiles (XOR-swizzled, conflict-free ds_read_b128) and one K-step is 128 bytes
// of K (32 fp32 or 64 bf16/fp16 elements).  The MFMA "A" operand is the weight tile and "B" the pixel tile,
// so each lane's 4 accumulator registers are 4 consecutive output channels of one pixel (vector stores).
//
// A-operand producers (template AMODE):
//   AM_CONV1 / AM_CONV3  plain 1x1 / 3x3 conv (any stride / pad, groups); nn.Linear is AM_CONV1 with H=W=1.
//   AM_ROWS              stem convs on a zero-padded NHWC4 image: one K row-chunk = 8 consecutive pixels x 4 ch
//                        of one filter row (7x7/2: backbone.py:34 via torchvision; 3x3/2: svtrv2.py:110).
//   AM_DCN               modulated deformable 3x3 (torchvision.ops.deform_conv2d as called in dcn.py:48-57):
//                        4-corner bilinear gather of NHWC channel vectors, blended in fp32, written to LDS.
#pragma once
#include "common.h"

namespace ocrvi {

enum { AM_CONV1 = 0, AM_CONV3 = 1, AM_ROWS = 2, AM_DCN = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { RES_NONE = 0, RES_SAME = 1, RES_UP2 = 2 };        // RES_UP2: nearest 2x upsample of a half-res tensor (neck.py:36-38)
enum { ST_NHWC = 0, ST_SHUFFLE2 = 1, ST_DCN_OFFS = 2 };  // ST_SHUFFLE2: ConvTranspose2d(k=2,s=2) pixel shuffle (head.py:13,16)

struct ConvParams {
    const void* x = nullptr;      // T   [n_img][H][W][Cin]   (AM_ROWS: [n_img][Hp][Wp][4])
    const void* w = nullptr;      // T   [groups][Np][Kp]
    const float* bias = nullptr;  // f32 [groups*N_g] or null
    void* out = nullptr;          // T or f32
    const void* res = nullptr;    // T or f32, or null
    const float* offs = nullptr;  // AM_DCN: f32 [M][32] = 18 offsets (dy,dx per tap), 9 sigmoided masks, 5 pad
    int n_img = 1, H = 1, W = 1, Cin = 0;
    int OH = 1, OW = 1;
    int KH = 1, SH = 1, SW = 1, PH = 0, PW = 0;
    int Cin_g = 0, cin_off = 0;
    int N_g = 0, Np = 0, Kp = 0, groups = 1;
    int M = 0;
    int ldo = 0, out_coff = 0, ldr = 0;
    int act = ACT_NONE, res_mode = RES_NONE, store_mode = ST_NHWC, out_f32 = 0, res_f32 = 0;
    int res_post = 0;  // 0: act(acc + bias + res) (ResNet);  1: act(acc + bias) + res (x + mixer(..), svtrv2.py:98-101)
    int Hp = 0, Wp = 0;
    int shuffle_co = 0;  // ST_SHUFFLE2: channels per output pixel (n = (a*2+b)*shuffle_co + co)
};

template <typename T, int AMODE, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
    constexpr int EPC = TypeInfo<T>::EPC;  // elements per 16-byte chunk
    constexpr int BKE = 8 * EPC;           // elements per K-step (128 bytes)
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int APASS = BM / 32, BPASS = BN / 32;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto As = [&](int buf) -> char* { return smem + buf * (BM + BN) * 128; };
    auto Bs = [&](int buf) -> char* { return smem + buf * (BM + BN) * 128 + BM * 128; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = p.Np / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = tile % ntiles, mt = tile / ntiles;
    const int grp = blockIdx.y;
    const int m0 = mt * BM, n0 = nt * BN;
    const T* __restrict__ X = (const T*)p.x;
    const T* __restrict__ Wt = (const T*)p.w + (size_t)grp * p.Np * p.Kp;

    // ---- per-thread staging geometry: rows (tid>>3)+32*i, 16-byte chunk j = tid&7
    const int lrow = tid >> 3, j = tid & 7;
    int a_pix[APASS], a_ih0[APASS], a_iw0[APASS];
    size_t a_base[APASS];
    bool a_ok[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int ow = mm % p.OW, t = mm / p.OW, oh = t % p.OH, img = t / p.OH;
        a_ih0[i] = oh * p.SH - p.PH;
        a_iw0[i] = ow * p.SW - p.PW;
        a_pix[i] = img * p.H * p.W;
        if constexpr (AMODE == AM_ROWS) a_base[i] = ((size_t)(img * p.Hp + oh * p.SH) * p.Wp + ow * p.SW) * 4;
        if constexpr (AMODE == AM_DCN) a_base[i] = (size_t)mm * 32;
    }
    const int cbase = p.cin_off + grp * p.Cin_g;

    struct Stage {
        uint4 a[APASS];
        uint4 b[BPASS];
        uint4 c[AMODE == AM_DCN ? APASS : 1][4];
        float cw[AMODE == AM_DCN ? APASS : 1][4];
    } st;

    auto issue = [&](int ks) {
        // weights: rows lrow+32*i of the N tile, always in range (padded)
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            st.b[i] = *(const uint4*)(Wt + (size_t)(n0 + lrow + 32 * i) * p.Kp + (size_t)ks * BKE + j * EPC);
        const int koff = ks * BKE + j * EPC;
        if constexpr (AMODE == AM_CONV1 || AMODE == AM_CONV3) {
            constexpr int KS = AMODE == AM_CONV1 ? 1 : 3;
            const int tap = koff / p.Cin_g, c = koff - tap * p.Cin_g;
            const int r = tap / KS, s = tap - r * KS;
            const bool tap_ok = tap < KS * KS;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int ih = a_ih0[i] + r, iw = a_iw0[i] + s;
                const bool ok = a_ok[i] && tap_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *(const uint4*)(X + (size_t)(a_pix[i] + ih * p.W + iw) * p.Cin + cbase + c);
                st.a[i] = v;
            }
        } else if constexpr (AMODE == AM_ROWS) {
            const int r = koff >> 5, col = koff & 31;
            const bool ok_r = r < p.KH;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a_ok[i] && ok_r) v = *(const uint4*)(X + a_base[i] + (size_t)r * p.Wp * 4 + col);
                st.a[i] = v;
            }
        } else {  // AM_DCN: tap is uniform over the K-step (Cin_g % BKE == 0)
            const int tap = (ks * BKE) / p.Cin_g, c = koff - tap * p.Cin_g;
            const int r = tap / 3, s = tap - r * 3;
            const bool tap_ok = tap < 9;
            const int tp = tap_ok ? tap : 0;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const float* o = p.offs + a_base[i];
                const float dy = o[2 * tp], dx = o[2 * tp + 1], mk = o[18 + tp];
                const float py = (float)(a_ih0[i] + r) + dy, px = (float)(a_iw0[i] + s) + dx;
                const bool inside = a_ok[i] && tap_ok && py > -1.f && py < (float)p.H && px > -1.f && px < (float)p.W;
                // clamp before float->int so wild / NaN offsets cannot overflow (their weight is already 0)
                const float cy = fminf(fmaxf(py, -2.f), (float)p.H + 1.f), cx = fminf(fmaxf(px, -2.f), (float)p.W + 1.f);
                const float fy = floorf(cy), fx = floorf(cx);
                const float ly = cy - fy, lx = cx - fx, hy = 1.f - ly, hx = 1.f - lx;
                const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
                const bool oy0 = y0 >= 0, oy1 = y1 <= p.H - 1, ox0 = x0 >= 0, ox1 = x1 <= p.W - 1;
                const float m = inside ? mk : 0.f;
                st.cw[i][0] = (oy0 && ox0) ? hy * hx * m : 0.f;
                st.cw[i][1] = (oy0 && ox1) ? hy * lx * m : 0.f;
                st.cw[i][2] = (oy1 && ox0) ? ly * hx * m : 0.f;
                st.cw[i][3] = (oy1 && ox1) ? ly * lx * m : 0.f;
                const int yc0 = min(max(y0, 0), p.H - 1), yc1 = min(max(y1, 0), p.H - 1);
                const int xc0 = min(max(x0, 0), p.W - 1), xc1 = min(max(x1, 0), p.W - 1);
                const T* base = X + (size_t)a_pix[i] * p.Cin + cbase + c;
                st.c[i][0] = *(const uint4*)(base + (size_t)(yc0 * p.W + xc0) * p.Cin);
                st.c[i][1] = *(const uint4*)(base + (size_t)(yc0 * p.W + xc1) * p.Cin);
                st.c[i][2] = *(const uint4*)(base + (size_t)(yc1 * p.W + xc0) * p.Cin);
                st.c[i][3] = *(const uint4*)(base + (size_t)(yc1 * p.W + xc1) * p.Cin);
            }
        }
    };

    auto commit = [&](int buf) {
        if constexpr (AMODE == AM_DCN) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float acc[EPC], f[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    Chunk<T>::unpack(st.c[i][q], f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[e] = fmaf(st.cw[i][q], f[e], acc[e]);
                }
                st.a[i] = Chunk<T>::pack(acc);
            }
        }
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int row = lrow + 32 * i;
            *(uint4*)(As(buf) + row * 128 + ((j ^ swz128(row)) << 4)) = st.a[i];
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int row = lrow + 32 * i;
            *(uint4*)(Bs(buf) + row * 128 + ((j ^ swz128(row)) << 4)) = st.b[i];
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lr = lane & 15, g = lane >> 4;
    const int sw = swz128(lr);
    const int fo0 = ((2 * g) ^ sw) << 4, fo1 = ((2 * g + 1) ^ sw) << 4;

    const int nk = p.Kp / BKE;
    issue(0);
    commit(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) issue(ks + 1);
        uint4 xf[MI][2], wf[NI][2];
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const char* r = As(cur) + (wm * TM + b * 16 + lr) * 128;
            xf[b][0] = *(const uint4*)(r + fo0);
            xf[b][1] = *(const uint4*)(r + fo1);
        }
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const char* r = Bs(cur) + (wn * TN + a * 16 + lr) * 128;
            wf[a][0] = *(const uint4*)(r + fo0);
            wf[a][1] = *(const uint4*)(r + fo1);
        }
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) Mma<T>::run(wf[a], xf[b], acc[a][b]);
        if (ks + 1 < nk) commit(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels n..n+3 (n = .. + 4g) of pixel m (= .. + lr)
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = m0 + wm * TM + b * 16 + lr;
        if (m >= p.M) continue;
        size_t orow, rrow = 0;
        if (p.store_mode == ST_SHUFFLE2 || p.res_mode == RES_UP2) {
            const int ow = m % p.OW, t = m / p.OW, oh = t % p.OH, img = t / p.OH;
            orow = (size_t)m;
            if (p.res_mode == RES_UP2) rrow = ((size_t)img * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1);
            if (p.store_mode == ST_SHUFFLE2) orow = ((size_t)img * (2 * p.OH) + 2 * oh) * (2 * p.OW) + 2 * ow;
        } else {
            orow = (size_t)m;
        }
        if (p.res_mode == RES_SAME) rrow = (size_t)m;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = n0 + wn * TN + a * 16 + 4 * g;  // within group
            if (n >= p.N_g) continue;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            const int nglob = grp * p.N_g + n;
            if (p.store_mode == ST_DCN_OFFS) {
                float* o = (float*)p.out + orow * 32 + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nn = n + r;
                    float t = nn < p.N_g ? v[r] + p.bias[nn] : 0.f;
                    if (nn >= 18) t = nn < p.N_g ? 1.0f / (1.0f + expf(-t)) : 0.f;  // mask = sigmoid (dcn.py:46)
                    o[r] = t;
                }
                continue;
            }
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += p.bias[nglob + r];
            }
            if (p.res_post) {
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (p.act == ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
            }
            if (p.res_mode != RES_NONE) {
                const size_t ro = rrow * p.ldr + nglob;
                if (p.res_f32) {
                    const float4 rv = *(const float4*)((const float*)p.res + ro);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
                    const T* rp = (const T*)p.res + ro;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32<T>(rp[r]);
                }
            }
            if (!p.res_post) {
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (p.act == ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                }
            }
            size_t oo;
            if (p.store_mode == ST_SHUFFLE2) {
                const int ab = n / p.shuffle_co, co = n - ab * p.shuffle_co;
                oo = (orow + (size_t)(ab >> 1) * (2 * p.OW) + (ab & 1)) * p.ldo + p.out_coff + grp * p.shuffle_co + co;
            } else {
                oo = orow * p.ldo + p.out_coff + nglob;
            }
            if (p.out_f32) {
                *(float4*)((float*)p.out + oo) = make_float4(v[0], v[1], v[2], v[3]);
            } else if constexpr (sizeof(T) == 4) {
                *(float4*)((float*)p.out + oo) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                union { T h[4]; uint2 u; } pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk.h[r] = from_f32<T>(v[r]);
                *(uint2*)((T*)p.out + oo) = pk.u;
            }
        }
    }
}

// Tile choice shared by the packer (host) and the launcher.
static inline int conv_bn_for(int n_g) { return n_g > 64 ? 128 : (n_g > 32 ? 64 : 32); }
static inline int conv_bke(int dtype) { return dtype == OCRVI_F32 ? 32 : 64; }

template <typename T> int launch_conv(const ConvParams& p, int amode, hipStream_t stream);
int launch_conv_dt(int dtype, const ConvParams& p, int amode, hipStream_t stream);

// Host-side packing into [groups][Np][Kp] (dtype T bytes), K order = (tap, cin) / ROWS / deconv.
struct PackedConv {
    std::vector<char> bytes;
    std::vector<float> bias;  // expanded (ST_SHUFFLE2: 4x)
    int Np = 0, Kp = 0, N_g = 0, Cin_g = 0, groups = 1, KH = 1;
};
// w: [Cout][Cin_g][KH][KW] fp32 (conv) -- amode AM_CONV1/AM_CONV3/AM_DCN/AM_ROWS
PackedConv pack_conv(const float* w, const float* bias, int cout, int cin_g, int kh, int kw, int groups, int amode, int dtype);
// ConvTranspose2d(k=2,s=2) weight [Cin][Cout][2][2] -> GEMM [n=(a*2+b)*Cout+co][k=ci]
PackedConv pack_deconv2(const float* w, const float* bias, int cin, int cout, int dtype);

}  // namespace ocrvi
