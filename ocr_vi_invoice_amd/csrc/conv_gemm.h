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
#include <stdlib.h>

#include "common.h"

namespace ocrvi {

enum { AM_CONV1 = 0, AM_CONV3 = 1, AM_ROWS = 2, AM_DCN = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
enum { RES_NONE = 0, RES_SAME = 1, RES_UP2 = 2 };        // RES_UP2: nearest 2x upsample of a half-res tensor (neck.py:36-38)
enum { ST_NHWC = 0, ST_SHUFFLE2 = 1, ST_DCN_OFFS = 2, ST_DB_TAIL = 3 };  // ST_SHUFFLE2: ConvTranspose2d(k=2,s=2) pixel shuffle (head.py:13,16)
// ST_DB_TAIL: ST_SHUFFLE2 (64 ch, +bias, ReLU) followed IN THE EPILOGUE by the next ConvTranspose2d(64,1,2,2) (head.py:16): the wave's 64
// columns are all channels of one sub-pixel, so the 64->1 deconv is an in-register dot + butterfly; writes fp32 logit maps [n,1,4*OH,4*OW].

struct ConvParams {
    const void* x = nullptr;      // T   [n_img][H][W][Cin]   (AM_ROWS: [n_img][Hp][Wp][4])
    const void* w = nullptr;      // T   [groups][Np][Kp]
    const float* bias = nullptr;  // f32 [groups*N_g] or null
    void* out = nullptr;          // T or f32
    void* out2 = nullptr;         // ST_DB_TAIL: logit map of group 1 (threshold branch); `out` is group 0's
    const void* zero_page = nullptr;  // gemm_ring: >= 128 B of zeros (source of A rows past M)
    void* dump_page = nullptr;        // gemm_ring: >= 1 KiB sink for the stores of out-of-range lanes
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
    unsigned long long mg_ow = 0, mg_oh = 0;  // floor(2^40/d)+1: n/d == (n*mg)>>40 for n < 2^23 (filled by launch_conv)
    int identity_pix = 0;  // 1x1, stride 1, pad 0: input pixel index == m (no decomposition needed)
    int epi_lds = 0;       // stage the output tile through LDS and store whole rows (ST_NHWC only; set by launch_conv)
    int res_in_store = 0;  // fp32 out + fp32 residual, no activation: add the residual in the coalesced store phase
    int patch_lw = 7;      // dcn_pipe: a tile is a (128 >> patch_lw) x (1 << patch_lw) patch of output pixels
    unsigned out_bytes = 0;   // gemm_ring: bytes of the output tensor the stores may touch (buffer descriptor range; filled by launch_gemm_ring)
    int nt_out = 0;        // gemm_ring / gemm_duo: non-temporal output stores (set by the launcher for N >= 2 K)
    float wscale = 1.f;    // f16x2: the weights are stored multiplied by 2^s (one power of two per layer, chosen by the packer so that their
                           // lo halves are normal fp16 numbers); every epilogue multiplies the accumulator by wscale = 2^-s (exact)
};

// accumulator -> pre-bias value: the f16x2 weight scale (a no-op for the other types)
template <typename T> __device__ __forceinline__ float unscale(float acc, float wscale) {
    if constexpr (IsSplit<T>::value) return acc * wscale; else return acc;
}

__device__ __forceinline__ int fastdiv(int n, unsigned long long mg) { return (int)(((unsigned long long)(unsigned)n * mg) >> 40); }

// Resident workgroups per CU the launcher sizes the persistent grid for (bounded by VGPRs: 168 / 112 / 88 per lane).
template <int AMODE, int BM, int BN> struct ConvOcc { static constexpr int value = AMODE == AM_DCN ? 2 : (BN >= 128 ? 3 : (BN >= 64 ? 4 : 5)); };
// ... per operand type: the f16x2 build keeps a half-swapped copy of every weight fragment (Mma<f16x2_t>), which does not fit 168 VGPRs at
// 128 x 128 (37 spilled dwords measured): two workgroups per CU there
// the 16-bit builds of the 128 x 128 tile spilled 10-12 VGPRs at three workgroups per CU (168 registers): two per CU as well
template <typename T, int AMODE, int BM, int BN> struct ConvOccT {
    static constexpr int value = (!IsF32<T>::value && BN >= 128 && AMODE != AM_DCN) ? 2 : ConvOcc<AMODE, BM, BN>::value;
};

// Persistent implicit-GEMM kernel.  A workgroup (4 waves) walks output tiles  first, first+G, first+2G, ...  One LDS stage
// (A tile + W tile, 128 B of K per row); global loads for the NEXT K-step -- or the next TILE's first K-step -- are issued
// into registers before the MFMAs of the current one and written to LDS after them, so the load latency, the MFMA phase, the
// epilogue and the store drain of consecutive tiles overlap (most layers here have only 1..8 K-steps per tile, so a
// one-tile-per-workgroup kernel runs load -> MFMA -> store strictly in series).
// PERSIST = false (shipped): one tile per workgroup.  Measured on MI355X (profiles/r01_conv_variants.md): holding the next
// tile's loads across the epilogue costs ~40 VGPRs, i.e. one resident workgroup per CU, and loses 10-35 % on every shape;
// occupancy (3-5 workgroups per CU) hides more latency than cross-tile prefetch does.
// PERSIST: 0 = one tile per workgroup; 1 = persistent with cross-tile prefetch (next tile's loads issued before this tile's
// epilogue; +~40 VGPRs); 2 = persistent, tiles strictly one after another (no extra registers; tile i's stores drain under
// tile i+1's loads).
template <typename T, int AMODE, int BM, int BN, int WM, int WN, int PERSIST>
__global__ __launch_bounds__(256, (ConvOccT<T, AMODE, BM, BN>::value)) void conv_gemm_kernel(const ConvParams p) {
    constexpr int EPC = TypeInfo<T>::EPC;  // elements per 16-byte chunk
    constexpr int BKE = 8 * EPC;           // elements per K-step (128 bytes)
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int APASS = BM / 32, BPASS = BN / 32;
    constexpr int LDS_BYTES = (BM + BN) * 128;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;
    char* const Bs = smem + BM * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = p.Np / BN;
    const int total = ((p.M + BM - 1) / BM) * ntiles;
    const int G = gridDim.x;
    const int grp = blockIdx.y;
    const T* __restrict__ X = (const T*)p.x;
    const T* __restrict__ Wt = (const T*)p.w + (size_t)grp * p.Np * p.Kp;
    const int cbase = p.cin_off + grp * p.Cin_g;
    const int lrow = tid >> 3, j = tid & 7;  // staging geometry: rows lrow+32*i, 16-byte chunk j
    const int lr = lane & 15, g = lane >> 4;
    const int sw = swz128(lr);
    const int fo0 = ((2 * g) ^ sw) << 4, fo1 = ((2 * g + 1) ^ sw) << 4;
    const int nk = p.Kp / BKE;

    int m0 = 0, n0 = 0;
    int a_pix[APASS], a_ih0[APASS], a_iw0[APASS];
    size_t a_base[APASS];
    bool a_ok[APASS];

    auto setup = [&](int tile) {  // n-tile fastest: neighbouring workgroups share the A rows through L2
        const int mt = tile / ntiles;
        n0 = (tile - mt * ntiles) * BN;
        m0 = mt * BM;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int m = m0 + lrow + 32 * i;
            a_ok[i] = m < p.M;
            const int mm = a_ok[i] ? m : 0;
            int ow = 0, oh = 0, img = 0;
            if (AMODE == AM_CONV1 && p.identity_pix) {
                a_pix[i] = mm;  // input pixel == output pixel
            } else {
                const int t = fastdiv(mm, p.mg_ow);
                ow = mm - t * p.OW;
                img = fastdiv(t, p.mg_oh);
                oh = t - img * p.OH;
                a_pix[i] = img * p.H * p.W;
            }
            a_ih0[i] = oh * p.SH - p.PH;
            a_iw0[i] = ow * p.SW - p.PW;
            if constexpr (AMODE == AM_ROWS) a_base[i] = ((size_t)(img * p.Hp + oh * p.SH) * p.Wp + ow * p.SW) * 4;
            if constexpr (AMODE == AM_DCN) a_base[i] = (size_t)mm * 32;
        }
    };

    struct Stage {
        uint4 a[APASS];
        uint4 b[BPASS];
        uint4 c[AMODE == AM_DCN ? APASS : 1][4];
    } st;
    // AM_DCN: sampling geometry of the current tap, reused by every K-step of that tap (Cin_g / BKE of them)
    float dcw[AMODE == AM_DCN ? APASS : 1][4];   // bilinear weight x mask per corner (0 when the corner is outside)
    int dco[AMODE == AM_DCN ? APASS : 1][4];     // clamped corner pixel offset inside the image
    int dc_tap = -1;

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
                if (ok) v = *(const uint4*)(X + (size_t)(a_pix[i] + ih * p.W + iw) * p.Cin + cbase + c);  // identity_pix: ih = iw = 0
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
            if (tap != dc_tap) {  // workgroup-uniform: first K-step of a new tap
                dc_tap = tap;
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
                    dcw[i][0] = (oy0 && ox0) ? hy * hx * m : 0.f;
                    dcw[i][1] = (oy0 && ox1) ? hy * lx * m : 0.f;
                    dcw[i][2] = (oy1 && ox0) ? ly * hx * m : 0.f;
                    dcw[i][3] = (oy1 && ox1) ? ly * lx * m : 0.f;
                    const int yc0 = min(max(y0, 0), p.H - 1), yc1 = min(max(y1, 0), p.H - 1);
                    const int xc0 = min(max(x0, 0), p.W - 1), xc1 = min(max(x1, 0), p.W - 1);
                    dco[i][0] = a_pix[i] + yc0 * p.W + xc0;
                    dco[i][1] = a_pix[i] + yc0 * p.W + xc1;
                    dco[i][2] = a_pix[i] + yc1 * p.W + xc0;
                    dco[i][3] = a_pix[i] + yc1 * p.W + xc1;
                }
            }
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    st.c[i][q] = *(const uint4*)(X + (size_t)dco[i][q] * p.Cin + cbase + c);
            }
        }
    };

    auto commit = [&]() {
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
                    for (int e = 0; e < EPC; ++e) acc[e] = fmaf(dcw[i][q], f[e], acc[e]);
                }
                st.a[i] = Chunk<T>::pack(acc);
            }
        }
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int row = lrow + 32 * i;
            *(uint4*)(As + row * 128 + ((j ^ swz128(row)) << 4)) = st.a[i];
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int row = lrow + 32 * i;
            *(uint4*)(Bs + row * 128 + ((j ^ swz128(row)) << 4)) = st.b[i];
        }
    };

    f32x4 acc[NI][MI];

    auto activate = [&](float (&v)[4]) {
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
    };

    // ---- epilogue of the tile at (em0, en0): lane holds channels n..n+3 (n = .. + 4g) of pixel m (= .. + lr)
    auto epilogue = [&](int em0, int en0) {
        if (p.epi_lds) {
            // Coalesced path: bias / residual / activation in registers, tile -> LDS ([rows][BN] of the output type, 16-byte chunks
            // XOR-swizzled by row), then every thread stores 16-byte chunks so a row segment leaves as one contiguous run.
            const int osz = (p.out_f32 || sizeof(T) == 4) ? 4 : 2;
            const int row_bytes = BN * osz;                    // 64..512, a power of two
            const int cpr = row_bytes >> 4;                    // 16-byte chunks per row
            const int passes = (BM * row_bytes + LDS_BYTES - 1) / LDS_BYTES;   // 1 or 2 (2: fp32 output of a 128-wide tile)
            const int rows_pp = BM / passes;
            for (int ps = 0; ps < passes; ++ps) {
                __syncthreads();  // main loop (or previous pass) is done with the LDS
                if ((wm * TM) / rows_pp == ps) {
#pragma unroll
                    for (int b = 0; b < MI; ++b) {
                        const int rl = wm * TM + b * 16 + lr - ps * rows_pp;  // row inside this pass
                        const int m = em0 + wm * TM + b * 16 + lr;
                        const size_t rrow = (size_t)(m < p.M ? m : 0);
#pragma unroll
                        for (int a = 0; a < NI; ++a) {
                            const int nl = wn * TN + a * 16 + 4 * g;  // column inside the tile
                            const int n = en0 + nl;
                            float v[4] = {unscale<T>(acc[a][b][0], p.wscale), unscale<T>(acc[a][b][1], p.wscale), unscale<T>(acc[a][b][2], p.wscale),
                                          unscale<T>(acc[a][b][3], p.wscale)};
                            const int nglob = grp * p.N_g + (n < p.N_g ? n : 0);
                            if (p.bias) {
                                const float4 bv = *(const float4*)(p.bias + nglob);
                                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                            }
                            if (p.res_post) activate(v);
                            if (p.res_mode == RES_SAME && !p.res_in_store && m < p.M && n < p.N_g) {
                                const size_t ro = rrow * p.ldr + nglob;
                                if (p.res_f32) {
                                    const float4 rv = *(const float4*)((const float*)p.res + ro);
                                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                                } else {
                                    float rv[4];
                                    load4<T>((const T*)p.res + ro, rv);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                                }
                            }
                            if (!p.res_post) activate(v);
                            const int byte = nl * osz;
                            char* dst = smem + rl * row_bytes + ((((byte >> 4) ^ rl) & (cpr - 1)) << 4) + (byte & 15);
                            if (osz == 4) {
                                if (IsSplit<T>::value && !p.out_f32) {
                                    f16x2_raise(f16x2_out_of_range(v));
                                    *(uint4*)dst = Chunk<T>::pack(v);
                                } else {
                                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                                }
                            } else if constexpr (sizeof(T) == 2) {
                                union { T h[4]; uint2 u; } pk;
#pragma unroll
                                for (int r = 0; r < 4; ++r) pk.h[r] = from_f32<T>(v[r]);
                                *(uint2*)dst = pk.u;
                            }
                        }
                    }
                }
                __syncthreads();
                const int nvalid_bytes = (min(BN, p.N_g - en0)) * osz;  // N tail of the last column tile (multiple of 16)
                for (int idx = tid; idx < rows_pp * cpr; idx += 256) {
                    const int rl = idx / cpr, c = idx - rl * cpr;
                    const int m = em0 + ps * rows_pp + rl;
                    if (m < p.M && (c << 4) < nvalid_bytes) {
                        uint4 val = *(const uint4*)(smem + rl * row_bytes + (((c ^ rl) & (cpr - 1)) << 4));
                        if (p.res_in_store) {  // x += ...: whole-row float4 reads of the fp32 residual stream
                            const float4 rv = *(const float4*)((const char*)p.res + ((size_t)m * p.ldr + grp * p.N_g + en0) * 4 + (c << 4));
                            val.x = __float_as_uint(__uint_as_float(val.x) + rv.x);
                            val.y = __float_as_uint(__uint_as_float(val.y) + rv.y);
                            val.z = __float_as_uint(__uint_as_float(val.z) + rv.z);
                            val.w = __float_as_uint(__uint_as_float(val.w) + rv.w);
                        }
                        char* o = (char*)p.out + ((size_t)m * p.ldo + p.out_coff + grp * p.N_g + en0) * osz + (c << 4);
                        *(uint4*)o = val;
                    }
                }
            }
            return;
        }
        if constexpr (TN == 64) {
            if (p.store_mode == ST_DB_TAIL) {
                const float* w2 = p.offs + grp * 256;    // [64][4] second-deconv weights of this branch (c_in, a'b')
                const float b2 = p.offs[512 + grp];
                float* map = grp == 0 ? (float*)p.out : (float*)p.out2;
                const int ab = (en0 + wn * TN) >> 6;     // this wave's sub-pixel (a, b) of the first deconv
#pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const int m = em0 + wm * TM + b * 16 + lr;
                    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int a = 0; a < NI; ++a) {
                        const int n = en0 + wn * TN + a * 16 + 4 * g;
                        const int co = n & 63;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = fmaxf(unscale<T>(acc[a][b][r], p.wscale) + p.bias[grp * p.N_g + n + r], 0.f);   // deconv1 + BN + ReLU
                            const float4 wv = *(const float4*)(w2 + (co + r) * 4);
                            q[0] = fmaf(v, wv.x, q[0]); q[1] = fmaf(v, wv.y, q[1]); q[2] = fmaf(v, wv.z, q[2]); q[3] = fmaf(v, wv.w, q[3]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        q[i] += __shfl_xor(q[i], 16);
                        q[i] += __shfl_xor(q[i], 32);
                    }
                    if (m < p.M) {  // lane group g writes output sub-position (a', b') = (g>>1, g&1)
                        const int t = fastdiv(m, p.mg_ow), ow = m - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
                        const float val = (g == 0 ? q[0] : (g == 1 ? q[1] : (g == 2 ? q[2] : q[3]))) + b2;
                        const size_t row = (size_t)img * (4 * p.OH) + 4 * oh + 2 * (ab >> 1) + (g >> 1);
                        map[row * (4 * p.OW) + 4 * ow + 2 * (ab & 1) + (g & 1)] = val;
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = em0 + wm * TM + b * 16 + lr;
            if (m >= p.M) continue;
            size_t orow = (size_t)m, rrow = 0;
            if (p.store_mode == ST_SHUFFLE2 || p.res_mode == RES_UP2) {
                const int t = fastdiv(m, p.mg_ow), ow = m - t * p.OW, img = fastdiv(t, p.mg_oh), oh = t - img * p.OH;
                if (p.res_mode == RES_UP2) rrow = ((size_t)img * (p.OH >> 1) + (oh >> 1)) * (p.OW >> 1) + (ow >> 1);
                if (p.store_mode == ST_SHUFFLE2) orow = ((size_t)img * (2 * p.OH) + 2 * oh) * (2 * p.OW) + 2 * ow;
            }
            if (p.res_mode == RES_SAME) rrow = (size_t)m;
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                const int n = en0 + wn * TN + a * 16 + 4 * g;  // within group
                if (n >= p.N_g) continue;
                float v[4] = {unscale<T>(acc[a][b][0], p.wscale), unscale<T>(acc[a][b][1], p.wscale), unscale<T>(acc[a][b][2], p.wscale),
                              unscale<T>(acc[a][b][3], p.wscale)};
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
                if (p.res_post) activate(v);
                if (p.res_mode != RES_NONE) {
                    const size_t ro = rrow * p.ldr + nglob;
                    if (p.res_f32) {
                        const float4 rv = *(const float4*)((const float*)p.res + ro);
                        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                    } else {
                        float rv[4];
                        load4<T>((const T*)p.res + ro, rv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    }
                }
                if (!p.res_post) activate(v);
                size_t oo;
                if (p.store_mode == ST_SHUFFLE2) {
                    const int ab = n / p.shuffle_co, co = n - ab * p.shuffle_co;
                    oo = (orow + (size_t)(ab >> 1) * (2 * p.OW) + (ab & 1)) * p.ldo + p.out_coff + grp * p.shuffle_co + co;
                } else {
                    oo = orow * p.ldo + p.out_coff + nglob;
                }
                if (p.out_f32 || IsF32<T>::value) *(float4*)((float*)p.out + oo) = make_float4(v[0], v[1], v[2], v[3]);
                else store4<T>((T*)p.out + oo, v);
            }
        }
    };

    // ---- persistent tile loop
    int tile = xcd_remap(blockIdx.x, G);
    if (tile >= total) return;
    setup(tile);
    issue(0);
    commit();
    __syncthreads();
    for (;;) {
        const int em0 = m0, en0 = n0;
        const int next = tile + G;
        const bool has_next = PERSIST == 1 && next < total;
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) {
                issue(ks + 1);
            } else if (has_next) {  // cross the tile boundary: the next tile's first K-step flies during this tile's last MFMAs + epilogue
                setup(next);
                issue(0);
            }
            // f16x2, 128-wide tiles (two workgroups per CU: registers to spare): both chunks at once, regrouped into (hi, lo) quartets, three
            // MFMAs per fragment pair.  The narrower tiles live on occupancy (4-5 workgroups per CU) and keep the two-chunk loop below
            // (four MFMAs per pair): holding both chunks cost them a workgroup per CU and 8-13 % (measured).
            // (round 4: with the in-place regrouping the narrow tiles keep their occupancy in the three-product form too, but gain only 2-5 %
            // -- they are not bound by the matrix pipe -- so they stay as they are)
            if constexpr (IsSplit<T>::value && BN >= 128) {
                typedef typename Mma<T>::u4v U;
                U xH[MI], xL[MI];
#pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const char* r = As + (wm * TM + b * 16 + lr) * 128;
                    Mma<T>::regroup(*(const uint4*)(r + fo0), *(const uint4*)(r + fo1), xH[b], xL[b]);
                }
                // weight fragments one row ahead of their MFMAs, fenced (left alone hipcc reads a row's two fragments right in front of its MFMAs
                // and waits lgkmcnt(0) for them: an exposed LDS round trip per row, four per K-step)
                uint4 wq[2][2];
                auto wread = [&](int a, int set) {
                    const char* r = Bs + (wn * TN + a * 16 + lr) * 128;
                    wq[set][0] = *(const uint4*)(r + fo0);
                    wq[set][1] = *(const uint4*)(r + fo1);
                };
                wread(0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    if (a + 1 < NI) wread(a + 1, (a + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    U wH, wL;    // (copies, not the in-place swap: pinned right in front of the MFMAs an inline-asm v_swap_b32 fed them stale registers
                                 // -- hipcc places wait states only for instructions it can see; tests/test_gpu_kernels.py::test_conv_kernel[case6-f16x2])
                    Mma<T>::regroup(wq[a & 1][0], wq[a & 1][1], wH, wL);
#pragma unroll
                    for (int b = 0; b < MI; ++b) Mma<T>::three(wH, wL, xH[b], xL[b], acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // the two 16-byte halves of this lane's 32-byte K slice
                const int fo = h == 0 ? fo0 : fo1;
                uint4 xf[MI], wf[NI];
#pragma unroll
                for (int b = 0; b < MI; ++b) xf[b] = *(const uint4*)(As + (wm * TM + b * 16 + lr) * 128 + fo);
#pragma unroll
                for (int a = 0; a < NI; ++a) wf[a] = *(const uint4*)(Bs + (wn * TN + a * 16 + lr) * 128 + fo);
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b) Mma<T>::half(wf[a], xf[b], acc[a][b]);
            }
            if (ks + 1 < nk) {
                __syncthreads();  // every wave is done reading the stage
                commit();
                __syncthreads();
            }
        }
        epilogue(em0, en0);
        if constexpr (PERSIST == 2) {
            if (next >= total) break;
            setup(next);
            issue(0);         // the stores of the tile just finished are still draining
            __syncthreads();  // LDS (staged output tile) is free again
            commit();
            __syncthreads();
            tile = next;
            continue;
        }
        if (!has_next) break;
        __syncthreads();  // LDS (last K-step's operands, or the staged output tile) is free again
        commit();
        __syncthreads();
        tile = next;
    }
}

// Tile choice shared by the packer (host) and the launcher.
static inline int conv_bn_for(int n_g) {
    static const int cap = getenv("OCRVI_CONV_BN") ? atoi(getenv("OCRVI_CONV_BN")) : 128;  // experiment knob (packing + launch agree)
    const int bn = n_g > 64 ? 128 : (n_g > 32 ? 64 : 32);
    return bn > cap ? cap : bn;
}
static inline int conv_bke(int dtype) { return dtype_size(dtype) == 4 ? 32 : 64; }   // elements per 128-byte K-step

// Deformable layers with whole 128-byte channel blocks (64 channels in 16 bits, 32 in fp32) run on dcn_pipe.h, whose K order is
// (channel block, tap, channel) instead of (tap, channel): packer and launcher must agree, so both ask this.
static inline int dcn_pipe_block(int dtype) { return dtype_size(dtype) == 4 ? 32 : 64; }
static inline bool dcn_pipe_packing(int dtype, int cin_g) {
    static const bool on = !(getenv("OCRVI_DCN_PIPE") && atoi(getenv("OCRVI_DCN_PIPE")) == 0);   // A/B switch
    // fp32: measured equal to conv_gemm's AM_DCN mode (both sit at ~56 % of the fp32 MFMA peak: with fp32 MFMAs 16x slower the gather is
    // hidden either way and what is left is tile-count quantisation: 600 tiles of 128 pixels on 256 CUs), so fp32 stays on conv_gemm
    static const bool on32 = getenv("OCRVI_DCN_PIPE_F32") && atoi(getenv("OCRVI_DCN_PIPE_F32")) != 0;
    // (f16x2: MFMAs are 4x shorter than fp32's per K-step, so the gather is no longer hidden behind them: pipelined, like the 16-bit types)
    return on && (dtype != OCRVI_F32 || on32) && cin_g % dcn_pipe_block(dtype) == 0;
}

template <typename T> int launch_conv(const ConvParams& p, int amode, hipStream_t stream);
int launch_conv_dt(int dtype, const ConvParams& p, int amode, hipStream_t stream);

// Host-side packing into [groups][Np][Kp] (dtype T bytes), K order = (tap, cin) / ROWS / deconv.
struct PackedConv {
    std::vector<char> bytes;
    std::vector<float> bias;  // expanded (ST_SHUFFLE2: 4x)
    int Np = 0, Kp = 0, N_g = 0, Cin_g = 0, groups = 1, KH = 1;
    float wscale = 1.f;       // f16x2: what the epilogue multiplies the accumulator by (the stored weights are w / wscale)
};
// w: [Cout][Cin_g][KH][KW] fp32 (conv) -- amode AM_CONV1/AM_CONV3/AM_DCN/AM_ROWS
PackedConv pack_conv(const float* w, const float* bias, int cout, int cin_g, int kh, int kw, int groups, int amode, int dtype);
// `groups` ConvTranspose2d(k=2,s=2) weights [Cin][Cout][2][2] -> grouped GEMM [group][n=(a*2+b)*Cout+co][k=ci]
PackedConv pack_deconv2(const float* const* w, const float* const* bias, int groups, int cin, int cout, int dtype);

}  // namespace ocrvi
