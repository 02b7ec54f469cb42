// Pipelined modulated deformable 3x3 convolution (torchvision.ops.deform_conv2d as called in dcn.py:48-57), all three compute types.
//
// conv_gemm's AM_DCN mode runs gather -> wait -> blend -> LDS -> barrier -> MFMA -> barrier strictly in series per K-step and stages
// both operands through registers, so every K-step pays a full L2 round trip (measured 13-16 % of the MFMA peak).  Here the same
// implicit GEMM is a three-stage software pipeline, one barrier per K-step (64 channels of one tap):
//   step s:   issue the 4-corner row gathers of step s + 2        (inline-asm global loads into one of two register sets)
//             issue the weight stage of step s + 2                (LDS-DMA ring of three stages, as gemm_ring.h)
//             blend step s + 1 in fp32 from the other register set -> its [128 pixels][128 B] slab in LDS (two slabs)
//             MFMA step s from the slab and weight stage written earlier
// K order = (64-channel block, tap): the nine taps of a channel block follow each other, and a tile is a 2-D patch of output pixels
// (8 x 16, 16 x 8, ...), so the ~(patch + halo) x 128 B of input a channel block needs stay in the CU's 32-KiB L1 across its nine
// steps; the weight stages bypass L1 (sc1) so they do not evict them.
// The blend (VALU) of step s + 1 and the MFMAs of step s are independent, so they interleave inside a wave and across the two waves
// of a SIMD; a gather has one whole step to come back.  One workgroup = 8 waves = 128 output pixels x BN output channels (BN = all
// channels up to 256, so every (pixel, tap, channel) is gathered and blended once per tile): waves 4 (M) x 2 (N), 32 pixels x BN/2
// channels each.  All VMEM traffic in the loop is hand-counted (gathers: asm loads + counted vmcnt; weights: DMA); the sampling
// geometry of every (pixel, tap) is computed once per tile into LDS, so the compiler never inserts a vmcnt(0) into the pipeline.
#pragma once
#include "gemm_ring.h"

namespace ocrvi {

// PROF (development only, -DOCRVI_RING_PROF_BUILD + OCRVI_RING_PROF=1): per wave half (waves 0-3 / 4-7) the shader-clock cycles spent in the tile's
// geometry pass / waiting for own VMEM / at the barrier / issuing gathers and weight DMA / blending / ds_read + MFMA / in the epilogue, summed into p.out2.
template <typename T, int BN, bool PROF = false>
__global__ __launch_bounds__(512, 2) void dcn_pipe_kernel(const ConvParams p) {
    constexpr int EPC = TypeInfo<T>::EPC, BM = 128;     // elements per 16-byte chunk: 8 (16-bit types) or 4 (fp32)
    constexpr int CB = 8 * EPC;                         // channels per K-step (128 bytes): 64 or 32
    constexpr int WST = BN * 128, SLAB = BM * 128;
    constexpr int NI = BN / 32;            // 16-channel MFMA blocks per wave (BN / 2 channels)
    constexpr int GW = BN / 64;            // weight DMA instructions per wave per stage (BN / 8 pieces over 8 waves)
#ifdef OCRVI_TIMING_DCN_ONEC                // (development, timing only: one corner load per item instead of four -- wrong results)
    constexpr int GG = 2;
#else
    constexpr int GG = 8;                  // gather loads per lane per step: 2 (row, chunk) items x 4 corners
#endif
    static_assert(BN == 128 || BN == 256, "column tile");
    constexpr bool PERM = sizeof(T) == 2;               // 16-bit output: weight rows permuted so a lane ends with 8 consecutive channels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Wr = smem;                                  // [3][BN][128 B]
    char* const Sl = smem + 3 * WST;                        // [2][128][128 B]
    float4* const Gw = (float4*)(smem + 3 * WST + 2 * SLAB);        // [9][128] bilinear weight x mask of the 4 corners (0 outside)
    unsigned* const Go = (unsigned*)(smem + 3 * WST + 2 * SLAB + 9 * BM * 16);  // [9][128] byte offset of corner (y0, x0)'s pixel | dx | dy << 1
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    long long tk[7] = {0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if constexpr (PROF) t0 = clock64();
    auto tick = [&](int k) {
        if constexpr (PROF) {
            const long long t = clock64();
            tk[k] += t - t0;
            t0 = t;
        }
    };
    auto wait_bar = [&](auto NN) {   // counted wait, then the step's barrier
        constexpr int N = decltype(NN)::value;
        if constexpr (PROF) {
            wait_vm_only<N>();
            tick(1);
            asm volatile("s_barrier" ::: "memory");
            tick(2);
        } else {
            wait_vm_barrier<N>();
        }
    };
    const char* const xbase = uniform_ptr((const char*)p.x);
    const int rowb = p.Cin * (int)sizeof(T);                           // bytes per input pixel
    const int nk = 9 * (p.Cin_g / CB);
    const int lw = p.patch_lw, PW = 1 << lw, PH = BM >> lw;            // patch: PH rows x PW columns of output pixels
    const int pcols = (p.OW + PW - 1) >> lw, prows = (p.OH + PH - 1) / PH;
    const int ntiles = p.Np / BN, mtiles = p.n_img * prows * pcols, total = mtiles * ntiles;
    const int ldw_b = p.Kp * (int)sizeof(T);

    // gather items of this thread: rows grow = tid >> 3 and grow + 64 of the tile, 16-byte chunk gj = tid & 7 of the 128-byte K-step
    const int grow = tid >> 3, gj = tid & 7;
    // weight DMA: piece (wave + 8 j) = rows 8 (wave + 8 j) + prow of the stage; source chunk swizzled (rule 21)
    const int prow = lane >> 3;
    const unsigned w_voff = (unsigned)((8 * wave + prow) * ldw_b + (((lane & 7) ^ swz128(8 * wave + prow)) << 4));
    // MFMA fragment addressing (conv_gemm / gemm_ring conventions; B rows permuted so a lane ends with 8 consecutive channels)
    const int swa = swz128(lr);
    const int foa0 = ((2 * g) ^ swa) << 4, foa1 = ((2 * g + 1) ^ swa) << 4;
    const int brow = PERM ? 8 * (lr >> 2) + (lr & 3) : lr, swb = swz128(brow);
    const int fob0 = ((2 * g) ^ swb) << 4, fob1 = ((2 * g + 1) ^ swb) << 4;

    for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < total; tile += gridDim.x) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const int n0 = nt * BN;
        const int img = mt / (prows * pcols), pr = (mt - img * prows * pcols) / pcols, pc = mt - (img * prows + pr) * pcols;
        const int oh0 = pr * PH, ow0 = pc << lw;
        // output pixel of tile row `row`: (oh0 + (row >> lw), ow0 + (row & (PW - 1))); -1 when it falls outside the map
        auto pixel_of = [&](int row) -> int {
            const int oh = oh0 + (row >> lw), ow = ow0 + (row & (PW - 1));
            return (oh < p.OH && ow < p.OW) ? (img * p.OH + oh) * p.OW + ow : -1;
        };
        __syncthreads();  // the previous tile is done with the LDS
        // ---- sampling geometry of every (pixel, tap) of the tile, once: sampling point p = (oh s - 1 + i + dy, ow s - 1 + j + dx), bilinear
        // with out-of-range corners contributing 0, the whole sample 0 outside (-1, H) x (-1, W)  (torchvision deform_conv2d)
        for (int i = tid; i < 9 * BM; i += 512) {
            const int row = i & (BM - 1), tap = i >> 7;
            const int m = pixel_of(row);
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned ov = (unsigned)(img * p.H * p.W) * (unsigned)rowb;
            if (m >= 0) {
                const float* o = p.offs + (size_t)m * 32;
                const float dy = o[2 * tap], dx = o[2 * tap + 1], mk = o[18 + tap];
                const int r = tap / 3, s = tap - 3 * r;
                const int oh = oh0 + (row >> lw), ow = ow0 + (row & (PW - 1));
                const float py = (float)(oh * p.SH - p.PH + r) + dy, px = (float)(ow * p.SW - p.PW + s) + dx;
                const bool inside = py > -1.f && py < (float)p.H && px > -1.f && px < (float)p.W;
                // clamp before float->int so wild / NaN offsets cannot overflow (their weight is already 0)
                const float cy = fminf(fmaxf(py, -2.f), (float)p.H + 1.f), cx = fminf(fmaxf(px, -2.f), (float)p.W + 1.f);
                const float fy = floorf(cy), fx = floorf(cx);
                const float ly = cy - fy, lx = cx - fx, hy = 1.f - ly, hx = 1.f - lx;
                const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
                const bool oy0 = y0 >= 0, oy1 = y1 <= p.H - 1, ox0 = x0 >= 0, ox1 = x1 <= p.W - 1;
                const float mm = inside ? mk : 0.f;
                wv.x = (oy0 && ox0) ? hy * hx * mm : 0.f;
                wv.y = (oy0 && ox1) ? hy * lx * mm : 0.f;
                wv.z = (oy1 && ox0) ? ly * hx * mm : 0.f;
                wv.w = (oy1 && ox1) ? ly * lx * mm : 0.f;
                const int yc0 = min(max(y0, 0), p.H - 1), yc1 = min(max(y1, 0), p.H - 1);
                const int xc0 = min(max(x0, 0), p.W - 1), xc1 = min(max(x1, 0), p.W - 1);
                ov = ((unsigned)(img * p.H * p.W + yc0 * p.W + xc0) * (unsigned)rowb) | (unsigned)(xc1 - xc0) | ((unsigned)(yc1 - yc0) << 1);
            }
#ifdef OCRVI_TIMING_DCN_NEAR   // (development, timing only: every sample reads its own output pixel's neighbourhood -- the gathers hit L1)
            if (m >= 0) {
                const int oh = min(oh0 + (row >> lw), p.OH - 1) * p.SH, ow = min(ow0 + (row & (PW - 1)), p.OW - 1) * p.SW;
                ov = ((unsigned)(img * p.H * p.W + min(oh, p.H - 2) * p.W + min(ow, p.W - 2)) * (unsigned)rowb) | 3u;
            }
#endif
            Gw[i] = wv;
            Go[i] = ov;
        }
        __syncthreads();
        wait_vm_only<0>();  // nothing of this wave is in flight when the counted pipeline starts
        tick(0);

        const char* const w_tile = uniform_ptr((const char*)p.w + (size_t)n0 * ldw_b);
        auto issue_w = [&](int ks) {  // weight stage ks -> ring slot ks % 3
            const char* src = w_tile + (size_t)ks * 128;
            const unsigned dst = lds0 + (ks % 3) * WST + wave * 1024;
#pragma unroll
            for (int j = 0; j < GW; ++j) glds16_sc1(src + (size_t)(64 * j) * ldw_b, w_voff, __builtin_amdgcn_readfirstlane(dst + j * 8192));
        };
        // a gathered step: the 8 corner chunks plus the blend weights that go with them (the geometry registers move on to the
        // next tap while the loads are still in flight)
        struct GSet {
            u32x4 c[2][4];
            float w[2][4];
        };
        auto issue_gather = [&](int ks, GSet& S) {
            const int cb = ks / 9, tap = ks - 9 * cb;
            const unsigned coff = (unsigned)(p.cin_off + cb * CB + gj * EPC) * (unsigned)sizeof(T);   // byte offset of this lane's chunk inside a pixel
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int gi = tap * BM + grow + 64 * it;
                const float4 wv = Gw[gi];
                const unsigned ov = Go[gi];
                const unsigned o00 = (ov & ~3u) + coff, dxo = (ov & 1u) ? (unsigned)rowb : 0u, dyo = (ov & 2u) ? (unsigned)(rowb * p.W) : 0u;
                S.w[it][0] = wv.x; S.w[it][1] = wv.y; S.w[it][2] = wv.z; S.w[it][3] = wv.w;
                gload16s(S.c[it][0], xbase, o00);
#ifndef OCRVI_TIMING_DCN_ONEC
                gload16s(S.c[it][1], xbase, o00 + dxo);
                gload16s(S.c[it][2], xbase, o00 + dyo);
                gload16s(S.c[it][3], xbase, o00 + dyo + dxo);
#endif
            }
        };
        auto blend = [&](int ks, GSet& S) {  // fp32 blend of step ks -> slab ks & 1
            char* slab = Sl + (ks & 1) * SLAB;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bind16(S.c[it][q]);
#ifdef OCRVI_TIMING_DCN_ONEC
                S.c[it][1] = S.c[it][2] = S.c[it][3] = S.c[it][0];
#endif
                float acc[EPC], f[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 u = make_uint4(S.c[it][q].x, S.c[it][q].y, S.c[it][q].z, S.c[it][q].w);
                    if constexpr (IsSplit<T>::value) {
                        Chunk<T>::fma(u, S.w[it][q], acc);      // both halves of every channel in mixed-precision fmas, no separate conversions
                    } else {
                        Chunk<T>::unpack(u, f);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[e] = fmaf(S.w[it][q], f[e], acc[e]);
                    }
                }
                const int row = grow + 64 * it;
                if constexpr (IsSplit<T>::value) {
                    // the slab holds (hi, lo) QUARTETS: chunk 2 j = the hi halves of channels 8 j .. 8 j + 7, chunk 2 j + 1 their lo halves (the
                    // weights are packed the same way, pack_conv), so the MFMA stage reads its operands without regrouping them: 12 + 6 NI
                    // v_mov less per wave and step in a kernel that is VALU-bound
                    const uint4 e = Chunk<T>::pack(acc);
                    char* d = slab + row * 128 + (gj & 1) * 8;
                    *(uint2*)(d + (((gj & ~1) ^ swz128(row)) << 4)) = make_uint2(e.x, e.y);
                    *(uint2*)(d + (((gj | 1) ^ swz128(row)) << 4)) = make_uint2(e.z, e.w);
                } else {
                    *(uint4*)(slab + row * 128 + ((gj ^ swz128(row)) << 4)) = Chunk<T>::pack(acc);
                }
            }
        };
        f32x4 acc[NI][2];
#pragma unroll
        for (int a = 0; a < NI; ++a) acc[a][0] = acc[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mma = [&](int ks) {
            const char* As = Sl + (ks & 1) * SLAB + (wm * 32 + lr) * 128;
            const char* Bs = Wr + (ks % 3) * WST + (wn * (BN / 2) + brow) * 128;
            if constexpr (IsSplit<T>::value) {   // f16x2: three MFMAs per fragment pair (Mma<f16x2_t>::regroup / three)
                typedef typename Mma<T>::u4v U;
                auto q16 = [](const char* q) -> U { const uint4 v = *(const uint4*)q; return (U){v.x, v.y, v.z, v.w}; };
                const U xH[2] = {q16(As + foa0), q16(As + 16 * 128 + foa0)}, xL[2] = {q16(As + foa1), q16(As + 16 * 128 + foa1)};   // quartets (blend / pack_conv)
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const char* r = Bs + (16 * a) * 128;      // (PERM is false for a 4-byte type)
                    const U wH = q16(r + fob0), wL = q16(r + fob1);
                    Mma<T>::three(wH, wL, xH[0], xL[0], acc[a][0]);
                    Mma<T>::three(wH, wL, xH[1], xL[1], acc[a][1]);
                }
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int foa = h == 0 ? foa0 : foa1, fob = h == 0 ? fob0 : fob1;
                const uint4 x0 = *(const uint4*)(As + foa), x1 = *(const uint4*)(As + 16 * 128 + foa);
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const uint4 wf = *(const uint4*)(Bs + (PERM ? 32 * (a >> 1) + 4 * (a & 1) : 16 * a) * 128 + fob);
                    Mma<T>::half(wf, x0, acc[a][0]);
                    Mma<T>::half(wf, x1, acc[a][1]);
                }
            }
        };

        // ---- prologue.  VMEM queue order per step: gathers first, weight stage second, so that "all but the GW youngest" covers
        // the gathers the next blend needs and leaves the newest weight stage in flight.
        // The two waves of a SIMD (w and w + 4) run the step's two halves in opposite order, so that one blends (VALU) while the other
        // multiplies (matrix pipe) instead of both doing the same thing behind the step's barrier:
        //   waves 0-3:  wait + barrier | gathers(s+2) -> the other register set, weights(s+2) | blend(s+1) | MFMA(s)
        //   waves 4-7:  wait + barrier | weights(s+2) | MFMA(s) | blend(s+1) | gathers(s+2) -> the SAME register set (one set is enough:
        //               it is refilled right after the blend has consumed it, and has until the end of the next step to land)
        // Both orders finish blend(s+1) and MFMA(s) inside step s, so the slab / weight-ring hand-over at the barrier is unchanged; the
        // counted waits differ because the VMEM queue order does (waves 4-7: weights(s+1), gathers(s+1), weights(s+2), ...).
        GSet SA, SB;
        if (wave & 4) {
            issue_w(0);
            issue_gather(0, SA);
            wait_vm_only<0>();
            blend(0, SA);
            if (nk > 1) {
                issue_w(1);
                issue_gather(1, SA);
            }
            tick(3);
            for (int s = 0; s < nk; ++s) {
                // stage s has landed once everything but weights(s+1) and gathers(s+1) is done
                if (s + 1 < nk) wait_bar(IC<GG + GW>{}); else wait_bar(IC<0>{});
                if (s + 2 < nk) issue_w(s + 2);
                tick(3);
                mma(s);
                tick(5);
                if (s + 1 < nk) {
                    if (s + 2 < nk) wait_vm_only<GW>(); else wait_vm_only<0>();      // gathers(s+1): only weights(s+2) is younger
                    tick(1);
                    blend(s + 1, SA);
                    tick(4);
                    if (s + 2 < nk) issue_gather(s + 2, SA);
                    tick(3);
                }
            }
        } else {
            issue_w(0);
            issue_gather(0, SA);
            if (nk > 1) {
                issue_gather(1, SB);
                issue_w(1);
                wait_vm_only<GG + GW>();
            } else {
                wait_vm_only<0>();
            }
            blend(0, SA);
            // ---- steady state, unrolled by two so the register sets have static names:  step s blends s + 1 and multiplies s
            tick(3);
            auto step = [&](int s, GSet& Sblend, GSet& Snext) {
                if (s + 1 < nk) wait_bar(IC<GW>{}); else wait_bar(IC<0>{});
                if (s + 2 < nk) {
                    issue_gather(s + 2, Snext);
                    issue_w(s + 2);
                }
                tick(3);
                if (s + 1 < nk) blend(s + 1, Sblend);
                tick(4);
                mma(s);
                tick(5);
            };
            for (int s = 0; s < nk; s += 2) {
                step(s, SB, SA);          // blends s + 1 (set B), refills set A with s + 2
                if (s + 1 < nk) step(s + 1, SA, SB);
            }
        }
        // ---- epilogue: bias + activation; one 16-byte store per lane (16-bit: 8 consecutive channels thanks to the row permutation,
        // fp32: the accumulator's 4), 64 contiguous bytes per pixel and instruction
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = pixel_of(wm * 32 + 16 * b + lr);
            if constexpr (PERM) {
                float4 bvs[NI / 2][2];   // (bias of all blocks first, clamped addresses, outside the per-block branches: see below)
#pragma unroll
                for (int hh = 0; hh < NI / 2; ++hh) {
                    const int n = min(n0 + wn * (BN / 2) + 32 * hh + 8 * g, p.N_g - 8);
                    bvs[hh][0] = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    bvs[hh][1] = p.bias ? *(const float4*)(p.bias + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int hh = 0; hh < NI / 2; ++hh) {
                    const int n = n0 + wn * (BN / 2) + 32 * hh + 8 * g;
                    if (m >= 0 && n < p.N_g) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[2 * hh][b][r];
                            v[4 + r] = acc[2 * hh + 1][b][r];
                        }
                        {
                            v[0] += bvs[hh][0].x; v[1] += bvs[hh][0].y; v[2] += bvs[hh][0].z; v[3] += bvs[hh][0].w;
                            v[4] += bvs[hh][1].x; v[5] += bvs[hh][1].y; v[6] += bvs[hh][1].z; v[7] += bvs[hh][1].w;
                        }
                        if (p.act == ACT_RELU) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        *(uint4*)((T*)p.out + (size_t)m * p.ldo + p.out_coff + n) = Chunk<T>::pack(v);
                    }
                }
            } else {
                // (the bias of all NI blocks first, from clamped addresses and outside the per-block branches: inside them every block's load got
                // its own vmcnt(0) in front of its store)
                float4 bvs[NI];
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int n = min(n0 + wn * (BN / 2) + 16 * a + 4 * g, p.N_g - 4);
                    bvs[a] = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int n = n0 + wn * (BN / 2) + 16 * a + 4 * g;
                    if (m >= 0 && n < p.N_g) {
                        float v[4] = {unscale<T>(acc[a][b][0], p.wscale), unscale<T>(acc[a][b][1], p.wscale), unscale<T>(acc[a][b][2], p.wscale),
                                      unscale<T>(acc[a][b][3], p.wscale)};
                        {
                            v[0] += bvs[a].x; v[1] += bvs[a].y; v[2] += bvs[a].z; v[3] += bvs[a].w;
                        }
                        if (p.act == ACT_RELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        if constexpr (IsSplit<T>::value) f16x2_raise(f16x2_out_of_range(v));
                        *(uint4*)((T*)p.out + (size_t)m * p.ldo + p.out_coff + n) = Chunk<T>::pack(v);
                    }
                }
            }
        }
        tick(6);
    }
    if constexpr (PROF) {
        if (lane == 0)
            for (int k = 0; k < 7; ++k) atomicAdd((unsigned long long*)p.out2 + (wave >> 2) * 8 + k, (unsigned long long)tk[k]);
    }
}

static inline bool dcn_pipe_eligible(const ConvParams& p, int dtype) {
    const size_t esz = dtype_size(dtype);
    return dcn_pipe_packing(dtype, p.Cin_g) && p.groups == 1 && p.Kp == 9 * p.Cin_g && p.Np % 128 == 0 && p.N_g % 8 == 0 &&
           p.store_mode == ST_NHWC && p.res_mode == RES_NONE && (!p.out_f32 || dtype == OCRVI_F32) && (p.act == ACT_NONE || p.act == ACT_RELU) &&
           p.ldo % 8 == 0 && p.out_coff % 8 == 0 && p.offs != nullptr && (size_t)p.n_img * p.H * p.W * p.Cin * esz < ((size_t)1 << 32);
}

template <typename T>
static int launch_dcn_pipe(const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const bool wide = p.Np % 256 == 0;
    const int bn = wide ? 256 : 128;
    // patch shape: the (rows x columns) = (128 >> lw) x (1 << lw) that wastes the fewest pixels on this map; ties go to the squarer one
    int best_lw = 4;
    long best = -1;
    for (int lw : {4, 3, 5, 2, 6, 7}) {
        const int pw = 1 << lw, ph = 128 >> lw;
        const long covered = (long)cdiv(p.OH, ph) * ph * cdiv(p.OW, pw) * pw;
        if (best < 0 || covered < best) { best = covered; best_lw = lw; }
    }
    p.patch_lw = best_lw;
    const int total = p.n_img * cdiv(p.OH, 128 >> best_lw) * cdiv(p.OW, 1 << best_lw) * (p.Np / bn);
    const int grid = cdiv(total, cdiv(total, std::min(total, n_cu)));   // one persistent workgroup per CU, equal tile counts
    const int smem = 3 * bn * 128 + 2 * 128 * 128 + 9 * 128 * 20;
#ifdef OCRVI_RING_PROF_BUILD
    static const bool prof = getenv("OCRVI_RING_PROF") && atoi(getenv("OCRVI_RING_PROF"));
    if (prof) {  // development aid: cycle breakdown per phase and wave half, printed per launch (synchronises)
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) OCRVI_HIP(hipMalloc((void**)&dbuf, 128));
        OCRVI_HIP(hipMemsetAsync(dbuf, 0, 128, stream));
        ConvParams q = p;
        q.out2 = dbuf;
        if (wide) {
            auto k = dcn_pipe_kernel<T, 256, true>;
            OCRVI_TRY(ensure_max_smem((const void*)k, smem));
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, q);
        } else {
            auto k = dcn_pipe_kernel<T, 128, true>;
            OCRVI_TRY(ensure_max_smem((const void*)k, smem));
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, q);
        }
        unsigned long long h[16];
        OCRVI_HIP(hipMemcpyAsync(h, dbuf, 128, hipMemcpyDeviceToHost, stream));
        OCRVI_HIP(hipStreamSynchronize(stream));
        const int nk = 9 * (p.Cin_g / (int)(128 / sizeof(T)));
        const double tiles_per_wg = (double)total / grid;
        for (int hf = 0; hf < 2; ++hf) {
            const double w = 4.0 * grid;
            const unsigned long long* v = h + 8 * hf;
            double tot = 0;
            for (int k = 0; k < 7; ++k) tot += (double)v[k];
            fprintf(stderr, "dcn_pipe BN%d Cin %d %dx%d s%d grid %d tiles/wg %.1f nk %d waves %d-%d: cycles/wave geometry %.0f vm-wait %.0f barrier %.0f issue %.0f blend %.0f mma %.0f epilogue %.0f total %.0f | per step: wait %.0f barrier %.0f issue %.0f blend %.0f mma %.0f\n",
                    bn, p.Cin_g, p.H, p.W, p.SH, grid, tiles_per_wg, nk, 4 * hf, 4 * hf + 3, v[0] / w, v[1] / w, v[2] / w, v[3] / w, v[4] / w, v[5] / w, v[6] / w, tot / w,
                    v[1] / w / (tiles_per_wg * nk), v[2] / w / (tiles_per_wg * nk), v[3] / w / (tiles_per_wg * nk), v[4] / w / (tiles_per_wg * nk), v[5] / w / (tiles_per_wg * nk));
        }
        return OCRVI_OK;
    }
#endif
    if (wide) {
        auto k = dcn_pipe_kernel<T, 256>;
        OCRVI_TRY(ensure_max_smem((const void*)k, smem));
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, p);
    } else {
        auto k = dcn_pipe_kernel<T, 128>;
        OCRVI_TRY(ensure_max_smem((const void*)k, smem));
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, p);
    }
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

}  // namespace ocrvi
