// launch_conv<T>: host-side validation + tile dispatch for conv_gemm_kernel.  Included by conv_{f32,bf16,f16}.hip
// (one translation unit per dtype so they compile in parallel).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "conv_gemm.h"
#include "gemm_ring.h"
#include "gemm_duo.h"
#include "gconv32.h"
#include "dcn_pipe.h"
#include "offs_conv.h"

namespace ocrvi {

template <typename T, int AMODE, int BM, int BN, int WM, int WN>
static int launch_tile(const ConvParams& p, hipStream_t stream) {
    constexpr int smem = (BM + BN) * 128;
    // one tile per workgroup (PERSIST = 0): the persistent variants of conv_gemm_kernel were measured and lose 10-35 % on every shape
    // (profiles/r01_conv_variants.md); they are no longer instantiated
    const int total = cdiv(p.M, BM) * (p.Np / BN);
    hipLaunchKernelGGL((conv_gemm_kernel<T, AMODE, BM, BN, WM, WN, 0>), dim3(total, p.groups), dim3(256), smem, stream, p);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T, int BM, int NW, int SPS, bool F32O, bool C3, int BN, int ACT>
static int launch_ring_act(const ConvParams& p, int grid, hipStream_t stream) {
    constexpr int smem = 3 * (BM + BN) * 128 + 512;  // ring + bias
    auto kern = gemm_ring_kernel<T, BM, NW, SPS, F32O, C3, BN, false, ACT>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, stream, p);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

template <typename T, int BM, int NW, int SPS, bool F32O, bool C3, int BN = 128>
static int launch_ring_f(const ConvParams& p, int grid, hipStream_t stream) {
    constexpr int smem = 3 * (BM + BN) * 128 + 512;  // ring + bias
#ifdef OCRVI_RING_PROF_BUILD
    static const bool prof = getenv("OCRVI_RING_PROF") && atoi(getenv("OCRVI_RING_PROF"));
    if (prof) {  // development aid: cycle breakdown per phase, printed per launch (synchronises)
        auto pk = gemm_ring_kernel<T, BM, NW, SPS, F32O, C3, BN, true>;
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) {
            OCRVI_HIP(hipMalloc((void**)&dbuf, 64));
            OCRVI_HIP(hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        OCRVI_HIP(hipMemsetAsync(dbuf, 0, 64, stream));
        ConvParams q = p;
        q.out2 = dbuf;
        hipLaunchKernelGGL(pk, dim3(grid), dim3(NW * 64), smem, stream, q);
        unsigned long long h[5];
        OCRVI_HIP(hipMemcpyAsync(h, dbuf, 40, hipMemcpyDeviceToHost, stream));
        OCRVI_HIP(hipStreamSynchronize(stream));
        const double w = (double)NW * grid, tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
        fprintf(stderr, "ring BM%d NW%d SPS%d M%d N%d K%d grid %d nk %d act %d res %d f32o %d: cycles/wave own-DMA wait %.0f barrier %.0f issue %.0f mma %.0f epi %.0f (%.0f%% %.0f%% %.0f%% %.0f%% %.0f%%)\n",
                BM, NW, SPS, p.M, p.N_g, p.Kp, grid, p.Kp / (int)(128 / sizeof(T)), p.act, p.res_mode, p.out_f32, h[4] / w, h[0] / w, h[1] / w, h[2] / w, h[3] / w,
                100 * h[4] / tot, 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot);
        return OCRVI_OK;
    }
#endif
    // one instantiation per activation (compile-time epilogue: no run-time selects in the slice groups)
    switch (p.act) {
        case ACT_NONE: return launch_ring_act<T, BM, NW, SPS, F32O, C3, BN, ACT_NONE>(p, grid, stream);
        case ACT_RELU: return launch_ring_act<T, BM, NW, SPS, F32O, C3, BN, ACT_RELU>(p, grid, stream);
        case ACT_GELU: return launch_ring_act<T, BM, NW, SPS, F32O, C3, BN, ACT_GELU>(p, grid, stream);
    }
    set_error("gemm_ring: unknown activation %d", p.act);
    return OCRVI_EINVAL;
}

template <typename T, int BM, int NW, int SPS>
static int launch_ring_cfg(const ConvParams& p, int amode, int grid, hipStream_t stream) {
    if constexpr (sizeof(T) == 4) {
        return launch_ring_f<T, BM, NW, SPS, true, false>(p, grid, stream);
    } else {
        if (amode == AM_CONV3) {  // 3x3 mode: 16-bit output, 256-row tiles only (checked by gemm_ring_eligible)
            if constexpr (BM == 256) return launch_ring_f<T, BM, NW, SPS, false, true>(p, grid, stream);
            set_error("gemm_ring: 3x3 mode needs 256-row tiles");
            return OCRVI_EINVAL;
        }
        if (p.out_f32) return launch_ring_f<T, BM, NW, SPS, true, false>(p, grid, stream);
        return launch_ring_f<T, BM, NW, SPS, false, false>(p, grid, stream);
    }
}

template <typename T>
int launch_gemm_ring(const ConvParams& p_in, int amode, hipStream_t stream) {
    ConvParams p = p_in;
    OCRVI_TRY(ring_pages(&p.zero_page, &p.dump_page));
    {   // range of the output buffer descriptor (gemm_ring_eligible has checked that it is below 4 GiB)
        const size_t osz = (sizeof(T) == 4 || p.out_f32) ? 4 : 2;
        p.out_bytes = (unsigned)(((size_t)(p.M - 1) * p.ldo + p.out_coff + p.N_g) * osz);
    }
    {   // non-temporal output stores (OCRVI_RING_NT=1; off by default).  Alone on the chip a GEMM whose output outweighs its operands
        // (N >= 2 K) runs 3-23 % faster with them (K 128 N 512 244 -> 188 us, K 256 N 1024 181 -> 153, K 384 N 1152 226 -> 208: the output
        // no longer displaces the activation rows the other column tiles want from L2) -- but inside the models the next layer then
        // misses what it would have found in L2 / the Infinity Cache, and the ring total of a bench step does not move (131.8 vs 131.7 ms;
        // profiles/r04_duo.md section 3)
        static const int force = getenv("OCRVI_RING_NT") ? atoi(getenv("OCRVI_RING_NT")) : 0;
        p.nt_out = (force == 1 && sizeof(T) == 4 && p.N_g >= 2 * p.Kp && p.Kp >= 128) ? 1 : (force == 2 ? 1 : 0);
    }
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const int bn = p.Np % 128 == 0 ? 128 : 64;  // 64-channel layers: a 256x64 tile (16-bit types only, checked by gemm_ring_eligible)
    const int ntiles = p.Np / bn, nk = p.Kp / (int)(128 / sizeof(T));
    OCRVI_CHECK(ntiles >= 1 && ntiles <= n_cu && nk >= 1, OCRVI_EINVAL, "gemm_ring: Np=%d Kp=%d out of range", p.Np, p.Kp);
    // 256-row tiles (4 slice groups riding on the next tile's first 4 K-steps) when K is deep enough for that and M still gives every
    // CU work; otherwise 128-row tiles (one group)
    static const bool mid = !(getenv("OCRVI_RING_MID") && atoi(getenv("OCRVI_RING_MID")) == 0);  // experiment knob
    const bool big_m = amode == AM_CONV3 || cdiv(p.M, 256) * ntiles >= 192;
    const bool f32o = sizeof(T) == 4 || p.out_f32;
    // nk = 2..3 with a 16-bit output: 256-row tiles with two slice groups of two slices (the fp32-output build of that shape spills)
    const bool mid256 = mid && nk >= 2 && nk < 4 && !f32o && big_m && amode == AM_CONV1;
    // (fp32 GEMMs -- the parity mode -- stay on 128-row tiles: their 256-row build does not fit the register file without a spill)
    // (f16x2: 4-byte operands like fp32, but its 256-row build -- 255 VGPRs, no scratch -- fits, and the kernel is bound by the bytes it
    // streams per FLOP, not by the matrix pipe: 256 x 128 tiles move 2/3 of the bytes of 128 x 128 ones)
    const bool wide_ok = sizeof(T) == 2 || IsSplit<T>::value;
    // (f16x2 3x3 mode: 128-row tiles -- the 256-row build with the 3x3 mode's per-piece pointers and tap masks needs 8 VGPRs more than a wave has)
    const bool c3_x2 = IsSplit<T>::value && amode == AM_CONV3;
    const int bm = bn == 64 ? 256 : ((wide_ok && !c3_x2 && ((nk >= 4 && big_m) || mid256)) ? 256 : 128);
    // one persistent workgroup per CU; a workgroup keeps its column tile, so the grid is Gm row-tile lanes x ntiles, with Gm chosen
    // for equal row-tile counts
    const int mtiles = cdiv(p.M, bm);
    int gm = std::min(mtiles, std::max(1, n_cu / ntiles));
    gm = cdiv(mtiles, cdiv(mtiles, gm));
    const int grid = gm * ntiles;
    if constexpr (sizeof(T) == 2) {
        if (bn == 64) {  // 8 waves along M (32 rows each: one slice group of two slices)
            return launch_ring_f<T, 256, 8, 2, false, false, 64>(p, grid, stream);
        }
    }
    if constexpr (sizeof(T) == 2) {
        if (bm == 256 && mid256) return launch_ring_f<T, 256, 8, 2, false, false>(p, grid, stream);
    }
    if constexpr (sizeof(T) == 2 || IsSplit<T>::value) {
        if (bm == 256) return launch_ring_cfg<T, 256, 8, 1>(p, amode, grid, stream);
    }
    if constexpr (IsSplit<T>::value) {
        if (amode == AM_CONV3) return launch_ring_f<T, 128, 8, 2, true, true>(p, grid, stream);
    }
    return launch_ring_cfg<T, 128, 8, 2>(p, amode, grid, stream);  // MI = 2: one group
}

template <typename T, int AMODE>
static int launch_mode(const ConvParams& p, hipStream_t stream) {
    const int bn = conv_bn_for(p.N_g);
    OCRVI_CHECK(p.Np % bn == 0 && p.Np >= p.N_g, OCRVI_EINVAL, "conv: Np=%d not a multiple of BN=%d / < N_g=%d", p.Np, bn, p.N_g);
    if constexpr (AMODE == AM_DCN) {
        OCRVI_CHECK(bn == 128, OCRVI_EINVAL, "dcn: needs N_g > 64 (got %d)", p.N_g);
        // 64-row tiles (half the in-flight corner loads); 256-wide N tiles when C_out allows so the gathered + blended A rows are
        // shared by twice as many output channels (the blend is the VALU bottleneck of this kernel)
        // ... unless 64-row tiles leave the second round of the chip's 2 x CUs workgroup slots mostly empty (layer4: 600 tiles on 512
        // slots); 32-row tiles keep the sharing and halve the quantum (measured 1304 -> 1014 us there, 2-8 % slower elsewhere)
        if (p.Np % 256 == 0) {
            int n_cu = 0;
            OCRVI_TRY(device_cus(&n_cu));
            bool small = cdiv(p.M, 64) * (p.Np / 256) < 3 * n_cu;
            const char* fe = getenv("OCRVI_DCN_TILE_M");   // test knob (read per launch so that a test can sweep it): 32 or 64 rows
            const int force = fe ? atoi(fe) : 0;
            if (force == 32 || force == 64) small = force == 32;
            if (small) return launch_tile<T, AMODE, 32, 256, 1, 4>(p, stream);
            return launch_tile<T, AMODE, 64, 256, 2, 2>(p, stream);
        }
        return launch_tile<T, AMODE, 64, 128, 2, 2>(p, stream);
    } else if constexpr (AMODE == AM_ROWS) {
        OCRVI_CHECK(bn <= 64, OCRVI_EINVAL, "rows-mode stem conv: N_g=%d > 64 unsupported", p.N_g);
        if (bn == 64) return launch_tile<T, AMODE, 128, 64, 2, 2>(p, stream);
        return launch_tile<T, AMODE, 128, 32, 4, 1>(p, stream);
    } else {
        if (bn == 128) return launch_tile<T, AMODE, 128, 128, 2, 2>(p, stream);
        if (bn == 64) return launch_tile<T, AMODE, 128, 64, 2, 2>(p, stream);
        return launch_tile<T, AMODE, 128, 32, 4, 1>(p, stream);
    }
}

template <typename T>
int launch_conv(const ConvParams& p_in, int amode, hipStream_t stream) {
    constexpr int EPC = TypeInfo<T>::EPC, BKE = 8 * EPC;
    ConvParams p = p_in;
    OCRVI_CHECK(p.M > 0 && p.M < (1 << 23) && p.OW > 0 && p.OH > 0, OCRVI_EINVAL, "conv: M=%d outside (0, 2^23)", p.M);
    p.mg_ow = ((1ull << 40) / (unsigned long long)p.OW) + 1;
    p.mg_oh = ((1ull << 40) / (unsigned long long)p.OH) + 1;
    p.identity_pix = (amode == AM_CONV1 && p.SH == 1 && p.SW == 1 && p.PH == 0 && p.PW == 0 && p.H == p.OH && p.W == p.OW &&
                      p.store_mode == ST_NHWC && p.res_mode != RES_UP2) ? 1 : 0;
    {   // coalesced LDS-staged epilogue whenever 16-byte row chunks are aligned (A/B switch: OCRVI_CONV_EPI=direct)
        static const bool direct = getenv("OCRVI_CONV_EPI") && !strcmp(getenv("OCRVI_CONV_EPI"), "direct");
        const int osz = (p.out_f32 || sizeof(T) == 4) ? 4 : 2, per = 16 / osz;
        p.epi_lds = (!direct && p.store_mode == ST_NHWC && p.res_mode != RES_UP2 && p.N_g % per == 0 && p.ldo % per == 0 &&
                     p.out_coff % per == 0 && ((uintptr_t)p.out & 15) == 0) ? 1 : 0;
        // (raw fp32 on both sides: an fp32 model, or fp32 output AND fp32 residual of a 16-bit / f16x2 one)
        const bool raw32 = IsF32<T>::value || (p.out_f32 && p.res_f32);
        p.res_in_store = (p.epi_lds && p.res_mode == RES_SAME && osz == 4 && raw32 && p.act == ACT_NONE &&
                          p.ldr % 4 == 0 && ((uintptr_t)p.res & 15) == 0) ? 1 : 0;
    }
    OCRVI_CHECK(p.x && p.w && p.out, OCRVI_EINVAL, "conv: null operand");
    OCRVI_CHECK(p.M > 0 && p.M == p.n_img * p.OH * p.OW, OCRVI_EINVAL, "conv: M=%d != %d*%d*%d", p.M, p.n_img, p.OH, p.OW);
    OCRVI_CHECK(p.Kp > 0 && p.Kp % BKE == 0, OCRVI_EINVAL, "conv: Kp=%d not a multiple of %d", p.Kp, BKE);
    OCRVI_CHECK(p.groups >= 1 && p.N_g >= 1, OCRVI_EINVAL, "conv: bad groups/N");
    OCRVI_CHECK((size_t)p.M * (size_t)(p.ldo > 32 ? p.ldo : 32) < ((size_t)1 << 40), OCRVI_EINVAL, "conv: output too large");
    if (p.store_mode == ST_DB_TAIL) {
        OCRVI_CHECK(amode == AM_CONV1 && p.shuffle_co == 64 && p.N_g == 256 && p.Np == 256 && p.out2 && p.offs && p.bias && p.groups == 2,
                    OCRVI_EINVAL, "db-tail deconv: needs 2 groups of 4x64 columns, both logit maps and the second-deconv weights");
    } else if (p.store_mode != ST_DCN_OFFS) {
        OCRVI_CHECK(p.N_g % 4 == 0 && p.ldo % 4 == 0 && p.out_coff % 4 == 0, OCRVI_EINVAL,
                    "conv: N_g=%d ldo=%d coff=%d must be multiples of 4", p.N_g, p.ldo, p.out_coff);
    } else {
        OCRVI_CHECK(p.bias && p.N_g <= 32 && p.groups == 1, OCRVI_EINVAL, "dcn offset conv: needs bias, N<=32");
    }
    if (p.res_mode != RES_NONE) OCRVI_CHECK(p.res && p.ldr % 4 == 0, OCRVI_EINVAL, "conv: residual missing / ldr%%4");
    if (p.res_mode == RES_UP2) OCRVI_CHECK(p.OH % 2 == 0 && p.OW % 2 == 0, OCRVI_EINVAL, "conv: RES_UP2 needs even OH/OW");
    if (p.store_mode == ST_SHUFFLE2)
        OCRVI_CHECK(p.shuffle_co > 0 && p.shuffle_co % 4 == 0 && p.N_g == 4 * p.shuffle_co, OCRVI_EINVAL, "conv: bad pixel-shuffle N");
    if (amode == AM_ROWS) {
        OCRVI_CHECK(p.groups == 1 && p.Hp >= (p.OH - 1) * p.SH + p.KH && p.Wp >= (p.OW - 1) * p.SW + 8 && p.Wp % 2 == 0 &&
                        p.Kp >= p.KH * 32 && (p.SW * 4) % EPC == 0,
                    OCRVI_EINVAL, "rows-mode conv: padded input %dx%d too small for %dx%d out", p.Hp, p.Wp, p.OH, p.OW);
        return launch_mode<T, AM_ROWS>(p, stream);
    }
    OCRVI_CHECK(p.Cin_g % EPC == 0 && p.Cin % EPC == 0 && p.cin_off % EPC == 0 && p.cin_off + p.groups * p.Cin_g <= p.Cin,
                OCRVI_EINVAL, "conv: channel counts Cin=%d Cin_g=%d off=%d must be multiples of %d", p.Cin, p.Cin_g, p.cin_off, EPC);
    const int ks = amode == AM_CONV1 ? 1 : 3;
    OCRVI_CHECK(p.Kp >= ks * ks * p.Cin_g, OCRVI_EINVAL, "conv: Kp=%d < %d", p.Kp, ks * ks * p.Cin_g);
    OCRVI_CHECK((p.OH - 1) * p.SH - p.PH + ks - 1 < p.H + ks && (p.OW - 1) * p.SW - p.PW + ks - 1 < p.W + ks, OCRVI_EINVAL,
                "conv: output %dx%d inconsistent with input %dx%d", p.OH, p.OW, p.H, p.W);
    if constexpr (sizeof(T) == 2) {
        if (gconv32_eligible(p, amode, 2)) return launch_gconv32<T>(p, stream);
    }
    if (offs_conv_eligible(p, amode, TypeInfo<T>::dtype)) return launch_offs_conv<T>(p, stream);
    if constexpr (IsSplit<T>::value) {
        if (gemm_duo_eligible(p, amode, TypeInfo<T>::dtype)) return launch_gemm_duo<T>(p, stream);
    }
    if (gemm_ring_eligible(p, amode, TypeInfo<T>::dtype)) return launch_gemm_ring<T>(p, amode, stream);
    switch (amode) {
        case AM_CONV1: return launch_mode<T, AM_CONV1>(p, stream);
        case AM_CONV3: return launch_mode<T, AM_CONV3>(p, stream);
        case AM_DCN:
            OCRVI_CHECK(p.offs && p.Cin_g % BKE == 0 && p.groups == 1, OCRVI_EINVAL, "dcn: needs offsets and Cin %% %d == 0", BKE);
            if (dcn_pipe_eligible(p, TypeInfo<T>::dtype)) return launch_dcn_pipe<T>(p, stream);
            OCRVI_CHECK(!dcn_pipe_packing(TypeInfo<T>::dtype, p.Cin_g), OCRVI_EINVAL,
                        "dcn: weights are packed for the pipelined kernel but this call is not eligible for it (epilogue / alignment)");
            return launch_mode<T, AM_DCN>(p, stream);
        default: break;
    }
    set_error("conv: unknown A mode %d", amode);
    return OCRVI_EINVAL;
}

}  // namespace ocrvi
