// gemm_duo instantiations for f16x2_t (its own translation unit: parallel compile).
#include <stdlib.h>

#include <algorithm>

#include "gemm_duo.h"

namespace ocrvi {
OCRVI_RANGE_FLAG_TU()   // binds this unit's f16x2 range-flag pointer (common.h)

template <typename T, int NI, int ACT, int RESK, bool OUTF32, int F>
static int launch_duo_k(const ConvParams& p, const DuoPlan& plan, hipStream_t stream) {
    constexpr int smem = 2 * (64 * NI) * 128 + 6 * 128 * 128;
#ifdef OCRVI_RING_PROF_BUILD
    static const bool prof = getenv("OCRVI_RING_PROF") && atoi(getenv("OCRVI_RING_PROF"));
    if (prof) {  // development aid: cycle breakdown per phase, printed per launch (synchronises)
        auto pk = gemm_duo_kernel<T, NI, ACT, RESK, OUTF32, F, true>;
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) OCRVI_HIP(hipMalloc((void**)&dbuf, 128));
        OCRVI_TRY(ensure_max_smem((const void*)pk, smem));
        OCRVI_HIP(hipMemsetAsync(dbuf, 0, 128, stream));
        ConvParams q = p;
        q.out2 = dbuf;
        hipLaunchKernelGGL(pk, dim3(plan.gm * plan.ntiles), dim3(512), smem, stream, q, plan);
        unsigned long long h[16];
        OCRVI_HIP(hipMemcpyAsync(h, dbuf, 128, hipMemcpyDeviceToHost, stream));
        OCRVI_HIP(hipStreamSynchronize(stream));
        const double w = 4.0 * plan.gm * plan.ntiles;
        const int nk = p.Kp / 32, epi = 8 * NI / F, P = nk + epi;
        for (int g = 0; g < 2; ++g) {
            const unsigned long long* t = h + g * 8;
            fprintf(stderr, "duo NI%d F%d M%d N%d K%d grid %d nk %d P %d act %d resk %d nt %d group %d: cycles/wave own-DMA wait %.0f barrier %.0f mfma %.0f epi %.0f idle %.0f "
                    "book %.0f epi-plan %.0f epi-dma %.0f total %.0f\n", NI, F, p.M, p.N_g, p.Kp, plan.gm * plan.ntiles, nk, P, p.act, RESK, plan.nt, g, t[0] / w, t[1] / w, t[2] / w, t[3] / w, t[4] / w, t[5] / w,
                    t[6] / w, t[7] / w, (t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7]) / w);
        }
        return OCRVI_OK;
    }
#endif
    auto kern = gemm_duo_kernel<T, NI, ACT, RESK, OUTF32, F>;
    OCRVI_TRY(ensure_max_smem((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(plan.gm * plan.ntiles), dim3(512), smem, stream, p, plan);
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}

// the epilogue variants the two models use (gemm_duo_eligible admits exactly these): activation x residual kind x output format
template <typename T, int NI, int F>
static int launch_duo_f(const ConvParams& p, const DuoPlan& plan, int resk, hipStream_t stream) {
    const bool of32 = p.out_f32 != 0;
    if (resk == 0 && !of32) {
        if (p.act == ACT_NONE) return launch_duo_k<T, NI, ACT_NONE, 0, false, F>(p, plan, stream);
        if (p.act == ACT_RELU) return launch_duo_k<T, NI, ACT_RELU, 0, false, F>(p, plan, stream);
        if (p.act == ACT_GELU) return launch_duo_k<T, NI, ACT_GELU, 0, false, 2>(p, plan, stream);
    } else if (resk == 0 && of32) {
        if (p.act == ACT_NONE) return launch_duo_k<T, NI, ACT_NONE, 0, true, F>(p, plan, stream);
    } else if (resk == 2 && !of32) {
        if (p.act == ACT_NONE) return launch_duo_k<T, NI, ACT_NONE, 2, false, F>(p, plan, stream);
        if (p.act == ACT_RELU) return launch_duo_k<T, NI, ACT_RELU, 2, false, F>(p, plan, stream);
    } else if (resk == 1 && of32) {
        if (p.act == ACT_NONE) return launch_duo_k<T, NI, ACT_NONE, 1, true, F>(p, plan, stream);
    }
    set_error("gemm_duo: no build for act %d, residual kind %d, fp32 output %d", p.act, resk, (int)of32);
    return OCRVI_EINVAL;
}
// fragments per epilogue step: 8 plain ones (4 steps per tile at 64 columns per wave; 4 fragments per step -- twice the steps, each with
// its barrier and DMA duty -- measured 8-14 % slower at K <= 384, equal at K >= 1024: profiles/r04_duo.md), 2 with GELU
template <typename T, int NI>
static int launch_duo_ni(const ConvParams& p, const DuoPlan& plan, int resk, hipStream_t stream) {
    static const int f4 = getenv("OCRVI_DUO_F4") ? atoi(getenv("OCRVI_DUO_F4")) : 0;   // experiment knob: 4 fragments per step
    if (f4 && p.act != ACT_GELU) return launch_duo_f<T, NI, 4>(p, plan, resk, stream);
    return launch_duo_f<T, NI, 8>(p, plan, resk, stream);
}

template <>
int launch_gemm_duo<f16x2_t>(const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    p.out_bytes = (unsigned)(((size_t)(p.M - 1) * p.ldo + p.out_coff + p.N_g) * 4);
    int n_cu = 0;
    OCRVI_TRY(device_cus(&n_cu));
    const int bn = duo_bn_for(p.Np);
    OCRVI_CHECK(bn != 0, OCRVI_EINVAL, "gemm_duo: Np=%d is not a multiple of 256 or 192", p.Np);
    DuoPlan plan;
    plan.ntiles = p.Np / bn;
    plan.mtiles = cdiv(p.M, 128);
    OCRVI_CHECK(plan.ntiles >= 1 && plan.ntiles <= n_cu, OCRVI_EINVAL, "gemm_duo: Np=%d out of range", p.Np);
    // one persistent workgroup per CU; a workgroup keeps its column tile; row-tile lanes sized for equal (and even: two groups) counts
    int gm = std::min(plan.mtiles, std::max(1, n_cu / plan.ntiles));
    gm = cdiv(plan.mtiles, cdiv(plan.mtiles, gm));
    plan.gm = gm;
    plan.dbg = getenv("OCRVI_DUO_DBG") ? atoi(getenv("OCRVI_DUO_DBG")) : 0;
    plan.nt = (p.N_g >= 2 * p.Kp && p.Kp >= 128) ? 1 : 0;
    if (getenv("OCRVI_DUO_NT")) plan.nt = atoi(getenv("OCRVI_DUO_NT"));
    const int resk = p.res_mode == RES_NONE ? 0 : (p.res_f32 ? 1 : 2);
    if (bn == 256) return launch_duo_ni<f16x2_t, 4>(p, plan, resk, stream);
    return launch_duo_ni<f16x2_t, 3>(p, plan, resk, stream);
}

}  // namespace ocrvi
