// Host utilities: error string, blob parsing, dtype conversion, weight packing, dtype dispatch.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "conv_gemm.h"
#include "gemm_ring.h"
#include "gemm_duo.h"
#include "dcn_pipe.h"

namespace ocrvi {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error_cstr() { return g_err; }

// ------------------------------------------------------------------ blob
#pragma pack(push, 1)
struct BlobEntry {
    char name[64];
    uint32_t ndim;
    uint32_t dims[4];
    uint64_t offset;
    uint64_t count;
};
#pragma pack(pop)
static_assert(sizeof(BlobEntry) == 100, "entry layout");

int Blob::parse(const void* blob, size_t bytes) {
    OCRVI_CHECK(blob && bytes >= 16 && memcmp(blob, "OCRVIW1\0", 8) == 0, OCRVI_EBLOB, "weight blob: bad magic / too short");
    uint32_t n;
    memcpy(&n, (const char*)blob + 8, 4);
    OCRVI_CHECK(16 + (size_t)n * sizeof(BlobEntry) <= bytes, OCRVI_EBLOB, "weight blob: truncated table");
    tensors.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        BlobEntry e;
        memcpy(&e, (const char*)blob + 16 + (size_t)i * sizeof(BlobEntry), sizeof(e));
        e.name[63] = 0;
        OCRVI_CHECK(e.ndim <= 4 && e.offset % 4 == 0 && e.offset + e.count * 4 <= bytes, OCRVI_EBLOB,
                    "weight blob: tensor %s out of bounds", e.name);
        size_t prod = 1;
        for (uint32_t d = 0; d < e.ndim; ++d) prod *= e.dims[d];
        OCRVI_CHECK(prod == e.count, OCRVI_EBLOB, "weight blob: tensor %s count mismatch", e.name);
        BlobTensor& t = tensors[i];
        t.name = e.name;
        t.ndim = (int)e.ndim;
        for (int d = 0; d < 4; ++d) t.dims[d] = d < (int)e.ndim ? (int)e.dims[d] : 1;
        t.data = (const float*)((const char*)blob + e.offset);
        t.count = e.count;
    }
    return OCRVI_OK;
}
const BlobTensor* Blob::find(const std::string& name) const {
    for (const auto& t : tensors)
        if (t.name == name) return &t;
    return nullptr;
}
int Blob::get(const std::string& name, int d0, int d1, int d2, int d3, const BlobTensor** out) const {
    const BlobTensor* t = find(name);
    OCRVI_CHECK(t, OCRVI_EBLOB, "weight blob: missing tensor '%s'", name.c_str());
    const int want[4] = {d0, d1, d2, d3};
    for (int d = 0; d < 4; ++d) {
        const int have = d < t->ndim ? t->dims[d] : 1;
        OCRVI_CHECK(want[d] < 0 || have == (want[d] == 0 ? 1 : want[d]), OCRVI_EBLOB,
                    "weight blob: tensor '%s' dim %d is %d, expected %d", name.c_str(), d, have, want[d]);
    }
    *out = t;
    return OCRVI_OK;
}

// ------------------------------------------------------------------ dtype conversion (host, round-to-nearest-even)
static inline uint16_t f32_to_bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
void convert_to_dtype(const float* src, size_t n, int dtype, void* dst, float scale) {
    if (dtype == OCRVI_F32) {
        memcpy(dst, src, n * 4);
    } else if (dtype == OCRVI_F16X2) {   // chunks of 4 elements: [4 hi | 4 lo], x * scale = hi + lo (n % 4 == 0: rows are padded to 32)
        _Float16* d = (_Float16*)dst;
        for (size_t i = 0; i < n; ++i) {
            const float x = src[i] * scale;
            const _Float16 hi = (_Float16)x;
            const size_t c = i >> 2, j = i & 3;
            d[8 * c + j] = hi;
            d[8 * c + 4 + j] = (_Float16)(x - (float)hi);
        }
    } else if (dtype == OCRVI_BF16) {
        uint16_t* d = (uint16_t*)dst;
        for (size_t i = 0; i < n; ++i) d[i] = f32_to_bf16_bits(src[i]);
    } else {
        _Float16* d = (_Float16*)dst;
        for (size_t i = 0; i < n; ++i) d[i] = (_Float16)src[i];
    }
}

// ------------------------------------------------------------------ weight packing
static PackedConv finish_pack(const std::vector<float>& wt, int groups, int Np, int Kp, int dtype, bool quartets = false) {
    PackedConv pc;
    pc.bytes.resize(wt.size() * dtype_size(dtype));
    float scale = 1.f;
    if (dtype == OCRVI_F16X2) {
        // one power of two per layer that puts the largest |w| into [2^13, 2^14): far from fp16's 65504, and every weight within 2^-17
        // of the largest keeps a NORMAL lo half (full 2^-23 relative accuracy; smaller ones are off by at most 2^-39 of the largest).
        // The kernels' epilogues multiply the accumulator by the inverse, which is exact.
        float mx = 0.f;
        for (float v : wt) mx = std::max(mx, fabsf(v));
        if (mx > 0.f && std::isfinite(mx)) {
            int e = 0;
            (void)frexpf(mx, &e);                       // mx in [2^(e-1), 2^e)
            scale = ldexpf(1.0f, std::min(std::max(14 - e, -100), 100));
        }
        pc.wscale = 1.0f / scale;
    }
    convert_to_dtype(wt.data(), wt.size(), dtype, pc.bytes.data(), scale);
    if (quartets && dtype == OCRVI_F16X2) {
        // (hi, lo) quartet form for kernels that read a lane's 8 k-slots as two 16-byte operands without regrouping them in registers
        // (dcn_pipe.h): every 32-byte group of 8 consecutive k-slots [hi4 lo4 | hi4' lo4'] becomes [hi4 hi4' | lo4 lo4']
        uint64_t* q = (uint64_t*)pc.bytes.data();
        for (size_t i = 0; i + 3 < pc.bytes.size() / 8; i += 4) std::swap(q[i + 1], q[i + 2]);
    }
    pc.Np = Np;
    pc.Kp = Kp;
    pc.groups = groups;
    return pc;
}

PackedConv pack_conv(const float* w, const float* bias, int cout, int cin_g, int kh, int kw, int groups, int amode, int dtype) {
    const int bke = conv_bke(dtype);
    const int n_g = cout / groups;
    const int bn = conv_bn_for(n_g);
    const int Np = cdiv(n_g, bn) * bn;
    int Kp;
    if (amode == AM_ROWS) Kp = cdiv(kh * 32, bke) * bke;
    else Kp = cdiv(kh * kw * cin_g, bke) * bke;
    std::vector<float> wt((size_t)groups * Np * Kp, 0.f);
    for (int g = 0; g < groups; ++g)
        for (int n = 0; n < n_g; ++n) {
            const float* src = w + (size_t)(g * n_g + n) * cin_g * kh * kw;
            float* dst = wt.data() + ((size_t)g * Np + n) * Kp;
            for (int c = 0; c < cin_g; ++c)
                for (int r = 0; r < kh; ++r)
                    for (int s = 0; s < kw; ++s) {
                        const float v = src[(c * kh + r) * kw + s];
                        if (amode == AM_ROWS) dst[r * 32 + s * 4 + c] = v;          // [filter row][pixel s][ch c of 4]
                        else if (amode == AM_DCN && dcn_pipe_packing(dtype, cin_g)) {
                            const int cb = dcn_pipe_block(dtype);                                          // [channel block][tap][ch] (dcn_pipe.h)
                            dst[((c / cb) * kh * kw + r * kw + s) * cb + (c % cb)] = v;
                        }
                        else dst[(r * kw + s) * cin_g + c] = v;                      // [tap][cin]
                    }
        }
    PackedConv pc = finish_pack(wt, groups, Np, Kp, dtype, amode == AM_DCN && dcn_pipe_packing(dtype, cin_g));
    pc.N_g = n_g;
    pc.Cin_g = cin_g;
    pc.KH = kh;
    if (bias) pc.bias.assign(bias, bias + cout);
    return pc;
}

PackedConv pack_deconv2(const float* const* w, const float* const* bias, int groups, int cin, int cout, int dtype) {
    const int bke = conv_bke(dtype);
    const int n_g = 4 * cout;
    const int bn = conv_bn_for(n_g);
    const int Np = cdiv(n_g, bn) * bn, Kp = cdiv(cin, bke) * bke;
    std::vector<float> wt((size_t)groups * Np * Kp, 0.f);
    for (int g = 0; g < groups; ++g)
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co)
                for (int ab = 0; ab < 4; ++ab) wt[((size_t)g * Np + ab * cout + co) * Kp + ci] = w[g][((size_t)ci * cout + co) * 4 + ab];
    PackedConv pc = finish_pack(wt, groups, Np, Kp, dtype);   // (one f16x2 weight scale for all groups: they share a launch)
    pc.N_g = n_g;
    pc.Cin_g = cin;
    pc.bias.resize((size_t)groups * n_g);
    for (int g = 0; g < groups; ++g)
        for (int ab = 0; ab < 4; ++ab)
            for (int co = 0; co < cout; ++co) pc.bias[(size_t)g * n_g + ab * cout + co] = (bias && bias[g]) ? bias[g][co] : 0.f;
    return pc;
}

// ------------------------------------------------------------------ profiler
namespace {
struct ProfEntry { std::string tag; double flops, bytes; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfEntry> g_prof;
}  // namespace
bool prof_enabled() { return g_prof_on; }
ProfScope::ProfScope(const char* tag, double flops, double bytes, hipStream_t s) : stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfEntry e;
    e.tag = tag; e.flops = flops; e.bytes = bytes;
    if (hipEventCreate(&e.e0) != hipSuccess || hipEventCreate(&e.e1) != hipSuccess) return;
    (void)hipEventRecord(e.e0, s);
    g_prof.push_back(e);
    slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].e1, stream);
}

static const char* amode_name(int amode, const ConvParams& p) {
    switch (amode) {
        case AM_CONV1: return p.store_mode == ST_SHUFFLE2 ? "deconv2x2" : (p.store_mode == ST_DB_TAIL ? "deconv2x2_dbtail" : "conv1x1");
        case AM_CONV3: return p.store_mode == ST_DCN_OFFS ? "dcn_offset_conv3x3" : (p.groups > 1 ? "gconv3x3" : "conv3x3");
        case AM_ROWS: return "stem_conv";
        case AM_DCN: return "dcn3x3";
    }
    return "conv";
}

bool gemm_ring_eligible(const ConvParams& p, int amode, int dtype) {
    static const bool on = !(getenv("OCRVI_GEMM_RING") && atoi(getenv("OCRVI_GEMM_RING")) == 0);
    static const bool on3 = !(getenv("OCRVI_RING_CONV3") && atoi(getenv("OCRVI_RING_CONV3")) == 0);
    if (!on || p.groups != 1 || p.store_mode != ST_NHWC) return false;
    const int esz = (int)dtype_size(dtype), bke = conv_bke(dtype);
    if (amode == AM_CONV3) {  // 3x3 / stride 1 / pad 1 on the ring: 16-bit types, whole channel blocks per tap, 256-row tiles
        static const bool x2c3 = getenv("OCRVI_RING_CONV3_X2") && atoi(getenv("OCRVI_RING_CONV3_X2"));   // experiment: f16x2 3x3 on the ring's 128-row build
        if (!on3 || !(esz == 2 || (x2c3 && dtype == OCRVI_F16X2)) || (p.out_f32 && esz == 2) || p.KH != 3 || p.SH != 1 || p.SW != 1 || p.PH != 1 || p.PW != 1 || p.H != p.OH || p.W != p.OW) return false;
        // measured (profiles/r01_conv_variants.md): +5..9 % over conv_gemm at M >= 300 k rows, -12 % at 77 k (too few tiles per CU)
        static const int min_m = getenv("OCRVI_RING_CONV3_MIN_M") ? atoi(getenv("OCRVI_RING_CONV3_MIN_M")) : (1 << 18);
        if (p.Cin_g % bke != 0 || p.Kp != 9 * p.Cin_g || p.M < min_m || p.Np % 128 != 0) return false;  // (64-wide: conv_gemm is 6 % faster)
    } else {  // 1x1, stride 1, no padding: output pixel == input pixel (identity_pix, which launch_conv clears for RES_UP2 only)
        if (amode != AM_CONV1 || p.Kp != p.Cin_g || p.PH != 0 || p.PW != 0) return false;
        const bool unit = p.SH == 1 && p.SW == 1 && p.H == p.OH && p.W == p.OW;
        if (unit) {
            if (!p.identity_pix && p.res_mode != RES_UP2) return false;
            if (p.res_mode == RES_UP2 && ((p.OH | p.OW) & 1)) return false;
        } else {  // strided 1x1 (downsample): per-lane 32-bit offsets from the tensor base
            if (p.res_mode != RES_NONE || p.OH != (p.H - 1) / p.SH + 1 || p.OW != (p.W - 1) / p.SW + 1) return false;
            if ((unsigned long long)p.n_img * p.H * p.W * p.Cin * esz >= (1ull << 32)) return false;
        }
    }
    if (p.Cin_g % bke != 0 || p.N_g % 4 != 0) return false;
    if (p.Np % 128 != 0 && !(p.Np == 64 && p.N_g == 64 && esz == 2 && !p.out_f32)) return false;  // 128-wide column tiles, or one 64-wide
    if (p.N_g < 64) return false;
    if (((size_t)p.cin_off * esz) % 16 != 0 || ((size_t)p.Cin * esz) % 16 != 0 || ((uintptr_t)p.x & 15) != 0) return false;
    // 16-byte epilogue accesses: 4 fp32 or 8 16-bit channels
    const bool of32 = p.out_f32 || esz == 4, rf32 = p.res_f32 || esz == 4;
    const int og = of32 ? 4 : 8;
    if (p.N_g % og != 0 || p.ldo % og != 0 || p.out_coff % og != 0 || ((uintptr_t)p.out & 15) != 0) return false;
    if (p.res_mode != RES_NONE && (rf32 != of32 || p.ldr % og != 0 || ((uintptr_t)p.res & 15) != 0)) return false;
    if (p.bias && ((uintptr_t)p.bias & 15) != 0) return false;
    // the epilogue addresses the output with 32-bit byte offsets through a buffer descriptor
    if (((size_t)(p.M - 1) * p.ldo + p.out_coff + p.N_g) * (of32 ? 4 : 2) >= ((size_t)1 << 32) - 65536) return false;
    // ... and the residual with 32-bit byte offsets from its base (it has at most the output's rows)
    if (p.res_mode != RES_NONE && ((size_t)(p.M - 1) * p.ldr + p.N_g) * (rf32 ? 4 : 2) >= ((size_t)1 << 32) - 65536) return false;
    return true;
}

// The duo ring GEMM (gemm_duo.h): f16x2 1x1 convolutions / Linears that gemm_ring would take, whose column count tiles by 256 or 192,
// with at least four K-steps (below that the layers sit on their HBM roof in either kernel) and an epilogue variant that is built
// (duo_f16x2.hip: launch_duo_ni).  The choice depends on the layer only (N, K, epilogue), never on M, so a page alone and the same page
// inside a batch take the same kernel.  OFF by default (OCRVI_GEMM_DUO=1 enables it, read once per process): measured end to end it
// ties gemm_ring (260.1 vs 260.5 invoices/s) -- 12-20 % faster at K <= 384 with N >= 1024 and no GELU, equal at long K, slower with GELU;
// why (a wave alone on its SIMD drives the matrix pipe at about half the rate two waves reach together, so the "solo" MFMA steps beside
// the other group's epilogue are no shorter than shared ones) is measured in profiles/r04_duo.md with tools/mfma_mix.hip.
bool gemm_duo_eligible(const ConvParams& p, int amode, int dtype) {
    // OCRVI_GEMM_DUO: unset / 0 never; 1 every shape the kernel takes (ties gemm_ring end to end: profiles/r04_duo.md); 2 only wide outputs on a
    // short K with a plain epilogue -- the recogniser's qkv projections (N = 3 D >= 768, K = D <= 384): 17.6 against 18.6 ms per step there,
    // nothing measurable end to end (293-297 invoices/s either way).  The kernels are bit-identical, so the choice is invisible in the results.
    static const int mode = getenv("OCRVI_GEMM_DUO") ? atoi(getenv("OCRVI_GEMM_DUO")) : 0;
    if (mode == 0 || dtype != OCRVI_F16X2 || amode != AM_CONV1) return false;
    if (!gemm_ring_eligible(p, amode, dtype)) return false;
    if (mode == 2 && !(p.Np >= 768 && p.Kp <= 384 && p.act == ACT_NONE && p.res_mode == RES_NONE && !p.out_f32 && p.SH == 1 && p.SW == 1)) return false;
    static const int min_nk = getenv("OCRVI_DUO_MIN_NK") ? atoi(getenv("OCRVI_DUO_MIN_NK")) : 4;
    if (duo_bn_for(p.Np) == 0 || p.Kp / 32 < min_nk) return false;
    const int resk = p.res_mode == RES_NONE ? 0 : (p.res_f32 ? 1 : 2);
    const bool of32 = p.out_f32 != 0;
    // (GELU epilogues stay on gemm_ring: ~80 VALU instructions per fragment spread over the next tile's MFMA steps by the same wave beat
    // epilogue steps beside the other group's MFMAs -- 326 vs 390 us at M 61440, K 384, N 1536; OCRVI_DUO_GELU=1 routes them here anyway)
    static const bool gelu = getenv("OCRVI_DUO_GELU") && atoi(getenv("OCRVI_DUO_GELU"));
    if (p.act == ACT_GELU && !gelu) return false;
    if (resk == 0) return of32 ? p.act == ACT_NONE : true;
    if (resk == 2) return !of32 && p.act != ACT_GELU;
    return of32 && p.act == ACT_NONE;     // raw fp32 residual stream: fp32 output
}

int device_cus(int* n_cu) {
    static std::mutex mu;
    static std::map<int, int> cus;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    OCRVI_HIP(hipGetDevice(&dev));
    auto it = cus.find(dev);
    if (it == cus.end()) {
        hipDeviceProp_t prop;
        OCRVI_HIP(hipGetDeviceProperties(&prop, dev));
        it = cus.emplace(dev, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256).first;
    }
    *n_cu = it->second;
    return OCRVI_OK;
}

int ensure_max_smem(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;   // (kernel, device) -> largest size already granted
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    OCRVI_HIP(hipGetDevice(&dev));
    auto key = std::make_pair(kernel, dev);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return OCRVI_OK;
    OCRVI_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[key] = bytes;
    return OCRVI_OK;
}

int ring_pages(const void** zero_page, void** dump_page) {
    // one 8 KiB device allocation per process and device: [0, 4096) zeros, [4096, 8192) write sink.  Never freed.
    static std::mutex mu;
    static std::map<int, char*> pages;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    OCRVI_HIP(hipGetDevice(&dev));
    auto it = pages.find(dev);
    if (it == pages.end()) {
        char* p = nullptr;
        OCRVI_HIP(hipMalloc((void**)&p, 8192));
        OCRVI_HIP(hipMemset(p, 0, 8192));
        it = pages.emplace(dev, p).first;
    }
    *zero_page = it->second;
    *dump_page = it->second + 4096;
    return OCRVI_OK;
}

// ---- f16x2 range flag (common.h): one device word per device, every translation unit's pointer variable bound to it once
static std::vector<RangeFlagBinder>& range_flag_binders() {
    static std::vector<RangeFlagBinder> v;   // (function-local: filled during static initialisation of the other units)
    return v;
}
void range_flag_register(RangeFlagBinder fn) { range_flag_binders().push_back(fn); }
int range_flag_bind(unsigned** flag) {
    static std::mutex mu;
    static std::map<int, unsigned*> words;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    OCRVI_HIP(hipGetDevice(&dev));
    auto it = words.find(dev);
    if (it == words.end()) {
        unsigned* w = nullptr;
        OCRVI_HIP(hipMalloc((void**)&w, 256));
        OCRVI_HIP(hipMemset(w, 0, 256));
        for (RangeFlagBinder fn : range_flag_binders()) OCRVI_HIP(fn(w));
        OCRVI_HIP(hipDeviceSynchronize());
        it = words.emplace(dev, w).first;
    }
    *flag = it->second;
    return OCRVI_OK;
}

int RangeWatch::init() {
    OCRVI_TRY(range_flag_bind(&dev_word));
    OCRVI_HIP(hipHostMalloc((void**)&host_word, 64, hipHostMallocDefault));
    *host_word = 0;
    return OCRVI_OK;
}
int RangeWatch::snapshot(hipStream_t s) {
    if (!dev_word) return OCRVI_OK;
    OCRVI_HIP(hipMemcpyAsync(host_word, dev_word, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return OCRVI_OK;
}
int RangeWatch::status(const char* what) const {
    OCRVI_CHECK(!host_word || *(volatile unsigned*)host_word == 0, OCRVI_ERANGE,
                "%s: an f16x2 activation left fp16's range (|x| >= 65520) on this device since the last ocrvi_range_reset: the results of that "
                "forward are not valid -- run this model with dtype f32 (or f16x2 after rescaling the offending layer)", what);
    return OCRVI_OK;
}
RangeWatch::~RangeWatch() {
    if (host_word) (void)hipHostFree(host_word);
}
extern "C" int ocrvi_range_reset(int device, void* stream) {
    DeviceGuard dg(device);
    OCRVI_HIP(dg.err);
    unsigned* w = nullptr;
    OCRVI_TRY(range_flag_bind(&w));
    OCRVI_HIP(hipMemsetAsync(w, 0, sizeof(unsigned), (hipStream_t)stream));
    return OCRVI_OK;
}
extern "C" int ocrvi_range_flag(int device, int* raised) {
    OCRVI_CHECK(raised, OCRVI_EINVAL, "range_flag: null out");
    DeviceGuard dg(device);
    OCRVI_HIP(dg.err);
    unsigned* w = nullptr;
    unsigned v = 0;
    OCRVI_TRY(range_flag_bind(&w));
    OCRVI_HIP(hipDeviceSynchronize());
    OCRVI_HIP(hipMemcpy(&v, w, sizeof(v), hipMemcpyDeviceToHost));
    *raised = v != 0;
    return OCRVI_OK;
}

int launch_conv_dt(int dtype, const ConvParams& p, int amode, hipStream_t stream) {
    char tag[160];
    double flops = 0, bytes = 0;
    if (g_prof_on) {
        const int ks = amode == AM_CONV1 ? 1 : (amode == AM_ROWS ? p.KH : 3);
        const double kvalid = amode == AM_ROWS ? (double)p.KH * p.KH * 3 : (double)ks * ks * p.Cin_g;
        const double esz = (double)dtype_size(dtype);
        flops = 2.0 * p.M * p.N_g * p.groups * kvalid;
        // algorithmic bytes: input read once, weights once, output written once (+ residual / offsets read once)
        bytes = (double)p.n_img * p.H * p.W * (amode == AM_ROWS ? 4 : p.Cin_g * p.groups) * esz + (double)p.N_g * p.groups * kvalid * esz +
                (double)p.M * p.N_g * p.groups * (p.out_f32 ? 4.0 : esz) + (p.res ? (double)p.M * p.N_g * p.groups * (p.res_f32 ? 4.0 : esz) : 0.0) +
                (p.offs ? (double)p.M * 27 * 4 : 0.0);
        ConvParams q = p;   // identity_pix is set inside launch_conv; recompute it here for the tag only
        q.identity_pix = (amode == AM_CONV1 && p.SH == 1 && p.SW == 1 && p.PH == 0 && p.PW == 0 && p.H == p.OH && p.W == p.OW &&
                          p.store_mode == ST_NHWC && p.res_mode != RES_UP2) ? 1 : 0;
        const bool ring = gemm_ring_eligible(q, amode, dtype);
        const bool pipe = amode == AM_DCN && dcn_pipe_eligible(q, dtype);
        static const bool detail = getenv("OCRVI_PROF_DETAIL") != nullptr;
        const bool duo = gemm_duo_eligible(q, amode, dtype);
        if (duo && !detail)
            snprintf(tag, sizeof(tag), "gemm_duo_%s", dtype_name(dtype));
        else if (ring && !detail)
            snprintf(tag, sizeof(tag), "gemm_ring_%s", dtype_name(dtype));
        else if (pipe && !detail)
            snprintf(tag, sizeof(tag), "dcn3x3_pipe128x%d_%s", p.Np % 256 == 0 ? 256 : 128, dtype_name(dtype));
        else if (pipe)
            snprintf(tag, sizeof(tag), "dcn3x3_pipe128x%d_%s M%d N%d K%d g%d s%d", p.Np % 256 == 0 ? 256 : 128, dtype_name(dtype), p.M, p.N_g, (int)kvalid,
                     p.groups, p.SH);
        else if (detail)
            snprintf(tag, sizeof(tag), "%s%s_%dx%d_%s M%d N%d K%d g%d s%d", ring ? "ring_" : "", amode_name(amode, p), amode == AM_DCN ? 64 : 128, (amode == AM_DCN && p.Np % 256 == 0) ? 256 : conv_bn_for(p.N_g), dtype_name(dtype), p.M,
                     p.N_g, (int)kvalid, p.groups, p.SH);
        else
            snprintf(tag, sizeof(tag), "%s%s_%dx%d_%s", ring ? "ring_" : "", amode_name(amode, p), amode == AM_DCN ? 64 : 128, (amode == AM_DCN && p.Np % 256 == 0) ? 256 : conv_bn_for(p.N_g), dtype_name(dtype));
    }
    ProfScope ps(tag, flops, bytes, stream);
    switch (dtype) {
        case OCRVI_F32: return launch_conv<float>(p, amode, stream);
        case OCRVI_BF16: return launch_conv<bf16_t>(p, amode, stream);
        case OCRVI_F16: return launch_conv<f16_t>(p, amode, stream);
        case OCRVI_F16X2: return launch_conv<f16x2_t>(p, amode, stream);
    }
    set_error("unknown dtype %d", dtype);
    return OCRVI_EINVAL;
}

}  // namespace ocrvi

extern "C" int ocrvi_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(ocrvi::g_prof_mu);
    ocrvi::g_prof_on = on != 0;
    return OCRVI_OK;
}
extern "C" int ocrvi_prof_reset(void) {
    std::lock_guard<std::mutex> lk(ocrvi::g_prof_mu);
    for (auto& e : ocrvi::g_prof) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    ocrvi::g_prof.clear();
    return OCRVI_OK;
}
extern "C" int ocrvi_prof_report(char* buf, size_t cap) {
    using namespace ocrvi;
    OCRVI_CHECK(buf && cap > 2, OCRVI_EINVAL, "prof_report: null buffer");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    struct Agg { double ms = 0, flops = 0, bytes = 0; long n = 0; };
    std::map<std::string, Agg> agg;
    for (auto& e : g_prof) {
        OCRVI_HIP(hipEventSynchronize(e.e1));
        float ms = 0.f;
        OCRVI_HIP(hipEventElapsedTime(&ms, e.e0, e.e1));
        Agg& a = agg[e.tag];
        a.ms += ms; a.flops += e.flops; a.bytes += e.bytes; a.n += 1;
    }
    std::string js = "{";
    bool first = true;
    for (auto& kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
                 kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        js += line;
        first = false;
    }
    js += "}";
    OCRVI_CHECK(js.size() + 1 <= cap, OCRVI_ENOMEM, "prof_report: buffer too small (%zu needed)", js.size() + 1);
    memcpy(buf, js.c_str(), js.size() + 1);
    return OCRVI_OK;
}

// Test hook (host only, no GPU): the f16x2 weight packer's element format and power-of-two scale.  src [n] fp32 (n % 4 == 0) -> dst [n]
// 4-byte elements in chunks of [4 hi | 4 lo]; *wscale = what the kernels' epilogues multiply the accumulator by (1 / the storage scale).
extern "C" int ocrvi_test_pack_f16x2(const float* src, size_t n, void* dst, float* wscale) {
    using namespace ocrvi;
    OCRVI_CHECK(src && dst && wscale && n % 4 == 0, OCRVI_EINVAL, "test_pack_f16x2: bad argument");
    std::vector<float> wt(src, src + n);
    PackedConv pc = finish_pack(wt, 1, 1, (int)n, OCRVI_F16X2);
    memcpy(dst, pc.bytes.data(), pc.bytes.size());
    *wscale = pc.wscale;
    return OCRVI_OK;
}

extern "C" const char* ocrvi_last_error(void) { return ocrvi::last_error_cstr(); }
extern "C" int ocrvi_abi_version(void) { return 2; }
