// Shared host/device helpers for libocrvi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/ocrvi.h"

namespace ocrvi {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
#define OCRVI_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ::ocrvi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return OCRVI_EHIP;                                                                   \
        }                                                                                        \
    } while (0)
#define OCRVI_CHECK(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ::ocrvi::set_error(__VA_ARGS__);    \
            return (code);                      \
        }                                       \
    } while (0)
#define OCRVI_TRY(expr)          \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != OCRVI_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------- element types
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
// "f16x2": an fp32-equivalent element kept as TWO fp16 halves, x = hi + lo with hi = RN16(x), lo = RN16(x - hi) (|x - hi - lo| <= 2^-23 |x|
// while lo is a normal fp16 number, i.e. |x| >= 2^-3; below that the absolute error is at most 2^-25 -- weights are therefore stored
// scaled by a power of two per layer, ConvParams::wscale).  4 bytes per element, laid out per 16-byte chunk of 4 consecutive elements as
// [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]: every byte-level rule of the fp32 path (16-byte chunk = 4 channels, 128-byte K-step = 32 channels,
// XOR swizzles, LDS-DMA pieces) holds unchanged, and an MFMA operand register quartet is (hi, lo) of 4 k-slots: with both operands in
// that form   mfma16x16x32_f16(a, b) = sum hi_a hi_b + lo_a lo_b   and   mfma16x16x32_f16(swap(a), b) = sum lo_a hi_b + hi_a lo_b,
// i.e. all four partial products of 16 k-steps in two instructions on the 16-bit matrix pipe (conv_gemm's 64- and 32-column tiles use
// this chunk form); the ring GEMM, dcn_pipe, offs_conv, attention and the wide conv_gemm tiles regroup two chunks into a hi and a lo
// quartet and run THREE products per 32 k-steps, hi hi + hi lo + lo hi (Mma<f16x2_t>::three: lo lo <= 2^-22 of a product is dropped),
// accumulated in fp32.  Only ever addressed in whole chunks (Chunk<f16x2_t>) or through load_elem / store_elem.
struct f16x2_t { uint32_t raw; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct TypeInfo;
template <> struct TypeInfo<float> { static constexpr int dtype = OCRVI_F32; static constexpr int EPC = 4; };
template <> struct TypeInfo<bf16_t> { static constexpr int dtype = OCRVI_BF16; static constexpr int EPC = 8; };
template <> struct TypeInfo<f16_t> { static constexpr int dtype = OCRVI_F16; static constexpr int EPC = 8; };
template <> struct TypeInfo<f16x2_t> { static constexpr int dtype = OCRVI_F16X2; static constexpr int EPC = 4; };
template <typename T> struct IsSplit { static constexpr bool value = false; };
template <> struct IsSplit<f16x2_t> { static constexpr bool value = true; };
template <typename T> struct IsF32 { static constexpr bool value = false; };
template <> struct IsF32<float> { static constexpr bool value = true; };

static inline size_t dtype_size(int dt) { return (dt == OCRVI_F32 || dt == OCRVI_F16X2) ? 4 : 2; }
static inline bool dtype_valid(int dt) { return dt >= OCRVI_F32 && dt <= OCRVI_F16X2; }

template <typename T> __host__ __device__ inline T from_f32(float v) { return (T)v; }
template <typename T> __host__ __device__ inline float to_f32(T v) { return (float)v; }

// A 16-byte chunk of T elements <-> floats.
template <typename T> struct Chunk;  // EPC elements
template <> struct Chunk<float> {
    static constexpr int N = 4;
    __device__ static inline void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    __device__ static inline void unpack(const uint4& u, float* f) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static inline uint4 pack(const float* f) {
        union { bf16_t h[8]; uint4 u; } r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.h[i] = (bf16_t)f[i];
        return r.u;
    }
};
template <> struct Chunk<f16_t> {
    static constexpr int N = 8;
    __device__ static inline void unpack(const uint4& u, float* f) {
        union { uint4 u; f16_t h[8]; } r;
        r.u = u;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)r.h[i];
    }
    __device__ static inline uint4 pack(const float* f) {
        union { f16_t h[8]; uint4 u; } r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.h[i] = (f16_t)f[i];
        return r.u;
    }
};

// v_fma_mix_f32: an fp32 fma whose operands may be fp16 halves of a register, converted on the fly (hipcc does not select it from C++:
// it emits a v_cvt_f32_f16 per half first).  d = a * (float)half(b) + c with the LOW / HIGH half of the packed register b.
__device__ __forceinline__ float fma_mix_lo(float a, unsigned b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float a, unsigned b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <> struct Chunk<f16x2_t> {   // [4 hi | 4 lo]: u.x = (hi0, hi1), u.y = (hi2, hi3), u.z = (lo0, lo1), u.w = (lo2, lo3)
    static constexpr int N = 4;
    // acc[e] += w * (hi_e + lo_e): eight mixed-precision fmas, the fp16 halves are never converted separately (small term first)
    __device__ static inline void fma(const uint4& u, float w, float* acc) {
        acc[0] = fma_mix_lo(w, u.x, fma_mix_lo(w, u.z, acc[0]));
        acc[1] = fma_mix_hi(w, u.x, fma_mix_hi(w, u.z, acc[1]));
        acc[2] = fma_mix_lo(w, u.y, fma_mix_lo(w, u.w, acc[2]));
        acc[3] = fma_mix_hi(w, u.y, fma_mix_hi(w, u.w, acc[3]));
    }
    __device__ static inline void unpack(const uint4& u, float* f) {
        union { uint2 u; f16_t h[4]; } r;
        r.u = make_uint2(u.x, u.y);
        f[0] = fma_mix_lo(1.0f, u.z, (float)r.h[0]);
        f[1] = fma_mix_hi(1.0f, u.z, (float)r.h[1]);
        f[2] = fma_mix_lo(1.0f, u.w, (float)r.h[2]);
        f[3] = fma_mix_hi(1.0f, u.w, (float)r.h[3]);
    }
    // (the conversions as two-element vector converts: hipcc then emits v_cvt_pk_f16_f32 -- one instruction per PAIR, round to nearest even;
    // written element by element it emitted v_cvt_f16_f32 + v_cvt_f16_f32_sdwa + v_or per pair: 16 vector instructions per chunk instead of 8,
    // in epilogues that issue in the shadow of the partner wave's MFMAs)
    __device__ static inline uint4 pack(const float* f) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){f[0], f[1]}, h2v));
        const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){f[2], f[3]}, h2v));
        const float l0 = fma_mix_lo(-1.0f, h0, f[0]), l1 = fma_mix_hi(-1.0f, h0, f[1]);    // x - hi is exact in fp32
        const float l2 = fma_mix_lo(-1.0f, h1, f[2]), l3 = fma_mix_hi(-1.0f, h1, f[3]);
        const unsigned q0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){l0, l1}, h2v));
        const unsigned q1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2v){l2, l3}, h2v));
        return make_uint4(h0, h1, q0, q1);
    }
};

// ---------------------------------------------------------------- f16x2 range flag
// An f16x2 element is only as wide as fp16's exponent: |x| >= 65520 rounds its hi half to infinity (and the element to NaN).  Every kernel
// that WRITES f16x2 activations tests the fp32 values it packs (two v_max + one v_cmp per 16-byte chunk; the wave's verdict is a scalar
// mask, so the hot path has no divergent branch) and raises a per-device flag word instead of passing the value on silently; the model
// graphs copy the word to the handle after a forward and ocrvi_{det,rec}_status report OCRVI_ERANGE (include/ocrvi.h).  The pointer to
// the flag word is a per-translation-unit device variable, bound once per device by range_flag_bind() (host_util.hip) through the
// binders the OCRVI_RANGE_FLAG_TU() macro registers; an unbound pointer (null) makes the check a no-op.  (In-register operands that
// cannot leave the range of what they are built from -- attention probabilities, the deformable blend -- are packed unchecked.)
static __device__ unsigned* g_f16x2_range_flag = nullptr;
// wave mask of the lanes whose chunk holds a value fp16 cannot carry (an all-NaN chunk counts; a single NaN beside finite values does not:
// v_max drops it -- NaNs only arise downstream of an infinity, which is flagged where it is produced)
__device__ __forceinline__ unsigned long long f16x2_out_of_range(const float* f) {
    const float m = fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3])));
    return __builtin_amdgcn_ballot_w64(!(m < 65520.0f));
}
__device__ __forceinline__ void f16x2_raise(unsigned long long mask) {
    if (mask) {   // wave-uniform
        unsigned* q = g_f16x2_range_flag;
        if (q) *q = 1u;
    }
}
typedef hipError_t (*RangeFlagBinder)(unsigned*);
void range_flag_register(RangeFlagBinder fn);          // called at static-initialisation time by every translation unit with f16x2 kernels
int range_flag_bind(unsigned** flag);                  // current device: allocates the flag word once and points every unit's variable at it
// A handle's view of its device's flag: `snapshot` enqueues a copy of the device word into the handle's pinned host word at the end
// of a forward (no host synchronisation); `status` is what ocrvi_{det,rec}_status return once the caller has synchronised the stream.
struct RangeWatch {
    unsigned* dev_word = nullptr;
    unsigned* host_word = nullptr;   // pinned
    int init();                      // current device
    int snapshot(hipStream_t s);
    int status(const char* what) const;
    ~RangeWatch();
};
#define OCRVI_RANGE_FLAG_TU()                                                                                              \
    static hipError_t range_flag_binder_(unsigned* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_f16x2_range_flag), &p, sizeof(p)); } \
    static struct RangeFlagReg_ { RangeFlagReg_() { ::ocrvi::range_flag_register(&range_flag_binder_); } } range_flag_reg_;

// 4 consecutive T elements (16-byte aligned for fp32 / f16x2, 8-byte for the 16-bit types) <-> float[4]
template <typename T> __device__ __forceinline__ void load4(const T* p, float* f) {
    if constexpr (IsSplit<T>::value) {
        Chunk<T>::unpack(*(const uint4*)p, f);
    } else if constexpr (sizeof(T) == 4) {
        const float4 v = *(const float4*)p;
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        union { uint2 u; T h[4]; } r;
        r.u = *(const uint2*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)r.h[i];
    }
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* f) {
    if constexpr (IsSplit<T>::value) {
        f16x2_raise(f16x2_out_of_range(f));
        *(uint4*)p = Chunk<T>::pack(f);
    } else if constexpr (sizeof(T) == 4) {
        *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
    } else {
        union { uint2 u; T h[4]; } r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.h[i] = (T)f[i];
        *(uint2*)p = r.u;
    }
}

// One element by index (test taps, the three-key FRM kernel): plain types index the array, f16x2 picks its two halves out of the chunk.
template <typename T> __device__ __forceinline__ float load_elem(const T* p, size_t i) { return to_f32<T>(p[i]); }
template <> __device__ __forceinline__ float load_elem<f16x2_t>(const f16x2_t* p, size_t i) {
    const f16_t* h = (const f16_t*)(p + (i & ~(size_t)3));
    return (float)h[i & 3] + (float)h[4 + (i & 3)];
}
template <typename T> __device__ __forceinline__ void store_elem(T* p, size_t i, float v) { p[i] = from_f32<T>(v); }
template <> __device__ __forceinline__ void store_elem<f16x2_t>(f16x2_t* p, size_t i, float v) {
    f16_t* h = (f16_t*)(p + (i & ~(size_t)3));
    f16x2_raise(__builtin_amdgcn_ballot_w64(!(fabsf(v) < 65520.0f)));
    const f16_t hi = (f16_t)v;
    h[i & 3] = hi;
    h[4 + (i & 3)] = (f16_t)(v - (float)hi);
}

// ---------------------------------------------------------------- MFMA wrappers (16x16 output tile)
// A fragment is the 32 bytes lane (r = lane&15, g = lane>>4) reads from row r of a [rows][128 B] LDS tile
// at byte offset 32*g.  The K order inside a 128-byte K-step is therefore permuted identically for both
// operands, which a dot product does not care about.
template <typename T> struct Mma;
template <> struct Mma<float> {
    __device__ static inline void half(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
    __device__ static inline void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x4& c) {
        const uint32_t aw[8] = {a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w};
        const uint32_t bw[8] = {b[0].x, b[0].y, b[0].z, b[0].w, b[1].x, b[1].y, b[1].z, b[1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(aw[j]), __uint_as_float(bw[j]), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    __device__ static inline void half(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static inline void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x4& c) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[s]), __builtin_bit_cast(bf16x8, b[s]), c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    __device__ static inline void half(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static inline void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x4& c) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[s]), __builtin_bit_cast(f16x8, b[s]), c, 0, 0, 0);
    }
};

template <> struct Mma<f16x2_t> {   // operands are chunks [4 hi | 4 lo]: hh + ll, then lh + hl with the halves of `a` exchanged
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    __device__ static inline void half(const uint4& a, const uint4& b, f32x4& c) {
        // (register vectors, not uint4 structs: with the structs hipcc kept fragment arrays in scratch memory to form the swapped operand)
        const u4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
        const u4v as = __builtin_shufflevector(av, av, 2, 3, 0, 1);
#ifndef OCRVI_TIMING_HALF_MFMA   /* timing experiment only (results are wrong): what the kernels would cost with half the matrix work */
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, as), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
#endif
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
    }
    __device__ static inline void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x4& c) {
        half(a[0], b[0], c);
        half(a[1], b[1], c);
    }
    // THREE-PRODUCT FORM.  A lane's operand for one 128-byte K-step is two chunks, c0 = [hi | lo] of k-group 2g and c1 = [hi | lo] of k-group
    // 2g + 1.  Regrouped in registers into H = (hi of both groups) and L = (lo of both groups) -- a permutation of the lane's 8 k-slots that
    // is applied to both operands alike -- the product needs only  wL.xH + wH.xL + wH.xH : three MFMAs per 32 k-steps instead of four (the
    // lo.lo term, 2^-22 of the product at most, is dropped; tools/split_probe.hip prices it).  No memory layout changes.
    __device__ static inline void regroup(const uint4& c0, const uint4& c1, u4v& H, u4v& L) {
        H = (u4v){c0.x, c0.y, c1.x, c1.y};
        L = (u4v){c0.z, c0.w, c1.z, c1.w};
    }
    // (Round 4 had an in-place form of this -- two inline-asm v_swap_b32 -- to save the copies' registers.  Pinned directly in front of the MFMAs
    // that read the swapped registers it produced wrong results in conv_gemm (hipcc places the wait states between a vector write and the matrix
    // instruction that reads it only for instructions it can see, as with the attention kernel's inline-asm conversions behind v_exp_f32): removed.)
    __device__ static inline u4v as_u4v(const uint4& c) { return (u4v){c.x, c.y, c.z, c.w}; }
    __device__ static inline void three(const u4v& wH, const u4v& wL, const u4v& xH, const u4v& xL, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wL), __builtin_bit_cast(f16x8, xH), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wH), __builtin_bit_cast(f16x8, xL), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wH), __builtin_bit_cast(f16x8, xH), c, 0, 0, 0);
    }
    // the same with the half-swapped form of `b` supplied by the caller (a register-resident operand that meets many `a` fragments is
    // swapped once instead of swapping every `a`): a.b = hh + ll, a.swap(b) = hl + lh
    __device__ static inline uint4 swapped(const uint4& b) { return make_uint4(b.z, b.w, b.x, b.y); }
    __device__ static inline void pair(const uint4& a, const uint4& b, const uint4& bs, f32x4& c) {
        const u4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w}, sv = {bs.x, bs.y, bs.z, bs.w};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, sv), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
    }
};

// Swizzle for [rows][128 B] LDS tiles read as (row = base + lane&15, chunks 2g, 2g+1) with ds_read_b128:
// conflict-free under the gfx950 lane-group / 64-bank rule (checked by simulation).
__device__ __forceinline__ int swz128(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }

// Exact (erf) GELU = x * Phi(x) with erfc by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off level):
//   z = |x| / sqrt(2),  t = 1 / (1 + p z),  erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),
//   Phi(x) = 1 - erfc(z) / 2 for x >= 0, erfc(z) / 2 otherwise.
// One v_rcp_f32 + one v_exp_f32 (both 1 ulp) + 6 fma: libm erff's polynomial ladder and a correctly rounded 1/x (v_div_scale /
// v_div_fmas / v_div_fixup) each made the GELU epilogue cost several MFMA main loops.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float q = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    q = fmaf(q, t, 0.5f * 1.421413741f);
    q = fmaf(q, t, 0.5f * -0.284496736f);
    q = fmaf(q, t, 0.5f * 0.254829592f);
    const float h = q * t * __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));  // erfc(z) / 2
    return fmaxf(x, 0.f) - ax * h;   // x (1 - h) for x >= 0, x h = -|x| h otherwise
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// XCD-aware block remap (cdna_hip_programming.md T1, bijective form): consecutive logical tiles land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------- workspace arena
// Bump allocator over the caller's workspace.  In planning mode (base == nullptr) it only measures.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool overflow = false;
    Arena(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
    bool planning() const { return base == nullptr; }
    void* alloc(size_t bytes) {
        off = align_up(off, 256);
        size_t o = off;
        off += bytes;
        if (off > peak) peak = off;
        if (planning()) return (void*)(uintptr_t)256;  // never dereferenced
        if (off > cap) { overflow = true; return nullptr; }
        return base + o;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

// ---------------------------------------------------------------- weight blob (host)
struct BlobTensor {
    std::string name;
    int ndim = 0;
    int dims[4] = {1, 1, 1, 1};
    const float* data = nullptr;
    size_t count = 0;
};
struct Blob {
    std::vector<BlobTensor> tensors;
    int parse(const void* blob, size_t bytes);
    const BlobTensor* find(const std::string& name) const;
    // fetch with shape check; dims<0 = wildcard
    int get(const std::string& name, int d0, int d1, int d2, int d3, const BlobTensor** out) const;
};

// ---------------------------------------------------------------- per-launch HIP-event profiler (off by default)
// When enabled (ocrvi_prof_enable), every instrumented launch is bracketed by two events recorded on the launch
// stream; ocrvi_prof_report aggregates elapsed time, algorithmic FLOPs and algorithmic bytes per kernel tag.
bool prof_enabled();
struct ProfScope {
    int slot = -1;
    hipStream_t stream;
    ProfScope(const char* tag, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};
static inline const char* dtype_name(int dt) { return dt == OCRVI_F32 ? "f32" : (dt == OCRVI_BF16 ? "bf16" : (dt == OCRVI_F16 ? "f16" : "f16x2")); }

// ---------------------------------------------------------------- per-device launch state
// CU count of the CURRENT device (cached per device id) and a once-per-(kernel, device) opt-in to > 64 KiB of dynamic LDS.
int device_cus(int* n_cu);
int ensure_max_smem(const void* kernel, int bytes);
// Makes the handle's device current for the duration of a call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Host fp32 -> T conversion into a byte buffer.
// (f16x2: src is multiplied by `scale` first and n must be a multiple of 4)
void convert_to_dtype(const float* src, size_t n, int dtype, void* dst, float scale = 1.f);

}  // namespace ocrvi
