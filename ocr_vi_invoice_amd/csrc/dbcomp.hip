// Device half of DB post-processing (the reference thresholds on the host, `pred[0] > self.thresh`,
// src/det/test.py:57, after copying the whole 4.9-MB map, pipeline2.py:320): threshold -> 1-bit mask -> 8-connected component
// labelling (the connectivity cv2.findContours follows for the foreground) -> per-component bounding box / pixel count / probability
// sum -> the probability values inside the component boxes packed into one compact buffer.  Only the bit mask (W*H/8 bytes), the
// component table and the packed values cross PCIe; the host (dbpost.hip) rebuilds a sparse map from them and runs the same
// contour / polygon / score / unclip code, so its results are those of the full-map path bit for bit.
//
// Labelling is label equivalence on a union-find forest in HBM (one int per pixel, parent index < child index, a root is the first
// pixel of its component in raster order): (1) init: every foreground pixel points at the start of its horizontal run inside its
// 64-pixel wave segment (a ballot, no memory traffic); (2) merge: atomicMin-based unions with the left segment and with the row
// above (N, else NW / NE), skipped where the previous pixel of the run already made the same link; (3) flatten; (4) roots take a
// compact id; (5) reduce: one atomic set per horizontal run, not per pixel (segmented wave scan).  Everything is integer work, so
// the table is reproducible except for its order (ids are handed out by an atomic counter; the host sorts by root).
#include <limits.h>

#include "common.h"

namespace ocrvi {

constexpr int kCompInts = 8;  // x0, y0, x1, y1, count, root, sum_lo, sum_hi  (sum = sum of round(prob * 2^20), unsigned 64-bit)

__device__ __forceinline__ int cc_load(const int* L, int i) { return __atomic_load_n(L + i, __ATOMIC_RELAXED); }
__device__ __forceinline__ int cc_find(const int* L, int a) {
    for (int p = cc_load(L, a); p != a; p = cc_load(L, a)) a = p;
    return a;
}
__device__ __forceinline__ void cc_unite(int* L, int a, int b) {
    for (;;) {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }  // a > b: hang the larger root under the smaller
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;  // somebody re-parented a meanwhile: carry on from there
    }
}

// grid (ceil(W / 256), H, pages), 256 threads: one pixel per thread, one 64-pixel row segment per wave
__global__ __launch_bounds__(256) void cc_init_kernel(const float* __restrict__ prob, int H, int W, float thresh, uint32_t* __restrict__ bits,
                                                      int* __restrict__ labels) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pg = blockIdx.z, lane = threadIdx.x & 63;
    const bool in = x < W;
    const size_t pix = ((size_t)pg * H + y) * W + x;
    const bool fg = in && prob[pix] > thresh;
    const unsigned long long b = __ballot(fg);
    if (in) {
        int lab = -1;
        if (fg) {  // start of this lane's run inside the segment: one past the nearest background lane below it
            const unsigned long long below = ~b & ((1ull << lane) - 1ull);
            const int start = below ? 64 - __builtin_clzll(below) : 0;
            lab = y * W + (x - lane) + start;
        }
        labels[pix] = lab;
        const int wpr = W >> 5;
        if (lane == 0) bits[((size_t)pg * H + y) * wpr + (x >> 5)] = (uint32_t)b;
        if (lane == 32) bits[((size_t)pg * H + y) * wpr + (x >> 5)] = (uint32_t)(b >> 32);
    }
}

__global__ __launch_bounds__(256) void cc_merge_kernel(int H, int W, int* __restrict__ labels) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pg = blockIdx.z, lane = threadIdx.x & 63;
    if (x >= W) return;
    int* L = labels + (size_t)pg * H * W;
    const int i = y * W + x;
    if (cc_load(L, i) < 0) return;
    const bool w = x > 0 && cc_load(L, i - 1) >= 0;
    if (w && lane == 0) cc_unite(L, i, i - 1);  // the run continues from the segment on the left
    if (y == 0) return;
    const bool n = cc_load(L, i - W) >= 0;
    const bool nw = x > 0 && cc_load(L, i - W - 1) >= 0;
    const bool ne = x + 1 < W && cc_load(L, i - W + 1) >= 0;
    if (n) {
        if (!(w && nw)) cc_unite(L, i, i - W);  // (w && nw: the pixel on the left made this link already, through its own N or NE)
    } else {
        if (nw && !w) cc_unite(L, i, i - W - 1);  // (w: the pixel on the left has nw as its N)
        if (ne) cc_unite(L, i, i - W + 1);
    }
}

__global__ __launch_bounds__(256) void cc_flatten_kernel(int H, int W, int* __restrict__ labels) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pg = blockIdx.z;
    if (x >= W) return;
    int* L = labels + (size_t)pg * H * W;
    const int i = y * W + x;
    if (L[i] >= 0) L[i] = cc_find(L, i);   // parents only ever move towards the root, so concurrent flattening is safe
}

// roots take a compact id; the root's own label becomes -2 - id (the others keep pointing at the root's index)
__global__ __launch_bounds__(256) void cc_roots_kernel(int H, int W, int* __restrict__ labels, int32_t* __restrict__ comps, int32_t* __restrict__ counts,
                                                       int cap) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pg = blockIdx.z;
    if (x >= W) return;
    int* L = labels + (size_t)pg * H * W;
    const int i = y * W + x;
    if (L[i] != i) return;
    const int id = atomicAdd(counts + pg, 1);
    if (id < cap) {
        int32_t* c = comps + ((size_t)pg * cap + id) * kCompInts;
        c[0] = INT_MAX; c[1] = INT_MAX; c[2] = -1; c[3] = -1; c[4] = 0; c[5] = i; c[6] = 0; c[7] = 0;
    }
    L[i] = -2 - id;   // ids >= cap are counted (the host sees the overflow) but own no table row
}

__global__ __launch_bounds__(256) void cc_reduce_kernel(const float* __restrict__ prob, int H, int W, const int* __restrict__ labels,
                                                        int32_t* __restrict__ comps, int cap) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, pg = blockIdx.z, lane = threadIdx.x & 63;
    const bool in = x < W;
    const int* L = labels + (size_t)pg * H * W;
    const int i = y * W + x;
    int id = -1;
    unsigned long long q = 0;
    if (in) {
        const int r = L[i];
        if (r != -1) {
            id = r < -1 ? -2 - r : -2 - L[r];
            q = (unsigned long long)__float2ll_rn(prob[(size_t)pg * H * W + i] * 1048576.0f);
        }
    }
    // inclusive wave prefix sum of q; a run's sum is a difference of two prefixes
    unsigned long long pre = q;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(pre, o);
        if (lane >= o) pre += t;
    }
    const int prev = __shfl_up(id, 1);
    const bool head = id >= 0 && (lane == 0 || prev != id);
    const unsigned long long ends = __ballot(id < 0 || head);   // a run ends before the next head or background lane
    const unsigned long long above = lane == 63 ? 0ull : (ends >> (lane + 1)) << (lane + 1);
    const int stop = above ? __builtin_ctzll(above) : 64;       // first lane past the run that starts here
    const unsigned long long hi = __shfl(pre, stop - 1), lo = __shfl(pre, lane ? lane - 1 : 0);   // (every lane takes part in the shuffles)
    if (!head || id >= cap) return;
    const unsigned long long run_sum = hi - (lane ? lo : 0ull);
    int32_t* c = comps + ((size_t)pg * cap + id) * kCompInts;
    atomicMin(c + 0, x);
    atomicMin(c + 1, y);
    atomicMax(c + 2, x + (stop - lane) - 1);
    atomicMax(c + 3, y);
    atomicAdd(c + 4, stop - lane);
    atomicAdd((unsigned long long*)(c + 6), run_sum);
}

// one block per page: offsets[id] = start (in floats) of component id's box in the packed buffer; offsets[min(count, cap)] = total
__global__ __launch_bounds__(256) void cc_offsets_kernel(const int32_t* __restrict__ comps, const int32_t* __restrict__ counts, int cap,
                                                         long long* __restrict__ offsets) {
    __shared__ long long part[257];
    const int pg = blockIdx.x, n = min(counts[pg], cap), t = threadIdx.x;
    const int per = (n + 255) / 256, lo = min(t * per, n), hi = min(lo + per, n);
    const int32_t* c = comps + (size_t)pg * cap * kCompInts;
    auto area = [&](int i) { return (long long)(c[i * kCompInts + 2] - c[i * kCompInts + 0] + 1) * (c[i * kCompInts + 3] - c[i * kCompInts + 1] + 1); };
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += area(i);
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int k = 0; k < 256; ++k) { const long long v = part[k]; part[k] = run; run += v; }
        part[256] = run;
    }
    __syncthreads();
    long long* o = offsets + (size_t)pg * (cap + 1);
    long long run = part[t];
    for (int i = lo; i < hi; ++i) { o[i] = run; run += area(i); }
    if (t == 0) o[n] = part[256];
}

// grid (cap blocks looped, pages): copy each component's box into the packed buffer (skipped for a page whose boxes exceed pack_cap)
__global__ __launch_bounds__(256) void cc_pack_kernel(const float* __restrict__ prob, int H, int W, const int32_t* __restrict__ comps,
                                                      const int32_t* __restrict__ counts, int cap, const long long* __restrict__ offsets,
                                                      float* __restrict__ packed, long long pack_cap) {
    const int pg = blockIdx.y, n = min(counts[pg], cap);
    const long long* o = offsets + (size_t)pg * (cap + 1);
    if (o[n] > pack_cap) return;
    const float* src = prob + (size_t)pg * H * W;
    float* dst = packed + (size_t)pg * pack_cap;
    for (int id = blockIdx.x; id < n; id += gridDim.x) {
        const int32_t* c = comps + ((size_t)pg * cap + id) * kCompInts;
        const int x0 = c[0], y0 = c[1], w = c[2] - c[0] + 1, h = c[3] - c[1] + 1;
        const long long base = o[id];
        for (int k = threadIdx.x; k < w * h; k += 256) {
            const int yy = k / w, xx = k - yy * w;
            dst[base + k] = src[(size_t)(y0 + yy) * W + x0 + xx];
        }
    }
}

}  // namespace ocrvi

extern "C" size_t ocrvi_db_components_workspace_bytes(int n_pages, int H, int W) { return (size_t)n_pages * H * W * sizeof(int32_t); }

extern "C" int ocrvi_db_components(int device, const float* prob, int n_pages, int H, int W, float thresh, uint32_t* mask_bits, int32_t* comps,
                                   int32_t* counts, int cap, long long* offsets, float* packed, long long pack_cap, void* workspace,
                                   void* stream) {
    using namespace ocrvi;
    OCRVI_CHECK(prob && mask_bits && comps && counts && workspace && n_pages > 0 && H > 0 && W > 0 && cap > 0, OCRVI_EINVAL,
                "db_components: bad argument");
    OCRVI_CHECK(W % 32 == 0 && (size_t)H * W < ((size_t)1 << 30), OCRVI_EINVAL, "db_components: W=%d must be a multiple of 32 (and H*W < 2^30)", W);
    OCRVI_CHECK((packed == nullptr) == (offsets == nullptr) && (!packed || pack_cap > 0), OCRVI_EINVAL, "db_components: packed and offsets go together");
    DeviceGuard dg(device);  // launches go to `device`; the caller's current device is restored on return
    OCRVI_HIP(dg.err);
    hipStream_t s = (hipStream_t)stream;
    int* labels = (int*)workspace;
    const dim3 grid((W + 255) / 256, H, n_pages), block(256);
    OCRVI_HIP(hipMemsetAsync(counts, 0, (size_t)n_pages * sizeof(int32_t), s));
    hipLaunchKernelGGL(cc_init_kernel, grid, block, 0, s, prob, H, W, thresh, mask_bits, labels);
    hipLaunchKernelGGL(cc_merge_kernel, grid, block, 0, s, H, W, labels);
    hipLaunchKernelGGL(cc_flatten_kernel, grid, block, 0, s, H, W, labels);
    hipLaunchKernelGGL(cc_roots_kernel, grid, block, 0, s, H, W, labels, comps, counts, cap);
    hipLaunchKernelGGL(cc_reduce_kernel, grid, block, 0, s, prob, H, W, (const int*)labels, comps, cap);
    if (packed) {
        hipLaunchKernelGGL(cc_offsets_kernel, dim3(n_pages), block, 0, s, (const int32_t*)comps, (const int32_t*)counts, cap, offsets);
        hipLaunchKernelGGL(cc_pack_kernel, dim3(std::min(cap, 256), n_pages), block, 0, s, prob, H, W, (const int32_t*)comps, (const int32_t*)counts,
                           cap, (const long long*)offsets, packed, pack_cap);
    }
    OCRVI_HIP(hipGetLastError());
    return OCRVI_OK;
}
