// Shared device helpers for the gfx950 kernels: wave64 scans, LDS radix
// select, ordered compaction, LDS bitonic sort, Philox counter keys.
// Built with -ffp-contract=off: every float op below rounds once, in the
// order written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/frcnn_hip.h"

#define FRCNN_WAVE 64

// Workgroup size of the training-target kernels (anchor assignment, both samplers, proposal
// labels).  One wave: such workgroups take wave slots as they free up while a kernel of
// another stream fills the chip; larger ones wait until that kernel's grid has drained.
#ifndef FRCNN_TARGET_BLOCK
#define FRCNN_TARGET_BLOCK 64
#endif

// hipGetLastError() also reports errors left behind by earlier, unrelated
// runtime calls of this thread: clear it before a launch, read it after.
#define FRCNN_LAUNCH(kernel, grid, block, lds, stream, ...)                         \
    do {                                                                            \
        (void)hipGetLastError();                                                    \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);          \
        if (hipGetLastError() != hipSuccess) return FRCNN_E_LAUNCH;                 \
    } while (0)

namespace frcnn {

// Host-side tuning values (kernel selection thresholds).  Each is read from its environment
// variable ONCE, when the library is loaded, and can be replaced afterwards with
// frcnn_set_tuning(); entry points read the stored value (an atomic), never the environment.
enum Tuning {
    TUNE_NMS_GK = 0,          // FRCNN_NMS_GK: mask workgroups per column block (0 = by batch size)
    TUNE_ROI_BAND_MIN_WG,     // FRCNN_ROI_BAND_MIN_WG: band kernel from n workgroups on (< 0: never)
    TUNE_BWD_CX,              // FRCNN_BWD_CX: gather backward: feature cells per wave (1 | 2); patch-owner
                              // backward: patch shape 10 * rows + columns (44 | 34 | 46 | 26)
    TUNE_BWD_CHUNK,           // FRCNN_BWD_CHUNK: RoIs per backward launch (0 = by batch size)
    TUNE_BWD_BAND,            // FRCNN_BWD_BAND: patch-owner backward (0: off, the gather kernel runs; else on)
    TUNE_NMS_FUSED_MAX_B,     // FRCNN_NMS_FUSED_MAX_B: one-launch NMS up to this many images
    TUNE_ROI_BAND_SPLIT,      // FRCNN_ROI_BAND_SPLIT: RoI ranges per band workgroup set (0 = by size)
    TUNE_TOPK_MERGE_MAX_RUNS, // FRCNN_TOPK_MERGE_MAX_RUNS: sorted-runs top-k up to this many 1,024-score blocks
                              // in the call (0: never, the radix sort runs)
    TUNE_COUNT
};
constexpr long long TUNE_UNSET = (long long)0x8000000000000000ull;
long long tuning(Tuning t);                       // TUNE_UNSET when neither set nor in the environment

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// ---------------------------------------------------------------- Philox ----
__device__ __forceinline__ uint32_t philox4x32_10_first(uint32_t c0, uint32_t c1, uint32_t c2,
                                                        uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0;
        const uint64_t p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// key of list position i in shuffle stream `stream` of image `image`
__device__ __forceinline__ uint32_t counter_key(uint32_t i, uint32_t stream, uint32_t image,
                                                uint64_t seed) {
    return philox4x32_10_first(i, stream, image, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// ------------------------------------------------- float <-> sortable key ----
// ascending key order == descending float order; -0 and +0 map to one key.
__device__ __forceinline__ uint32_t desc_key(float v) {
    if (v == 0.0f) v = 0.0f;
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending order key
    return ~u;
}
__device__ __forceinline__ float desc_key_inv(uint32_t k) {
    uint32_t u = ~k;
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}

// ------------------------------------------------------------ histogram add ----
// Wave-aggregated LDS histogram increment: lanes with equal digits share one
// atomic (sigmoid scores put most keys in two or three top-byte bins).
__device__ __forceinline__ void hist_add(uint32_t* hist, bool active, uint32_t digit) {
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const unsigned long long act = __ballot(active);
        if (act == 0ull) return;
        const int leader = __ffsll((long long)act) - 1;
        const uint32_t d0 = __shfl(digit, leader);
        const unsigned long long peers = __ballot(active && digit == d0);
        if (lane_id() == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(peers));
        if (digit == d0) active = false;
    }
    if (active) atomicAdd(&hist[digit], 1u);
}

// ------------------------------------------------------------- radix select ----
// m-th smallest (1-based) of key(0..M-1).  Returns T and count_less = #{key<T}.
// hist: 256 words of LDS; sh: 4 words of LDS.  All threads call; 1 <= m <= M.
template <int BLOCK, typename KeyFn>
__device__ void block_radix_select(KeyFn key, int M, int m, uint32_t* hist, uint32_t* sh,
                                   uint32_t& T, int& count_less) {
    const int tid = threadIdx.x;
    uint32_t prefix = 0, mask = 0;
    int remaining = m, less = 0;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int t = tid; t < 256; t += BLOCK) hist[t] = 0;
        __syncthreads();
        const int iters = (M + BLOCK - 1) / BLOCK;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            const int i = it * BLOCK + tid;
            uint32_t k = 0;
            bool act = false;
            if (i < M) { k = key(i); act = ((k & mask) == prefix); }
            hist_add(hist, act, (k >> shift) & 255u);
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t h0 = hist[tid * 4 + 0], h1 = hist[tid * 4 + 1];
            const uint32_t h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
            const uint32_t s = h0 + h1 + h2 + h3;
            uint32_t incl = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (tid >= d) incl += o;
            }
            const uint32_t excl = incl - s;
            if (excl < (uint32_t)remaining && incl >= (uint32_t)remaining) {
                uint32_t below = excl, digit;
                if (below + h0 >= (uint32_t)remaining) digit = 0;
                else if ((below += h0) + h1 >= (uint32_t)remaining) digit = 1;
                else if ((below += h1) + h2 >= (uint32_t)remaining) digit = 2;
                else { below += h2; digit = 3; }
                sh[0] = tid * 4 + digit;
                sh[1] = below;
            }
        }
        __syncthreads();
        const uint32_t d = sh[0], below = sh[1];
        less += (int)below;
        remaining -= (int)below;
        prefix |= d << shift;
        mask |= 0xFFu << shift;
        __syncthreads();
    }
    T = prefix;
    count_less = less;
}

// ------------------------------------------------------------------ collect ----
// Writes the m smallest (key,i) pairs as composites (key<<32 | i) into
// dst[0..m): every key < T (unordered), then the first m-count_less items
// with key == T in ascending i.  wcnt: BLOCK/64 words of LDS, ctr: 2 words.
template <int BLOCK, typename KeyFn>
__device__ void block_collect_smallest(KeyFn key, int M, int m, uint32_t T, int count_less,
                                       unsigned long long* dst, uint32_t* wcnt, uint32_t* ctr) {
    const int tid = threadIdx.x;
    const int lane = lane_id(), wave = wave_id();
    constexpr int NW = BLOCK / 64;
    if (tid == 0) { ctr[0] = 0; ctr[1] = 0; }
    __syncthreads();
    const int need_ties = m - count_less;
    const int iters = (M + BLOCK - 1) / BLOCK;
    int tie_base = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int i = it * BLOCK + tid;
        uint32_t k = 0xFFFFFFFFu;
        bool lt = false, eq = false;
        if (i < M) { k = key(i); lt = k < T; eq = (k == T); }
        // unordered append of the strictly-smaller items
        const unsigned long long blt = __ballot(lt);
        if (blt) {
            uint32_t base = 0;
            const int leader = __ffsll((long long)blt) - 1;
            if (lane == leader) base = atomicAdd(&ctr[0], (uint32_t)__popcll(blt));
            base = __shfl(base, leader);
            if (lt) dst[base + __popcll(blt & lanemask_lt())] =
                ((unsigned long long)k << 32) | (uint32_t)i;
        }
        // ordered append of ties while fewer than need_ties have been taken
        if (tie_base < need_ties) {             // block-uniform
            const unsigned long long beq = __ballot(eq);
            if (lane == 0) wcnt[wave] = (uint32_t)__popcll(beq);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int c = (int)wcnt[w];
                if (w < wave) before += c;
                total += c;
            }
            const int rank = tie_base + before + __popcll(beq & lanemask_lt());
            if (eq && rank < need_ties)
                dst[count_less + rank] = ((unsigned long long)k << 32) | (uint32_t)i;
            tie_base += total;
            __syncthreads();
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------- bitonic sort ----
// Ascending sort of s[0..n) (n a power of two) in LDS; all threads call.
template <int BLOCK>
__device__ void block_bitonic_sort(unsigned long long* s, int n) {
    const int tid = threadIdx.x;
    const int half = n >> 1;
#pragma unroll 1
    for (int k = 2; k <= n; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll 1
            for (int t = tid; t < half; t += BLOCK) {
                const int idx = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int prt = idx | j;
                const bool up = ((idx & k) == 0);
                const unsigned long long a = s[idx], b = s[prt];
                if ((a > b) == up) { s[idx] = b; s[prt] = a; }
            }
            __syncthreads();
        }
    }
}

__host__ __device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ------------------------------------------------------- one-instruction min/max ----
// fminf()/fmaxf() make hipcc canonicalise each operand first (an extra v_max per
// operand); these are the bare instructions.  NaN operands are not expected here.
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin2(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ------------------------------------------------------------ box arithmetic ----
// decode of one box, op order of module/data.py:163-175 (clip_wh=false) and
// :185-196 (clip_wh=true)
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

__device__ __forceinline__ float4 decode_box(float4 anc, float4 t, bool clip_wh) {
    float w_a = anc.w - anc.y;
    float h_a = anc.z - anc.x;
    if (clip_wh) {
        w_a = fminf(fmaxf(w_a, 1e-2f), 1.0f);
        h_a = fminf(fmaxf(h_a, 1e-2f), 1.0f);
    }
    float x = t.x * w_a * 0.1f + anc.y;
    float y = t.y * h_a * 0.1f + anc.x;
    const float w = expf(t.z * 0.2f) * w_a;
    const float h = expf(t.w * 0.2f) * h_a;
    float x_max = x + w;
    float y_max = y + h;
    float4 o;
    o.x = clip01(y); o.y = clip01(x); o.z = clip01(y_max); o.w = clip01(x_max);
    return o;
}

// target encode, op order of module/data.py:143-148 / :252-257
__device__ __forceinline__ float4 encode_box(float4 anc, float4 g, bool clip_wh) {
    float w_a = anc.w - anc.y;
    float h_a = anc.z - anc.x;
    if (clip_wh) {
        w_a = fminf(fmaxf(w_a, 1e-2f), 1.0f);
        h_a = fminf(fmaxf(h_a, 1e-2f), 1.0f);
    }
    float4 o;
    o.x = (g.y - anc.y) / w_a / 0.1f;
    o.y = (g.x - anc.x) / h_a / 0.1f;
    o.z = logf((g.w - g.y) / w_a) / 0.2f;
    o.w = logf((g.z - g.x) / h_a) / 0.2f;
    return o;
}

// ------------------------------------------------------ crop sample geometry ----
// tf.image.crop_and_resize's source coordinate of one sample (module/model.py:54).
// geometry of sample s along one axis; c1,c2 box corners, dim image size,
// n crop size.
__device__ __forceinline__ void axis_sample(float c1, float c2, int dim, int n, int s, int& lo,
                                            int& hi, float& frac, int& ok) {
    float in;
    if (n > 1) {
        const float scale = ((c2 - c1) * (float)(dim - 1)) / (float)(n - 1);
        in = c1 * (float)(dim - 1) + (float)s * scale;
    } else {
        in = (float)(0.5 * (double)(c1 + c2) * (double)(dim - 1));
    }
    ok = (in >= 0.0f && in <= (float)(dim - 1));      // NaN -> extrapolated
    const float f = floorf(in), c = ceilf(in);
    frac = in - f;
    lo = ok ? (int)f : 0;
    hi = ok ? (int)c : 0;
}

// ---- wave-uniform reads of one lane
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rlf(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


// The source coordinate of sample s is base + s * scale (axis_sample's expression): base and
// scale depend on the box only, so the lane that tests a RoI as a candidate computes them (one
// division) and the lanes that evaluate its samples later take them by readlane.
__device__ __forceinline__ void axis_line(float c1, float c2, int dim, int n, float& base,
                                          float& scale) {
    if (n > 1) {
        scale = ((c2 - c1) * (float)(dim - 1)) / (float)(n - 1);
        base = c1 * (float)(dim - 1);
    } else {
        scale = 0.0f;
        base = (float)(0.5 * (double)(c1 + c2) * (double)(dim - 1));
    }
}
__device__ __forceinline__ bool line_reaches(float base, float scale, int n, int t0, int t1) {
    const float last = n > 1 ? base + (float)(n - 1) * scale : base;
    const float lo = floorf(fminf(base, last)), hi = ceilf(fmaxf(base, last));
    return (float)t1 >= lo && (float)t0 <= hi;              // NaN -> false
}
__device__ __forceinline__ void line_sample(float base, float scale, int dim, int n, int s, int& lo,
                                            int& hi, float& frac, int& ok) {
    const float in = n > 1 ? base + (float)s * scale : base;
    ok = (in >= 0.0f && in <= (float)(dim - 1));      // NaN -> extrapolated
    const float f = floorf(in), c = ceilf(in);
    frac = in - f;
    lo = ok ? (int)f : 0;
    hi = ok ? (int)c : 0;
}

// RoI backward, patch-owner form (frcnn_roi_bwd.hip); FRCNN_E_UNSUPPORTED when the shape is not its
int launch_roi_bwd_patch(const float* dY, const uint8_t* winner, const float* rois, int B, int R, int H,
                         int W, int C, int crop_h, int crop_w, float* dfmap, hipStream_t stream);

}  // namespace frcnn
