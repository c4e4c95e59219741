// Proposal creator: per-image score top-k (LDS radix select + LDS bitonic
// sort, fused box decode) and greedy NMS (wave-ballot bitmask kernel followed
// by a one-wave scan per image).  gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"

namespace frcnn {

// =========================================================================
// top-k
// =========================================================================
constexpr int TBLOCK = 1024;
constexpr int TOPK_MAX = 16384;

// One workgroup per image.  LDS: n_pad composites | 256 hist | 4 | 16 | 2 words.
__global__ void __launch_bounds__(TBLOCK)
topk_kernel(const float* __restrict__ scores, int N, int K, int n_pad,
            const float4* __restrict__ anchors, const float4* __restrict__ deltas,
            float* __restrict__ top_scores, int32_t* __restrict__ top_idx,
            float4* __restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* hist = reinterpret_cast<uint32_t*>(sel + n_pad);
    uint32_t* sh = hist + 256;
    uint32_t* wcnt = sh + 4;
    uint32_t* ctr = wcnt + TBLOCK / 64;
    const int b = blockIdx.x;
    const float* sc = scores + (size_t)b * N;
    auto key = [&](int i) -> uint32_t { return desc_key(sc[i]); };
    uint32_t T;
    int less;
    block_radix_select<TBLOCK>(key, N, K, hist, sh, T, less);
    block_collect_smallest<TBLOCK>(key, N, K, T, less, sel, wcnt, ctr);
    for (int t = threadIdx.x + K; t < n_pad; t += TBLOCK) sel[t] = ~0ull;
    __syncthreads();
    block_bitonic_sort<TBLOCK>(sel, n_pad);
    for (int r = threadIdx.x; r < K; r += TBLOCK) {
        const uint32_t idx = (uint32_t)sel[r];
        const size_t o = (size_t)b * K + r;
        top_scores[o] = sc[idx];
        top_idx[o] = (int32_t)idx;
        if (cand) cand[o] = decode_box(anchors[idx], deltas[(size_t)b * N + idx], false);
    }
}

// =========================================================================
// NMS
// =========================================================================
// IoU of the TF NMS kernel: corners re-ordered, area <= 0 on either side -> 0,
// inter = max(dy,0)*max(dx,0), iou = inter / (area_i + area_j - inter); a pair
// suppresses when iou > thr (strict).
struct NBox { float y0, x0, y1, x1, area; };

__device__ __forceinline__ NBox norm_box(float4 v) {
    NBox n;
    n.y0 = fminf(v.x, v.z); n.y1 = fmaxf(v.x, v.z);
    n.x0 = fminf(v.y, v.w); n.x1 = fmaxf(v.y, v.w);
    n.area = (n.y1 - n.y0) * (n.x1 - n.x0);
    return n;
}

// Exact evaluation of  (inter / u > thr)  that divides only inside a narrow
// band around the threshold: outside it the comparison of inter with thr*u is
// already decided with a margin far above the rounding of the product and of
// the quotient.
__device__ __forceinline__ bool iou_over(const NBox& a, const NBox& b, float thr, float thr_lo,
                                         float thr_hi) {
    const float iy0 = fmaxf(a.y0, b.y0), ix0 = fmaxf(a.x0, b.x0);
    const float iy1 = fminf(a.y1, b.y1), ix1 = fminf(a.x1, b.x1);
    const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
    const float u = a.area + b.area - inter;
    const bool areas = a.area > 0.0f && b.area > 0.0f;
    const float lo = u * thr_lo, hi = u * thr_hi;
    bool over;
    if (lo > 1e-30f && inter > hi) over = true;
    else if (lo > 1e-30f && inter < lo) over = false;
    else over = (inter / u) > thr;
    return areas && over;
}

// Bitmask kernel.  Workgroup = 4 waves; all share one block of 64 "row" boxes
// (higher score) staged in LDS, each wave owns one block of 64 "column" boxes
// (one per lane).  For each row the wave ballots the suppress predicate: the
// 64-bit result is that row's mask word for this column block.
// mask[b][i][cb] bit l  <=>  box i suppresses box cb*64+l  (only j > i set).
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float4* __restrict__ boxes, int K, int CB, float thr, float thr_lo,
                float thr_hi, unsigned long long* __restrict__ mask) {
    const int ib = blockIdx.y;
    const int b = blockIdx.z;
    const int jb0 = blockIdx.x * 4;
    if (jb0 + 3 < ib) return;                       // whole group left of the diagonal
    __shared__ NBox s_row[64];
    const float4* bx = boxes + (size_t)b * K;
    if (threadIdx.x < 64) {
        const int i = ib * 64 + threadIdx.x;
        s_row[threadIdx.x] = norm_box(i < K ? bx[i] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __syncthreads();
    const int jb = jb0 + wave_id();
    if (jb < ib || jb >= CB) return;                // wave-uniform
    const int lane = lane_id();
    const int j = jb * 64 + lane;
    const NBox cb = norm_box(j < K ? bx[j] : make_float4(0.f, 0.f, 0.f, 0.f));
    unsigned long long mine = 0ull;
#pragma unroll 4
    for (int r = 0; r < 64; ++r) {
        const int i = ib * 64 + r;
        const bool p = (j > i) && (j < K) && iou_over(s_row[r], cb, thr, thr_lo, thr_hi);
        const unsigned long long w = __ballot(p);
        if (lane == r) mine = w;
    }
    const int i = ib * 64 + lane;
    if (i < K) mask[((size_t)b * K + i) * CB + jb] = mine;
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// One wave per image walks the candidates in score order, 64 at a time.
// rem[] (LDS) = union of the mask rows of every box kept so far.
__global__ void __launch_bounds__(64)
nms_scan_kernel(const float4* __restrict__ boxes, const unsigned long long* __restrict__ mask,
                int K, int CB, int post_n, float4* __restrict__ proposed,
                int32_t* __restrict__ keep_pos, int32_t* __restrict__ keep_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* rem = reinterpret_cast<unsigned long long*>(smem);
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const float4* bx = boxes + (size_t)b * K;
    const unsigned long long* mk = mask + (size_t)b * K * CB;
    float4* prop = proposed + (size_t)b * post_n;
    int32_t* kp = keep_pos ? keep_pos + (size_t)b * post_n : nullptr;
    for (int w = lane; w < CB; w += 64) rem[w] = 0ull;
    __builtin_amdgcn_wave_barrier();
    int kept = 0;
#pragma unroll 1
    for (int ib = 0; ib < CB && kept < post_n; ++ib) {
        const int i0 = ib * 64;
        const int i = i0 + lane;
        const unsigned long long diag = i < K ? mk[(size_t)i * CB + ib] : 0ull;
        const int nvalid = K - i0 < 64 ? K - i0 : 64;
        const unsigned long long vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        unsigned long long alive = uniform64(~rem[ib] & vmask);
        unsigned long long keptmask = 0ull;
        int room = post_n - kept;
        while (alive != 0ull && room > 0) {          // wave-uniform
            const int r = __ffsll((long long)alive) - 1;
            keptmask |= 1ull << r;
            --room;
            const unsigned long long d =
                ((unsigned long long)__builtin_amdgcn_readlane((int)(diag >> 32), r) << 32) |
                (unsigned int)__builtin_amdgcn_readlane((int)diag, r);
            alive &= ~(d | (1ull << r));
        }
        // emit the kept boxes of this block
        if ((keptmask >> lane) & 1ull) {
            const int slot = kept + __popcll(keptmask & lanemask_lt());
            float4 v = bx[i];
            v.x = clip01(v.x); v.y = clip01(v.y); v.z = clip01(v.z); v.w = clip01(v.w);
            prop[slot] = v;
            if (kp) kp[slot] = i;
        }
        kept += __popcll(keptmask);
        if (kept >= post_n) break;
        // fold the kept rows into rem[] for the column blocks still ahead
        for (int w = ib + 1 + lane; w < CB; w += 64) {
            unsigned long long acc = 0ull;
            unsigned long long km = keptmask;
            while (km != 0ull) {                     // wave-uniform
                int r0 = __ffsll((long long)km) - 1; km &= km - 1ull;
                int r1 = -1, r2 = -1, r3 = -1;
                if (km) { r1 = __ffsll((long long)km) - 1; km &= km - 1ull; }
                if (km) { r2 = __ffsll((long long)km) - 1; km &= km - 1ull; }
                if (km) { r3 = __ffsll((long long)km) - 1; km &= km - 1ull; }
                const unsigned long long a0 = mk[(size_t)(i0 + r0) * CB + w];
                const unsigned long long a1 = r1 >= 0 ? mk[(size_t)(i0 + r1) * CB + w] : 0ull;
                const unsigned long long a2 = r2 >= 0 ? mk[(size_t)(i0 + r2) * CB + w] : 0ull;
                const unsigned long long a3 = r3 >= 0 ? mk[(size_t)(i0 + r3) * CB + w] : 0ull;
                acc |= a0 | a1 | a2 | a3;
            }
            rem[w] |= acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // zero padding (tf.image.combined_non_max_suppression pads boxes with 0)
    for (int s = kept + lane; s < post_n; s += 64) {
        prop[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kp) kp[s] = -1;
    }
    if (keep_count && lane == 0) keep_count[b] = kept;
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_topk(const float* scores, int B, int N, int K, const float* anchors,
                          const float* deltas, float* top_scores, int32_t* top_idx,
                          float* cand_boxes, frcnn_stream_t stream) {
    if (!scores || !top_scores || !top_idx || B <= 0 || N <= 0 || K <= 0 || K > N)
        return FRCNN_E_BADARG;
    const bool fuse = anchors && deltas;
    if ((anchors != nullptr) != (deltas != nullptr)) return FRCNN_E_BADARG;
    if (fuse && !cand_boxes) return FRCNN_E_BADARG;
    if (K > TOPK_MAX) return FRCNN_E_UNSUPPORTED;
    const int n_pad = next_pow2(K);
    const size_t lds = (size_t)n_pad * 8 + (256 + 4 + TBLOCK / 64 + 2) * 4;
    static_assert((size_t)TOPK_MAX * 8 + (256 + 4 + TBLOCK / 64 + 2) * 4 <= 160 * 1024, "LDS");
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return FRCNN_E_LAUNCH;
    FRCNN_LAUNCH(topk_kernel, dim3(B), dim3(TBLOCK), lds, (hipStream_t)stream, scores, N, K,
                       n_pad, reinterpret_cast<const float4*>(anchors),
                       reinterpret_cast<const float4*>(deltas), top_scores, top_idx,
                       fuse ? reinterpret_cast<float4*>(cand_boxes) : nullptr);
    return FRCNN_OK;
}

extern "C" size_t frcnn_nms_workspace_bytes(int B, int K) {
    if (B <= 0 || K <= 0) return 0;
    const size_t CB = ((size_t)K + 63) / 64;
    return (size_t)B * (size_t)K * CB * sizeof(unsigned long long);
}

extern "C" int frcnn_nms(const float* boxes, int B, int K, float iou_threshold, int post_n,
                         float* proposed, int32_t* keep_pos, int32_t* keep_count, void* workspace,
                         size_t workspace_bytes, frcnn_stream_t stream) {
    if (!boxes || !proposed || B <= 0 || K <= 0 || post_n <= 0) return FRCNN_E_BADARG;
    if (!(iou_threshold >= 0.0f)) return FRCNN_E_BADARG;
    if (!workspace || workspace_bytes < frcnn_nms_workspace_bytes(B, K)) return FRCNN_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7u) != 0) return FRCNN_E_BADARG;
    const int CB = (K + 63) / 64;
    if (CB > 65535 || B > 65535) return FRCNN_E_UNSUPPORTED;
    if ((size_t)CB * 8 > 150 * 1024) return FRCNN_E_UNSUPPORTED;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(workspace);
    // band edges for the division-free fast path of iou_over()
    const float thr_lo = iou_threshold * 0.999998f;
    const float thr_hi = iou_threshold * 1.000002f;
    FRCNN_LAUNCH(nms_mask_kernel, dim3((CB + 3) / 4, CB, B), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const float4*>(boxes), K, CB,
                       iou_threshold, thr_lo, thr_hi, mask);
    const size_t lds = (size_t)CB * 8;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return FRCNN_E_LAUNCH;
    FRCNN_LAUNCH(nms_scan_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(boxes), mask, K, CB, post_n,
                       reinterpret_cast<float4*>(proposed), keep_pos, keep_count);
    return FRCNN_OK;
}
