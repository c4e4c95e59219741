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
// inter = max(dy,0)*max(dx,0), iou = inter / (area_i + area_j - inter); box i
// (higher score) suppresses box j when iou > thr (strict).
//
// Layout.  Candidates are cut into blocks of 64 in score order.  maskT[b][ib][j]
// (u64) holds, for column box j, one bit per row box of row block ib: bit r set
// <=> box ib*64+r suppresses box j (only rows before j can be set).  The greedy
// scan of column block jb needs the words of row blocks 0..jb only, and those
// loads do not depend on the scan state, so they stream.
//
// Work is issued in chunks of column blocks: mask kernel for the chunk (all row
// blocks up to the column block), then scan kernel for the chunk.  An image
// that has collected post_n boxes raises its `done` flag and every later block
// of that image returns at once, so the pair tests actually executed are about
// (last candidate examined)^2 / 2 instead of K^2 / 2.
struct NBox { float y0, x0, y1, x1, area, t; };          // t = thr_lo * area

__device__ __forceinline__ NBox norm_box(float4 v, float thr_lo) {
    NBox n;
    n.y0 = fminf(v.x, v.z); n.y1 = fmaxf(v.x, v.z);
    n.x0 = fminf(v.y, v.w); n.x1 = fmaxf(v.y, v.w);
    n.area = (n.y1 - n.y0) * (n.x1 - n.x0);
    n.t = thr_lo * n.area;
    return n;
}

constexpr float NMS_TINY_AREA = 1e-18f;

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// One wave = one 64x64 tile: lane = row box (registers), the 64 column boxes sit
// in LDS.  Per column: a 3-instruction conservative filter
//     inter*(1+thr) > thr*(1-4e-6)*(area_i+area_j)
// (implied by iou > thr with a margin far above float rounding), and only when
// some lane passes it the exact division; the wave ballot of the predicate is
// the column's mask word.
template <bool DIAG>
__device__ __forceinline__ unsigned long long mask_tile(const NBox& a, float c1, const NBox* cols,
                                                        float thr, int lane) {
    unsigned long long mine = 0ull;
#pragma unroll 8
    for (int c = 0; c < 64; ++c) {
        const NBox cb = cols[c];
        const float iy0 = vmax2(a.y0, cb.y0), ix0 = vmax2(a.x0, cb.x0);
        const float iy1 = vmin2(a.y1, cb.y1), ix1 = vmin2(a.x1, cb.x1);
        const float inter = vmax2(iy1 - iy0, 0.0f) * vmax2(ix1 - ix0, 0.0f);
        bool p = inter * c1 > a.t + cb.t;
        if (DIAG) p = p && (c > lane);
        if (__any(p)) {
            const float u = a.area + cb.area - inter;
            p = p && (a.area > 0.0f) && (cb.area > 0.0f) && (inter / u > thr);
        }
        const unsigned long long w = __ballot(p);
        if (lane == c) mine = w;
    }
    return mine;
}

__global__ void __launch_bounds__(256)
nms_mask_kernel(const float4* __restrict__ boxes, int K, int CB, int jb_begin, float thr, float c1,
                float thr_lo, const int* __restrict__ done, unsigned long long* __restrict__ maskT) {
    const int b = blockIdx.z;
    const int jb = jb_begin + blockIdx.y;
    const int ib0 = blockIdx.x * 4;
    if (ib0 > jb || done[b]) return;                   // block-uniform
    __shared__ NBox s_col[64];
    const float4* bx = boxes + (size_t)b * K;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < 64) {
        const int j = jb * 64 + threadIdx.x;
        s_col[threadIdx.x] = norm_box(j < K ? bx[j] : zero4, thr_lo);
    }
    __syncthreads();
    const int ib = ib0 + wave_id();
    if (ib > jb) return;                               // wave-uniform
    const int lane = lane_id();
    const int i = ib * 64 + lane;
    const NBox a = norm_box(i < K ? bx[i] : zero4, thr_lo);
    // a tiny row area could underflow the filter's products: +inf makes the filter pass
    // whenever the boxes intersect, and the exact test decides
    const float c1l = a.area < NMS_TINY_AREA ? __builtin_inff() : c1;
    const unsigned long long mine = (ib == jb) ? mask_tile<true>(a, c1l, s_col, thr, lane)
                                               : mask_tile<false>(a, c1l, s_col, thr, lane);
    maskT[((size_t)b * CB + ib) * ((size_t)CB * 64) + (size_t)jb * 64 + lane] = mine;
}

// One workgroup (4 waves) per image walks the chunk's column blocks in order.
// kept[] (LDS) = keep bits of every block so far.  For block jb the waves split
// the row blocks 0..jb-1: a column is dead if any (word & kept[ib]) is non-zero;
// wave 0 then resolves the block's own 64x64 triangle serially and emits.
__global__ void __launch_bounds__(256)
nms_scan_kernel(const float4* __restrict__ boxes, const unsigned long long* __restrict__ maskT,
                int K, int CB, int jb_begin, int jb_end, int post_n,
                unsigned long long* __restrict__ kept_g, int* __restrict__ count_g,
                int* __restrict__ done, float4* __restrict__ proposed,
                int32_t* __restrict__ keep_pos, int32_t* __restrict__ keep_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* kept = reinterpret_cast<unsigned long long*>(smem);     // CB words
    unsigned long long* deadw = kept + CB;                                      // 4 words
    int* s_cnt = reinterpret_cast<int*>(deadw + 4);
    const int b = blockIdx.x;
    if (done[b]) return;                               // block-uniform
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const float4* bx = boxes + (size_t)b * K;
    const size_t KP = (size_t)CB * 64;
    const unsigned long long* mk = maskT + (size_t)b * CB * KP;
    unsigned long long* kg = kept_g + (size_t)b * CB;
    float4* prop = proposed + (size_t)b * post_n;
    int32_t* kp = keep_pos ? keep_pos + (size_t)b * post_n : nullptr;
    for (int w = tid; w < jb_begin; w += 256) kept[w] = kg[w];
    if (tid == 0) *s_cnt = count_g[b];
    __syncthreads();
    int jb = jb_begin;
#pragma unroll 1
    for (; jb < jb_end; ++jb) {
        const int count = *s_cnt;
        if (count >= post_n) break;                    // block-uniform
        const int j0 = jb * 64;
        const unsigned long long* col = mk + j0 + lane;
        const unsigned long long diag = wave == 0 ? col[(size_t)jb * KP] : 0ull;
        unsigned long long hit = 0ull;
#pragma unroll 8
        for (int ib = wave; ib < jb; ib += 4) hit |= col[(size_t)ib * KP] & kept[ib];
        const unsigned long long dead = __ballot(hit != 0ull);
        if (lane == 0) deadw[wave] = dead;
        __syncthreads();
        if (wave == 0) {
            const int nvalid = K - j0 < 64 ? K - j0 : 64;
            const unsigned long long vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
            unsigned long long alive = uniform64(~(deadw[0] | deadw[1] | deadw[2] | deadw[3]) & vmask);
            unsigned long long keptmask = 0ull;
            int room = post_n - count;
            while (alive != 0ull && room > 0) {        // wave-uniform
                const int r = __ffsll((long long)alive) - 1;
                keptmask |= 1ull << r;
                --room;
                const unsigned long long sup = __ballot(((diag >> r) & 1ull) != 0ull);
                alive &= ~(sup | (1ull << r));
            }
            if ((keptmask >> lane) & 1ull) {
                const int slot = count + __popcll(keptmask & lanemask_lt());
                float4 v = bx[j0 + lane];
                v.x = clip01(v.x); v.y = clip01(v.y); v.z = clip01(v.z); v.w = clip01(v.w);
                prop[slot] = v;
                if (kp) kp[slot] = j0 + lane;
            }
            if (lane == 0) {
                kept[jb] = keptmask;
                *s_cnt = count + __popcll(keptmask);
            }
        }
        __syncthreads();
    }
    // persist the state for the next chunk; finish the image if it is complete
    for (int w = jb_begin + tid; w < jb; w += 256) kg[w] = kept[w];
    const int count = *s_cnt;
    if (count >= post_n || jb_end >= CB) {
        // zero padding (tf.image.combined_non_max_suppression pads boxes with 0)
        for (int s = count + tid; s < post_n; s += 256) {
            prop[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kp) kp[s] = -1;
        }
        if (tid == 0) {
            done[b] = 1;
            if (keep_count) keep_count[b] = count;
        }
    }
    if (tid == 0) count_g[b] = count;
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

// workspace = maskT [B][CB][CB*64] u64 | kept [B][CB] u64 | count [B] i32 | done [B] i32
static size_t nms_mask_bytes(int B, int K) {
    const size_t CB = ((size_t)K + 63) / 64;
    return (size_t)B * CB * CB * 64 * sizeof(unsigned long long);
}

extern "C" size_t frcnn_nms_workspace_bytes(int B, int K) {
    if (B <= 0 || K <= 0) return 0;
    const size_t CB = ((size_t)K + 63) / 64;
    return nms_mask_bytes(B, K) + (size_t)B * CB * 8 + (size_t)B * 8;
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
    const size_t lds = (size_t)CB * 8 + 4 * 8 + 16;
    if (lds > 150 * 1024) return FRCNN_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    unsigned long long* maskT = reinterpret_cast<unsigned long long*>(ws);
    unsigned long long* kept = reinterpret_cast<unsigned long long*>(ws + nms_mask_bytes(B, K));
    int* count = reinterpret_cast<int*>(ws + nms_mask_bytes(B, K) + (size_t)B * CB * 8);
    int* done = count + B;
    if (hipMemsetAsync(count, 0, (size_t)B * 8, st) != hipSuccess) return FRCNN_E_LAUNCH;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return FRCNN_E_LAUNCH;
    // filter constants: iou > thr  =>  inter*(1+thr) > thr*(1-4e-6)*(area_i+area_j)
    const float c1 = 1.0f + iou_threshold;
    const float thr_lo = iou_threshold * 0.999996f;
    // chunks of column blocks: 8 x 16, then 32,32,64,64,... (fine early: most images stop early)
    int jb = 0, width = 16, n = 0;
    while (jb < CB) {
        const int je = jb + width < CB ? jb + width : CB;
        FRCNN_LAUNCH(nms_mask_kernel, dim3((je + 3) / 4, je - jb, B), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(boxes), K, CB, jb, iou_threshold, c1, thr_lo,
                     done, maskT);
        FRCNN_LAUNCH(nms_scan_kernel, dim3(B), dim3(256), lds, st,
                     reinterpret_cast<const float4*>(boxes), maskT, K, CB, jb, je, post_n, kept,
                     count, done, reinterpret_cast<float4*>(proposed), keep_pos, keep_count);
        jb = je;
        if (++n >= 8 && n % 2 == 0 && width < 4096) width *= 2;
    }
    return FRCNN_OK;
}
