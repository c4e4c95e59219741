// Detection post-process: class argmax, per-class box decode, background drop,
// score-ordered greedy NMS with a small output budget, score cut.
// One 1024-thread workgroup per image.  gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"

namespace frcnn {

constexpr int DBLOCK = 1024;
constexpr int DMAX_R = 16384;
constexpr int DMAX_OUT = 1024;

// IoU of the TF NMS kernels (corners re-ordered; a non-positive area gives 0)
__device__ __forceinline__ float tf_nms_iou(float4 a, float4 b) {
    const float ay0 = fminf(a.x, a.z), ay1 = fmaxf(a.x, a.z), ax0 = fminf(a.y, a.w), ax1 = fmaxf(a.y, a.w);
    const float by0 = fminf(b.x, b.z), by1 = fmaxf(b.x, b.z), bx0 = fminf(b.y, b.w), bx1 = fmaxf(b.y, b.w);
    const float area_a = (ay1 - ay0) * (ax1 - ax0), area_b = (by1 - by0) * (bx1 - bx0);
    if (area_a <= 0.0f || area_b <= 0.0f) return 0.0f;
    const float iy0 = fmaxf(ay0, by0), ix0 = fmaxf(ax0, bx0);
    const float iy1 = fminf(ay1, by1), ix1 = fminf(ax1, bx1);
    const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
    return inter / (area_a + area_b - inter);
}

// LDS: n_pad composites | max_out kept proposal ids | max_out kept scores | 16 words
__global__ void __launch_bounds__(DBLOCK)
detect_post_kernel(const float4* __restrict__ proposed, const float* __restrict__ cls_prob,
                   const float4* __restrict__ reg, int R, int NC, int bg_label, int max_out,
                   float iou_thr, float score_thr, int n_pad, float4* __restrict__ decoded,
                   int32_t* __restrict__ cls_arg, float4* __restrict__ boxes,
                   float* __restrict__ scores, int32_t* __restrict__ classes,
                   int32_t* __restrict__ src, int32_t* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem);
    int32_t* k_src = reinterpret_cast<int32_t*>(sel + n_pad);
    float* k_score = reinterpret_cast<float*>(k_src + max_out);
    uint32_t* wmin = reinterpret_cast<uint32_t*>(k_score + max_out);     // DBLOCK/64 words
    const int b = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const size_t base = (size_t)b * R;
    // ---- argmax class, decode with that class's offsets (model_train.py:747-750)
    for (int r = tid; r < n_pad; r += DBLOCK) {
        unsigned long long comp = ~0ull;
        if (r < R) {
            const float* p = cls_prob + (base + r) * NC;
            float best = p[0];
            int arg = 0;
            for (int c = 1; c < NC; ++c) {
                const float v = p[c];
                if (v > best) { best = v; arg = c; }
            }
            const float4 box = decode_box(proposed[base + r], reg[(base + r) * NC + arg], true);
            decoded[base + r] = box;
            cls_arg[base + r] = arg;
            if (arg != bg_label) comp = ((unsigned long long)desc_key(best) << 32) | (uint32_t)r;
        }
        sel[r] = comp;
    }
    __syncthreads();
    block_bitonic_sort<DBLOCK>(sel, n_pad);
    // ---- greedy NMS over the score order; thread t owns sorted positions t, t+1024, ...
    uint32_t alive = 0;                                   // bit i <-> position tid + i*DBLOCK
    for (int i = 0; i * DBLOCK + tid < n_pad; ++i)
        if (sel[i * DBLOCK + tid] != ~0ull) alive |= 1u << i;
    int nk = 0;
#pragma unroll 1
    while (nk < max_out) {
        // first alive position in the block
        uint32_t mine = 0xFFFFFFFFu;
        if (alive) mine = (uint32_t)((__ffs((int)alive) - 1) * DBLOCK + tid);
        for (int d = 32; d > 0; d >>= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, d));
        if (lane == 0) wmin[wave] = mine;
        __syncthreads();
        uint32_t head = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < DBLOCK / 64; ++w) head = min(head, wmin[w]);
        if (head == 0xFFFFFFFFu) break;                   // block-uniform
        const unsigned long long hc = sel[head];
        const uint32_t hr = (uint32_t)hc;
        const float4 hb = decoded[base + hr];
        if (tid == 0) {
            k_src[nk] = (int32_t)hr;
            k_score[nk] = cls_prob[(base + hr) * NC + cls_arg[base + hr]];
        }
        ++nk;
        // suppress: every alive candidate after `head` whose IoU with it exceeds the threshold
        uint32_t a = alive;
        while (a) {
            const int i = __ffs((int)a) - 1;
            a &= a - 1u;
            const uint32_t pos = (uint32_t)(i * DBLOCK + tid);
            if (pos == head) { alive &= ~(1u << i); continue; }
            const uint32_t r = (uint32_t)sel[pos];
            if (tf_nms_iou(decoded[base + r], hb) > iou_thr) alive &= ~(1u << i);
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- score cut (scores descend, so the survivors are a prefix) and padded outputs
    int n = 0;
    for (int k = 0; k < nk; ++k) n += (k_score[k] >= score_thr) ? 1 : 0;
    for (int k = tid; k < max_out; k += DBLOCK) {
        const size_t o = (size_t)b * max_out + k;
        const bool ok = k < nk && k_score[k] >= score_thr;
        // positions of survivors: rank among the selected with score >= cut
        int slot = 0;
        for (int j = 0; j < k && j < nk; ++j) slot += (k_score[j] >= score_thr) ? 1 : 0;
        if (ok) {
            const int32_t r = k_src[k];
            const size_t oo = (size_t)b * max_out + slot;
            boxes[oo] = decoded[base + r];
            scores[oo] = k_score[k];
            classes[oo] = cls_arg[base + r];
            src[oo] = r;
        }
        if (k >= n) {
            boxes[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            scores[o] = 0.f;
            classes[o] = -1;
            src[o] = -1;
        }
    }
    if (tid == 0) count[b] = n;
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_detect_postprocess(const float* proposed, const float* cls_prob,
                                        const float* reg, int B, int R, int num_classes,
                                        int bg_label, int max_out, float iou_threshold,
                                        float score_threshold, float* decoded, int32_t* cls_arg,
                                        float* boxes, float* scores, int32_t* classes, int32_t* src,
                                        int32_t* count, frcnn_stream_t stream) {
    if (!proposed || !cls_prob || !reg || !decoded || !cls_arg || !boxes || !scores || !classes ||
        !src || !count || B <= 0 || R <= 0 || num_classes <= 0 || max_out <= 0)
        return FRCNN_E_BADARG;
    if (!(iou_threshold >= 0.0f)) return FRCNN_E_BADARG;
    if (R > DMAX_R || max_out > DMAX_OUT || B > 65535) return FRCNN_E_UNSUPPORTED;
    const int n_pad = next_pow2(R);
    const size_t lds = (size_t)n_pad * 8 + (size_t)max_out * 8 + (DBLOCK / 64 + 2) * 4;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(detect_post_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return FRCNN_E_LAUNCH;
    FRCNN_LAUNCH(detect_post_kernel, dim3(B), dim3(DBLOCK), lds, (hipStream_t)stream,
                 reinterpret_cast<const float4*>(proposed), cls_prob,
                 reinterpret_cast<const float4*>(reg), R, num_classes, bg_label, max_out,
                 iou_threshold, score_threshold, n_pad, reinterpret_cast<float4*>(decoded), cls_arg,
                 reinterpret_cast<float4*>(boxes), scores, classes, src, count);
    return FRCNN_OK;
}
