// Anchor grid, anchors x GT IoU + assignment flags, box decode / encode.
// gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"
#include <math.h>

namespace frcnn {

// ----------------------------------------------------------------- anchors ----
struct AnchorHalf {
    double hw[64];
    double hh[64];
};

// One thread per (y, x, a).  Half extents come from the host (IEEE sqrt);
// the kernel does the float64 centre, the add/sub, the clip and the cast
// (module/anchor.py:15-16, 30-33, 43).
__global__ void __launch_bounds__(256)
anchors_kernel(float4* __restrict__ out, int H, int W, int A, AnchorHalf half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = H * W * A;
    if (idx >= total) return;
    const int a = idx % A;
    const int cell = idx / A;
    const int x = cell % W, y = cell / W;
    const double cx = ((double)x + 0.5) / (double)W;
    const double cy = ((double)y + 0.5) / (double)H;
    const double hw = half.hw[a], hh = half.hh[a];
    const double xmin = fmin(fmax(cx - hw, 0.0), 1.0);
    const double xmax = fmin(fmax(cx + hw, 0.0), 1.0);
    const double ymin = fmin(fmax(cy - hh, 0.0), 1.0);
    const double ymax = fmin(fmax(cy + hh, 0.0), 1.0);
    out[idx] = make_float4((float)ymin, (float)xmin, (float)ymax, (float)xmax);
}

// -------------------------------------------------------------- IoU assign ----
// Tiled anchors x GT: the image's G boxes and areas sit in LDS; each thread
// owns one anchor and walks the GT list (LDS broadcast reads), keeping the
// first maximum.  Op order of module/data.py:82-96.
__global__ void __launch_bounds__(FRCNN_TARGET_BLOCK > 256 ? FRCNN_TARGET_BLOCK : 256)
iou_assign_kernel(const float4* __restrict__ anchors, int N, const float4* __restrict__ gt, int G,
                  float pos_thr, float neg_thr, float* __restrict__ max_iou,
                  int32_t* __restrict__ best_gt, uint8_t* __restrict__ flags,
                  float* __restrict__ iou_full) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_gt = reinterpret_cast<float4*>(smem);
    float* s_size = reinterpret_cast<float*>(s_gt + G);
    const int b = blockIdx.y;
    const float4* g_img = gt + (size_t)b * G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float4 v = g_img[g];
        s_gt[g] = v;
        s_size[g] = (v.z - v.x) * (v.w - v.y);
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4 a = anchors[i];
    const float a_size = (a.z - a.x) * (a.w - a.y);
    float best = 0.0f;
    int arg = 0;
    float* full = iou_full ? iou_full + ((size_t)b * N + i) * G : nullptr;
    for (int g = 0; g < G; ++g) {
        const float4 q = s_gt[g];
        const float xmin = fmaxf(a.y, q.y);
        const float ymin = fmaxf(a.x, q.x);
        const float xmax = fminf(a.w, q.w);
        const float ymax = fminf(a.z, q.z);
        const float inter = (xmax - xmin) * (ymax - ymin);
        const float inter2 = (xmax > xmin && ymax > ymin) ? inter : 0.0f;
        const float inter3 = inter2 > 0.0f ? inter2 : 0.0f;
        const float uni = s_size[g] + a_size - inter3;
        const float iou = inter3 / uni;
        if (full) full[g] = iou;
        if (g == 0 || iou > best) { best = iou; arg = g; }
    }
    const bool valid = a.x > 0.0f && a.x < 1.0f && a.y > 0.0f && a.y < 1.0f &&
                       a.z > 0.0f && a.z < 1.0f && a.w > 0.0f && a.w < 1.0f;
    uint8_t f = 0;
    if (valid) {
        f = FRCNN_F_VALID;
        if (best > pos_thr) f |= FRCNN_F_POSITIVE;
        if (best < neg_thr) f |= FRCNN_F_NEGATIVE;
    }
    const size_t o = (size_t)b * N + i;
    max_iou[o] = best;
    best_gt[o] = arg;
    flags[o] = f;
}

// ------------------------------------------------------------ decode/encode ----
__global__ void __launch_bounds__(256)
decode_kernel(const float4* __restrict__ anchors, const float4* __restrict__ deltas,
              long long n_anchor, long long total, int clip_wh, float4* __restrict__ boxes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    boxes[i] = decode_box(anchors[i % n_anchor], deltas[i], clip_wh != 0);
}

__global__ void __launch_bounds__(256)
encode_kernel(const float4* __restrict__ anchor, const float4* __restrict__ gt, long long n,
              int clip_wh, float4* __restrict__ target) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    target[i] = encode_box(anchor[i], gt[i], clip_wh != 0);
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_anchors(float* anchors, int H, int W, const double* scales, int n_scales,
                             const double* ratios, int n_ratios, double base_w, double base_h,
                             frcnn_stream_t stream) {
    if (!anchors || !scales || !ratios || H <= 0 || W <= 0 || n_scales <= 0 || n_ratios <= 0 ||
        !(base_w > 0.0) || !(base_h > 0.0))
        return FRCNN_E_BADARG;
    const int A = n_scales * n_ratios;
    if (A > 64) return FRCNN_E_UNSUPPORTED;
    if ((long long)H * W * A > (1ll << 30)) return FRCNN_E_UNSUPPORTED;
    AnchorHalf half;
    const double aspect = base_w / base_h;   // exactly 1.0 for the reference's square input
    for (int s = 0; s < n_scales; ++s)
        for (int r = 0; r < n_ratios; ++r) {
            const double q = scales[s] / base_w;                 // anchor.py:25
            const double w = sqrt((q * q) / ratios[r]);
            const double h = (w * ratios[r]) * aspect;           // anchor.py:26
            half.hw[s * n_ratios + r] = w / 2.0;                 // anchor.py:27
            half.hh[s * n_ratios + r] = h / 2.0;
        }
    const int total = H * W * A;
    FRCNN_LAUNCH(anchors_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<float4*>(anchors), H, W, A, half);
    return FRCNN_OK;
}

extern "C" int frcnn_iou_assign(const float* anchors, int N, const float* gt, int B, int G,
                                float pos_thr, float neg_thr, float* max_iou, int32_t* best_gt,
                                uint8_t* flags, float* iou_full, frcnn_stream_t stream) {
    if (!anchors || !gt || !max_iou || !best_gt || !flags || N <= 0 || B <= 0 || G <= 0)
        return FRCNN_E_BADARG;
    if (G > 2048 || B > 65535) return FRCNN_E_UNSUPPORTED;
    const size_t lds = (size_t)G * 20;
    // one-wave workgroups by default: they also get wave slots while another stream's kernel
    // fills the chip (workgroups of several waves wait until that grid has drained)
    FRCNN_LAUNCH(iou_assign_kernel, dim3((N + FRCNN_TARGET_BLOCK - 1) / FRCNN_TARGET_BLOCK, B),
                 dim3(FRCNN_TARGET_BLOCK), lds,
                       (hipStream_t)stream, reinterpret_cast<const float4*>(anchors), N,
                       reinterpret_cast<const float4*>(gt), G, pos_thr, neg_thr, max_iou, best_gt,
                       flags, iou_full);
    return FRCNN_OK;
}

extern "C" int frcnn_decode_clip(const float* anchors, const float* deltas, int B,
                                 int64_t n_anchor, int clip_wh, float* boxes,
                                 frcnn_stream_t stream) {
    if (!anchors || !deltas || !boxes || B <= 0 || n_anchor <= 0) return FRCNN_E_BADARG;
    const long long total = (long long)B * n_anchor;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    FRCNN_LAUNCH(decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(anchors),
                       reinterpret_cast<const float4*>(deltas), (long long)n_anchor, total, clip_wh,
                       reinterpret_cast<float4*>(boxes));
    return FRCNN_OK;
}

extern "C" int frcnn_encode(const float* anchor, const float* gt, int64_t n, int clip_wh,
                            float* target, frcnn_stream_t stream) {
    if (!anchor || !gt || !target || n <= 0) return FRCNN_E_BADARG;
    const long long blocks = ((long long)n + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    FRCNN_LAUNCH(encode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(anchor), reinterpret_cast<const float4*>(gt),
                       (long long)n, clip_wh, reinterpret_cast<float4*>(target));
    return FRCNN_OK;
}
