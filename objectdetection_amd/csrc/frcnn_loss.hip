// Loss assembly for the two sampled mini-batches (the step right after the region-proposal
// path): gathers by batch position / proposal index / class, binary and categorical
// cross-entropy with the Keras clipping rules, smooth-L1 on the positives, 2*cls + reg, and
// the gradients of that total with respect to the network outputs, scattered back to the
// dense output layout.  Reference: Faster_RCNN/model_train.py:598-600, 626-635, 646-657;
// Faster_RCNN/module/loss.py:4-14.
//
// Two launches per side: per-image partial sums (one workgroup per image, fixed reduction
// order), then a finish kernel in which every workgroup re-adds the B partials in image
// order (so all agree bit for bit), workgroup 0 writes the losses and workgroup b scatters
// image b's gradients.  gfx950, wave64, -ffp-contract=off.
#include "frcnn_common.h"

namespace frcnn {

constexpr float KERAS_EPS = 1e-7f;
constexpr int LOSS_BLOCK = 256;

// sum of three per-thread values over the workgroup, fixed order: lanes by xor-shuffle,
// then waves 0..3 in order
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float (*sh)[3]) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d);
        b += __shfl_xor(b, d);
        c += __shfl_xor(c, d);
    }
    if (lane_id() == 0) { sh[wave_id()][0] = a; sh[wave_id()][1] = b; sh[wave_id()][2] = c; }
    __syncthreads();
    a = 0.f; b = 0.f; c = 0.f;
#pragma unroll
    for (int w = 0; w < LOSS_BLOCK / 64; ++w) { a += sh[w][0]; b += sh[w][1]; c += sh[w][2]; }
    __syncthreads();
}

// smooth-L1 of module/loss.py:9-14 with threshold 1: value and d/d(y_pred)
__device__ __forceinline__ float huber(float y_true, float y_pred, float& dpred) {
    const float e = y_true - y_pred;
    const float ae = fabsf(e);
    if (ae < 1.0f) { dpred = -e; return e * e / 2.0f; }
    dpred = e > 0.0f ? -1.0f : 1.0f;
    return 1.0f * ae - 0.5f;
}

// Keras binary cross-entropy on probabilities (backend.binary_crossentropy, eager path)
__device__ __forceinline__ float bce(float y, float p, float& dp) {
    const float pc = fminf(fmaxf(p, KERAS_EPS), 1.0f - KERAS_EPS);
    const float a = pc + KERAS_EPS, b = 1.0f - pc + KERAS_EPS;
    float v = y * logf(a);
    v += (1.0f - y) * logf(b);
    const bool pass = p >= KERAS_EPS && p <= 1.0f - KERAS_EPS;       // clip_by_value's gradient
    dp = pass ? -(y / a - (1.0f - y) / b) : 0.0f;
    return -v;
}

// ---------------------------------------------------------------- RPN side ----
// partial[b] = {sum of BCE elements, sum of smooth-L1 elements, #positives, #bad indices}
__global__ void __launch_bounds__(LOSS_BLOCK)
rpn_loss_partial_kernel(const float* __restrict__ pred_obj, const float* __restrict__ pred_reg,
                        const int64_t* __restrict__ pos, const float* __restrict__ label,
                        const float* __restrict__ reg_gt, int N, int S, int H, int W, int A,
                        float* __restrict__ partial) {
    __shared__ float sh[LOSS_BLOCK / 64][3];
    const int b = blockIdx.x;
    float s_cls = 0.f, s_reg = 0.f, n_pos = 0.f, n_bad = 0.f;
    for (int i = threadIdx.x; i < S; i += LOSS_BLOCK) {
        const int64_t* p3 = pos + ((size_t)b * S + i) * 3;
        const int64_t y = p3[0], x = p3[1], a = p3[2];
        if (y < 0 || y >= H || x < 0 || x >= W || a < 0 || a >= A) { n_bad += 1.f; continue; }
        const size_t idx = (size_t)b * N + (size_t)((y * W + x) * A + a);
        const float lab = label[(size_t)b * S + i];
        float d;
        s_cls += bce(lab, pred_obj[idx], d);
        if ((int64_t)lab == 1) {
            n_pos += 1.f;
            const float4 t = reinterpret_cast<const float4*>(reg_gt)[(size_t)b * S + i];
            const float4 p = reinterpret_cast<const float4*>(pred_reg)[idx];
            s_reg += huber(t.x, p.x, d);
            s_reg += huber(t.y, p.y, d);
            s_reg += huber(t.z, p.z, d);
            s_reg += huber(t.w, p.w, d);
        }
    }
    block_sum3(s_cls, s_reg, n_pos, sh);
    float dummy0 = 0.f, dummy1 = 0.f;
    block_sum3(n_bad, dummy0, dummy1, sh);
    if (threadIdx.x == 0) {
        float* o = partial + (size_t)b * 4;
        o[0] = s_cls; o[1] = s_reg; o[2] = n_pos; o[3] = n_bad;
    }
}

// totals of the B partial rows, in image order (every caller gets the same bits)
__device__ __forceinline__ void loss_totals(const float* partial, int B, float& cls, float& reg,
                                            float& npos) {
    cls = 0.f; reg = 0.f; npos = 0.f;
    for (int i = 0; i < B; ++i) {
        cls += partial[(size_t)i * 4 + 0];
        reg += partial[(size_t)i * 4 + 1];
        npos += partial[(size_t)i * 4 + 2];
    }
}

__global__ void __launch_bounds__(LOSS_BLOCK)
rpn_loss_finish_kernel(const float* __restrict__ pred_obj, const float* __restrict__ pred_reg,
                       const int64_t* __restrict__ pos, const float* __restrict__ label,
                       const float* __restrict__ reg_gt, int B, int N, int S, int H, int W, int A,
                       const float* __restrict__ partial, float* __restrict__ loss,
                       float* __restrict__ d_obj, float* __restrict__ d_reg) {
    const int b = blockIdx.x;
    float t_cls, t_reg, t_pos;
    loss_totals(partial, B, t_cls, t_reg, t_pos);
    const float n_cls = (float)B * (float)S;
    const float n_reg = 4.0f * t_pos;
    if (b == 0 && threadIdx.x == 0) {
        const float l_cls = t_cls / n_cls;
        const float l_reg = n_reg > 0.0f ? t_reg / n_reg : 0.0f;         // div_no_nan
        loss[0] = l_cls; loss[1] = l_reg; loss[2] = 2.0f * l_cls + l_reg;
    }
    if (!d_obj && !d_reg) return;
    const float w_cls = 2.0f / n_cls;
    const float w_reg = n_reg > 0.0f ? 1.0f / n_reg : 0.0f;
    // The mini-batch can list one anchor twice (a negative's position is its rank in the
    // valid-anchor list, module/data.py:108-118, and that rank can equal a positive's anchor
    // id), and the gradient of a gather accumulates.  The first sample of an anchor owns its
    // gradient and adds the later ones in sample order: no atomics, one fixed order.
    auto anchor_of = [&](int i) -> long long {
        const int64_t* p3 = pos + ((size_t)b * S + i) * 3;
        const int64_t y = p3[0], x = p3[1], a = p3[2];
        if (y < 0 || y >= H || x < 0 || x >= W || a < 0 || a >= A) return -1;
        return (long long)((y * W + x) * A + a);
    };
    for (int i = threadIdx.x; i < S; i += LOSS_BLOCK) {
        const long long an = anchor_of(i);
        if (an < 0) continue;
        bool owner = true;
        for (int j = 0; j < i; ++j)
            if (anchor_of(j) == an) { owner = false; break; }
        if (!owner) continue;
        const size_t idx = (size_t)b * N + (size_t)an;
        const float p_obj = pred_obj[idx];
        const float4 p = reinterpret_cast<const float4*>(pred_reg)[idx];
        float g_obj = 0.f;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any_pos = false;
        for (int j = i; j < S; ++j) {
            if (j != i && anchor_of(j) != an) continue;
            const float lab = label[(size_t)b * S + j];
            float d;
            (void)bce(lab, p_obj, d);
            g_obj += w_cls * d;
            if ((int64_t)lab == 1) {
                const float4 t = reinterpret_cast<const float4*>(reg_gt)[(size_t)b * S + j];
                float4 h;
                (void)huber(t.x, p.x, h.x); (void)huber(t.y, p.y, h.y);
                (void)huber(t.z, p.z, h.z); (void)huber(t.w, p.w, h.w);
                g.x += h.x * w_reg; g.y += h.y * w_reg; g.z += h.z * w_reg; g.w += h.w * w_reg;
                any_pos = true;
            }
        }
        if (d_obj) d_obj[idx] = g_obj;
        if (d_reg && any_pos) reinterpret_cast<float4*>(d_reg)[idx] = g;
    }
}

// --------------------------------------------------------------- head side ----
// Keras categorical cross-entropy on probabilities: q = p / sum(p), clip, -sum(y log q).
// K <= 64 classes; the row is read twice (sum, then terms).
__device__ __forceinline__ float cce_row(const float* __restrict__ y, const float* __restrict__ p,
                                         int K, float& sum_p, float& sum_gp) {
    float sp = 0.f;
    for (int k = 0; k < K; ++k) sp += p[k];
    float loss = 0.f, gp = 0.f;
    for (int k = 0; k < K; ++k) {
        const float q = p[k] / sp;
        const float qc = fminf(fmaxf(q, KERAS_EPS), 1.0f - KERAS_EPS);
        loss += y[k] * logf(qc);
        const bool pass = q >= KERAS_EPS && q <= 1.0f - KERAS_EPS;
        const float g = pass ? -y[k] / qc : 0.0f;                      // dL/dq_k
        gp += g * p[k];
    }
    sum_p = sp; sum_gp = gp;
    return -loss;
}

__device__ __forceinline__ int argmax_row(const float* __restrict__ y, int K) {
    int c = 0;
    float best = y[0];
    for (int k = 1; k < K; ++k)
        if (y[k] > best) { best = y[k]; c = k; }                       // first maximum
    return c;
}

__global__ void __launch_bounds__(LOSS_BLOCK)
head_loss_partial_kernel(const float* __restrict__ cls_prob, const float* __restrict__ reg,
                         const int64_t* __restrict__ tindex, const float* __restrict__ label,
                         const int32_t* __restrict__ gt_mask, const float* __restrict__ gt_coord,
                         int R, int S, int K, float* __restrict__ partial) {
    __shared__ float sh[LOSS_BLOCK / 64][3];
    const int b = blockIdx.x;
    float s_cls = 0.f, s_reg = 0.f, n_pos = 0.f, n_bad = 0.f;
    for (int i = threadIdx.x; i < S; i += LOSS_BLOCK) {
        const int64_t r = tindex[(size_t)b * S + i];
        if (r < 0 || r >= R) { n_bad += 1.f; continue; }
        const float* y = label + ((size_t)b * S + i) * K;
        const float* p = cls_prob + ((size_t)b * R + (size_t)r) * K;
        float sp, gp;
        s_cls += cce_row(y, p, K, sp, gp);
        if (gt_mask[(size_t)b * S + i] != 0) {
            n_pos += 1.f;
            const int c = argmax_row(y, K);
            const float4 t = reinterpret_cast<const float4*>(gt_coord)[(size_t)b * S + i];
            const float4 q = reinterpret_cast<const float4*>(reg)[((size_t)b * R + (size_t)r) * K + c];
            float d;
            s_reg += huber(fminf(fmaxf(t.x, -100.f), 100.f), q.x, d);
            s_reg += huber(fminf(fmaxf(t.y, -100.f), 100.f), q.y, d);
            s_reg += huber(fminf(fmaxf(t.z, -100.f), 100.f), q.z, d);
            s_reg += huber(fminf(fmaxf(t.w, -100.f), 100.f), q.w, d);
        }
    }
    block_sum3(s_cls, s_reg, n_pos, sh);
    float dummy0 = 0.f, dummy1 = 0.f;
    block_sum3(n_bad, dummy0, dummy1, sh);
    if (threadIdx.x == 0) {
        float* o = partial + (size_t)b * 4;
        o[0] = s_cls; o[1] = s_reg; o[2] = n_pos; o[3] = n_bad;
    }
}

__global__ void __launch_bounds__(LOSS_BLOCK)
head_loss_finish_kernel(const float* __restrict__ cls_prob, const float* __restrict__ reg,
                        const int64_t* __restrict__ tindex, const float* __restrict__ label,
                        const int32_t* __restrict__ gt_mask, const float* __restrict__ gt_coord,
                        int B, int R, int S, int K, const float* __restrict__ partial,
                        float* __restrict__ loss, float* __restrict__ d_cls,
                        float* __restrict__ d_reg) {
    const int b = blockIdx.x;
    float t_cls, t_reg, t_pos;
    loss_totals(partial, B, t_cls, t_reg, t_pos);
    const float n_cls = (float)B * (float)S;
    const float n_reg = 4.0f * t_pos;
    if (b == 0 && threadIdx.x == 0) {
        const float l_cls = t_cls / n_cls;
        const float l_reg = n_reg > 0.0f ? t_reg / n_reg : 0.0f;
        loss[0] = l_cls; loss[1] = l_reg; loss[2] = 2.0f * l_cls + l_reg;
    }
    if (!d_cls && !d_reg) return;
    const float w_cls = 2.0f / n_cls;
    const float w_reg = n_reg > 0.0f ? 1.0f / n_reg : 0.0f;
    // tindex rows are distinct when they come from frcnn_sample_proposals; a caller's own list
    // may repeat a proposal, and the gradient of a gather accumulates: the first sample of a
    // proposal owns its rows and adds the later ones in sample order.
    for (int i = threadIdx.x; i < S; i += LOSS_BLOCK) {
        const int64_t r = tindex[(size_t)b * S + i];
        if (r < 0 || r >= R) continue;
        bool owner = true;
        for (int j = 0; j < i; ++j)
            if (tindex[(size_t)b * S + j] == r) { owner = false; break; }
        if (!owner) continue;
        const size_t row = (size_t)b * R + (size_t)r;
        const float* p = cls_prob + row * K;
        bool first = true;
        for (int j = i; j < S; ++j) {
            if (j != i && tindex[(size_t)b * S + j] != r) continue;
            const float* y = label + ((size_t)b * S + j) * K;
            if (d_cls) {
                float sp, gp;
                (void)cce_row(y, p, K, sp, gp);
                // dL/dp_j = g_j / sum - (sum_k g_k p_k) / sum^2
                const float tail = gp / (sp * sp);
                for (int k = 0; k < K; ++k) {
                    const float q = p[k] / sp;
                    const float qc = fminf(fmaxf(q, KERAS_EPS), 1.0f - KERAS_EPS);
                    const bool pass = q >= KERAS_EPS && q <= 1.0f - KERAS_EPS;
                    const float g = pass ? -y[k] / qc : 0.0f;
                    const float v = w_cls * (g / sp - tail);
                    if (first) d_cls[row * K + k] = v; else d_cls[row * K + k] += v;
                }
            }
            if (d_reg && gt_mask[(size_t)b * S + j] != 0) {
                const int c = argmax_row(y, K);
                const float4 t = reinterpret_cast<const float4*>(gt_coord)[(size_t)b * S + j];
                const float4 q = reinterpret_cast<const float4*>(reg)[row * K + c];
                float4 g;
                (void)huber(fminf(fmaxf(t.x, -100.f), 100.f), q.x, g.x);
                (void)huber(fminf(fmaxf(t.y, -100.f), 100.f), q.y, g.y);
                (void)huber(fminf(fmaxf(t.z, -100.f), 100.f), q.z, g.z);
                (void)huber(fminf(fmaxf(t.w, -100.f), 100.f), q.w, g.w);
                // d_reg was zero-filled by this call: adding keeps repeated (row, class) pairs
                float4* dst = reinterpret_cast<float4*>(d_reg) + (row * K + c);
                float4 acc = *dst;
                acc.x += g.x * w_reg; acc.y += g.y * w_reg; acc.z += g.z * w_reg; acc.w += g.w * w_reg;
                *dst = acc;
            }
            first = false;
        }
    }
}

// Zero fill as a kernel of this library (a memset node recorded into a HIP graph was seen to
// lose its place in the order on ROCm 7.2; a kernel keeps stream order under capture).
template <typename T>
__global__ void __launch_bounds__(256) zero_fill_kernel(T* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = T{};
}

}  // namespace frcnn

using namespace frcnn;

template <typename T>
static int zero_fill(T* p, size_t n, hipStream_t st) {
    if (!p || n == 0) return FRCNN_OK;
    const size_t want = (n + 255) / 256;
    FRCNN_LAUNCH(zero_fill_kernel<T>, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, p, n);
    return FRCNN_OK;
}

extern "C" int frcnn_rpn_loss(const float* pred_obj, const float* pred_reg, const int64_t* pos,
                              const float* label, const float* reg_gt, int B, int H, int W, int A,
                              int S, float* loss, float* partial, float* d_obj, float* d_reg,
                              frcnn_stream_t stream) {
    if (!pred_obj || !pred_reg || !pos || !label || !reg_gt || !loss || !partial) return FRCNN_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0 || A <= 0 || S <= 0) return FRCNN_E_BADARG;
    if ((long long)H * W * A > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(pred_reg) | reinterpret_cast<uintptr_t>(reg_gt) |
         reinterpret_cast<uintptr_t>(d_reg)) & 15u)
        return FRCNN_E_BADARG;
    const int N = H * W * A;
    hipStream_t st = (hipStream_t)stream;
    if (zero_fill(d_obj, (size_t)B * N, st) != FRCNN_OK ||
        zero_fill(reinterpret_cast<float4*>(d_reg), (size_t)B * N, st) != FRCNN_OK)
        return FRCNN_E_LAUNCH;
    FRCNN_LAUNCH(rpn_loss_partial_kernel, dim3(B), dim3(LOSS_BLOCK), 0, st, pred_obj, pred_reg, pos,
                 label, reg_gt, N, S, H, W, A, partial);
    FRCNN_LAUNCH(rpn_loss_finish_kernel, dim3(B), dim3(LOSS_BLOCK), 0, st, pred_obj, pred_reg, pos,
                 label, reg_gt, B, N, S, H, W, A, partial, loss, d_obj, d_reg);
    return FRCNN_OK;
}

extern "C" int frcnn_head_loss(const float* cls_prob, const float* reg, const int64_t* tindex,
                               const float* label, const int32_t* gt_mask, const float* gt_coord,
                               int B, int R, int S, int K, float* loss, float* partial,
                               float* d_cls, float* d_reg, frcnn_stream_t stream) {
    if (!cls_prob || !reg || !tindex || !label || !gt_mask || !gt_coord || !loss || !partial)
        return FRCNN_E_BADARG;
    if (B <= 0 || R <= 0 || S <= 0 || K <= 0) return FRCNN_E_BADARG;
    if (K > 64) return FRCNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(reg) | reinterpret_cast<uintptr_t>(gt_coord) |
         reinterpret_cast<uintptr_t>(d_reg)) & 15u)
        return FRCNN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (zero_fill(d_cls, (size_t)B * R * K, st) != FRCNN_OK ||
        zero_fill(reinterpret_cast<float4*>(d_reg), (size_t)B * R * K, st) != FRCNN_OK)
        return FRCNN_E_LAUNCH;
    FRCNN_LAUNCH(head_loss_partial_kernel, dim3(B), dim3(LOSS_BLOCK), 0, st, cls_prob, reg, tindex,
                 label, gt_mask, gt_coord, R, S, K, partial);
    FRCNN_LAUNCH(head_loss_finish_kernel, dim3(B), dim3(LOSS_BLOCK), 0, st, cls_prob, reg, tindex,
                 label, gt_mask, gt_coord, B, R, S, K, partial, loss, d_cls, d_reg);
    return FRCNN_OK;
}
