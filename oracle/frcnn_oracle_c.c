/* C restatement of the heavy stages of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Same float32 operation order as oracle/frcnn_oracle.py, which follows
 * Faster_RCNN/module/{data,model}.py of the reference (file:line cited per
 * function) and the TensorFlow op contracts for top_k, combined NMS,
 * crop_and_resize and MaxPool2D.  PARITY UNPINNED for the same reason as the
 * numpy oracle: the reference ships no tests or vectors and TensorFlow cannot
 * be installed here; tests/test_oracle_c.py checks this file bit-for-bit
 * against the numpy restatement and the committed golden fixtures.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* anchors x GT IoU, first-maximum argmax.  module/data.py:82-100 */
void oc_iou_argmax(const float* anchors, int N, const float* gt, int G, float* max_iou,
                   int32_t* best) {
    for (int i = 0; i < N; ++i) {
        const float* a = anchors + 4 * i;
        const float a_size = (a[2] - a[0]) * (a[3] - a[1]);
        float bestv = 0.f;
        int arg = 0;
        for (int g = 0; g < G; ++g) {
            const float* q = gt + 4 * g;
            const float g_size = (q[2] - q[0]) * (q[3] - q[1]);
            const float xmin = a[1] > q[1] ? a[1] : q[1];
            const float ymin = a[0] > q[0] ? a[0] : q[0];
            const float xmax = a[3] < q[3] ? a[3] : q[3];
            const float ymax = a[2] < q[2] ? a[2] : q[2];
            const float inter = (xmax - xmin) * (ymax - ymin);
            const float inter2 = (xmax > xmin && ymax > ymin) ? inter : 0.f;
            const float inter3 = inter2 > 0.f ? inter2 : 0.f;
            const float uni = g_size + a_size - inter3;
            const float iou = inter3 / uni;
            if (g == 0 || iou > bestv) { bestv = iou; arg = g; }
        }
        max_iou[i] = bestv;
        best[i] = arg;
    }
}

/* tf.math.top_k: descending, equal values -> lower index first.  module/model.py:45 */
typedef struct { float v; int32_t i; } oc_pair;
static int oc_cmp(const void* pa, const void* pb) {
    const oc_pair* a = (const oc_pair*)pa;
    const oc_pair* b = (const oc_pair*)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->i > b->i) - (a->i < b->i);
}
void oc_topk(const float* scores, int N, int K, float* vals, int32_t* idx) {
    oc_pair* p = (oc_pair*)malloc(sizeof(oc_pair) * (size_t)N);
    for (int i = 0; i < N; ++i) { p[i].v = scores[i]; p[i].i = i; }
    qsort(p, (size_t)N, sizeof(oc_pair), oc_cmp);
    for (int k = 0; k < K; ++k) { vals[k] = p[k].v; idx[k] = p[k].i; }
    free(p);
}

/* decode + clip of selected anchors.  module/data.py:163-175 */
static float oc_clip01(float v) { v = v > 0.f ? v : 0.f; return v < 1.f ? v : 1.f; }
void oc_decode_gather(const float* anchors, const float* deltas, const int32_t* idx, int K,
                      float* boxes) {
    for (int k = 0; k < K; ++k) {
        const float* a = anchors + 4 * (size_t)idx[k];
        const float* t = deltas + 4 * (size_t)idx[k];
        const float w_a = a[3] - a[1], h_a = a[2] - a[0];
        const float x = t[0] * w_a * 0.1f + a[1];
        const float y = t[1] * h_a * 0.1f + a[0];
        const float w = expf(t[2] * 0.2f) * w_a;
        const float h = expf(t[3] * 0.2f) * h_a;
        const float x_max = x + w, y_max = y + h;
        boxes[4 * k + 0] = oc_clip01(y); boxes[4 * k + 1] = oc_clip01(x);
        boxes[4 * k + 2] = oc_clip01(y_max); boxes[4 * k + 3] = oc_clip01(x_max);
    }
}

/* TF NMS IoU + greedy loop (most recent selected first, strict >).  module/model.py:51 */
static float oc_nms_iou(const float* bi, const float* bj) {
    const float ymin_i = fminf(bi[0], bi[2]), xmin_i = fminf(bi[1], bi[3]);
    const float ymax_i = fmaxf(bi[0], bi[2]), xmax_i = fmaxf(bi[1], bi[3]);
    const float ymin_j = fminf(bj[0], bj[2]), xmin_j = fminf(bj[1], bj[3]);
    const float ymax_j = fmaxf(bj[0], bj[2]), xmax_j = fmaxf(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    return inter / (area_i + area_j - inter);
}
int oc_nms(const float* boxes, int K, float thr, int post_n, float* proposed, int32_t* keep) {
    int n = 0;
    for (int i = 0; i < K && n < post_n; ++i) {
        int ok = 1;
        for (int j = n - 1; j >= 0; --j)
            if (oc_nms_iou(boxes + 4 * (size_t)i, boxes + 4 * (size_t)keep[j]) > thr) { ok = 0; break; }
        if (ok) keep[n++] = i;
    }
    for (int s = 0; s < post_n; ++s) {
        for (int c = 0; c < 4; ++c)
            proposed[4 * s + c] = s < n ? oc_clip01(boxes[4 * (size_t)keep[s] + c]) : 0.f;
        if (s >= n) keep[s] = -1;
    }
    return n;
}

/* crop_and_resize (bilinear, extrapolation 0) + pool x pool max, first maximum wins.
 * module/model.py:53-55 and :30.  fmap [H,W,C] of one image; rois [R,4];
 * out [R,ch/pool,cw/pool,C]; winner may be NULL. */
static void oc_axis(float c1, float c2, int dim, int n, int s, int* lo, int* hi, float* fr, int* ok) {
    float in;
    if (n > 1) {
        const float scale = ((c2 - c1) * (float)(dim - 1)) / (float)(n - 1);
        in = c1 * (float)(dim - 1) + (float)s * scale;
    } else {
        in = (float)(0.5 * (double)(c1 + c2) * (double)(dim - 1));
    }
    *ok = (in >= 0.f && in <= (float)(dim - 1));
    const float f = floorf(in), c = ceilf(in);
    *fr = in - f;
    *lo = *ok ? (int)f : 0;
    *hi = *ok ? (int)c : 0;
}
void oc_roi_crop_pool(const float* fmap, int H, int W, int C, const float* rois, int R, int ch,
                      int cw, int pool, float* out, uint8_t* winner) {
    const int ph = ch / pool, pw = cw / pool;
    int ylo[64], yhi[64], yok[64], xlo[64], xhi[64], xok[64];
    float yfr[64], xfr[64];
    float* tmp = (float*)malloc(sizeof(float) * (size_t)C);
    for (int r = 0; r < R; ++r) {
        const float* b = rois + 4 * (size_t)r;
        for (int s = 0; s < ch; ++s) oc_axis(b[0], b[2], H, ch, s, &ylo[s], &yhi[s], &yfr[s], &yok[s]);
        for (int s = 0; s < cw; ++s) oc_axis(b[1], b[3], W, cw, s, &xlo[s], &xhi[s], &xfr[s], &xok[s]);
        for (int p = 0; p < ph; ++p)
            for (int q = 0; q < pw; ++q) {
                float* o = out + (((size_t)r * ph + p) * pw + q) * C;
                uint8_t* wn = winner ? winner + (((size_t)r * ph + p) * pw + q) * C : NULL;
                for (int k = 0; k < pool * pool; ++k) {
                    const int sy = p * pool + k / pool, sx = q * pool + k % pool;
                    const int ok = yok[sy] && xok[sx];
                    const float* tl = fmap + ((size_t)ylo[sy] * W + xlo[sx]) * C;
                    const float* tr = fmap + ((size_t)ylo[sy] * W + xhi[sx]) * C;
                    const float* bl = fmap + ((size_t)yhi[sy] * W + xlo[sx]) * C;
                    const float* br = fmap + ((size_t)yhi[sy] * W + xhi[sx]) * C;
                    const float lx = xfr[sx], ly = yfr[sy];
                    if (ok) {
                        for (int c = 0; c < C; ++c) {
                            const float top = tl[c] + (tr[c] - tl[c]) * lx;
                            const float bot = bl[c] + (br[c] - bl[c]) * lx;
                            tmp[c] = top + (bot - top) * ly;
                        }
                    } else {
                        memset(tmp, 0, sizeof(float) * (size_t)C);
                    }
                    if (k == 0) {
                        memcpy(o, tmp, sizeof(float) * (size_t)C);
                        if (wn) memset(wn, 0, (size_t)C);
                    } else if (wn) {
                        for (int c = 0; c < C; ++c)
                            if (tmp[c] > o[c]) { o[c] = tmp[c]; wn[c] = (uint8_t)k; }
                    } else {
                        for (int c = 0; c < C; ++c) o[c] = tmp[c] > o[c] ? tmp[c] : o[c];
                    }
                }
            }
    }
    free(tmp);
}

/* Adjoint of oc_roi_crop_pool with respect to fmap, one image (absent in the reference: the
 * crop is under stop_gradient, module/model.py:71).  Same definition as the numpy
 * roi_crop_pool_bwd: the pooled gradient goes to the recorded winner sample only, an
 * extrapolated sample passes nothing, dtop = (1-ly)*g, dbot = ly*g, TL += (1-lx)*dtop,
 * TR += lx*dtop, BL += (1-lx)*dbot, BR += lx*dbot, every cell summing in float32 in the
 * order RoI ascending, sample (sy,sx) row-major, corner TL,TR,BL,BR -- which a sequential
 * scatter in exactly that loop order produces.  dY/winner [R,ph,pw,C]; dF [H,W,C] is
 * overwritten.  winner may be NULL for pool == 1. */
void oc_roi_crop_pool_bwd(const float* dY, const uint8_t* winner, const float* rois, int R, int H,
                          int W, int C, int ch, int cw, int pool, float* dF) {
    const int ph = ch / pool, pw = cw / pool;
    int ylo[64], yhi[64], yok[64], xlo[64], xhi[64], xok[64];
    float yfr[64], xfr[64];
    memset(dF, 0, sizeof(float) * (size_t)H * W * C);
    for (int r = 0; r < R; ++r) {
        const float* b = rois + 4 * (size_t)r;
        for (int s = 0; s < ch; ++s) oc_axis(b[0], b[2], H, ch, s, &ylo[s], &yhi[s], &yfr[s], &yok[s]);
        for (int s = 0; s < cw; ++s) oc_axis(b[1], b[3], W, cw, s, &xlo[s], &xhi[s], &xfr[s], &xok[s]);
        for (int sy = 0; sy < ph * pool; ++sy) {
            if (!yok[sy]) continue;
            for (int sx = 0; sx < pw * pool; ++sx) {
                if (!xok[sx]) continue;
                const int p = sy / pool, q = sx / pool;
                const int k = (sy % pool) * pool + (sx % pool);
                const size_t o = (((size_t)r * ph + p) * pw + q) * C;
                const float ly = yfr[sy], lx = xfr[sx];
                const float wt = 1.0f - ly, wb = ly, wl = 1.0f - lx, wr = lx;
                float* tl = dF + ((size_t)ylo[sy] * W + xlo[sx]) * C;
                float* tr = dF + ((size_t)ylo[sy] * W + xhi[sx]) * C;
                float* bl = dF + ((size_t)yhi[sy] * W + xlo[sx]) * C;
                float* br = dF + ((size_t)yhi[sy] * W + xhi[sx]) * C;
                for (int c = 0; c < C; ++c) {
                    if (pool > 1 && winner[o + c] != k) continue;
                    const float g = dY[o + c];
                    const float dtop = wt * g, dbot = wb * g;
                    tl[c] += wl * dtop;
                    tr[c] += wr * dtop;
                    bl[c] += wl * dbot;
                    br[c] += wr * dbot;
                }
            }
        }
    }
}

/* ---- next row f4: Flatten + Dense(4096, relu), Faster_RCNN/module/model.py:31-32 -------------
 * out[m][n] = act(acc + bias[n]), acc = a k-ordered chain of fused multiply-adds from 0:
 * acc = fmaf(x[m][k], w[k][n], acc) for k = 0 .. K-1 (one rounding per product).  That chain is
 * the build-defined summation order of the layer (TensorFlow's MatMul leaves the order to its
 * BLAS); it is what the float32 matrix instruction computes.  The body is built twice: with the
 * FMA instruction set where the host has it (fmaf inlines, the n loop vectorises) and through
 * libm's fmaf otherwise -- the same values either way. */
#define OC_FC1_BODY                                                                              \
    for (long m = 0; m < M; ++m) {                                                               \
        float* o = out + (size_t)m * N;                                                          \
        for (int n = 0; n < N; ++n) o[n] = 0.0f;                                                 \
        for (int k = 0; k < K; ++k) {                                                            \
            const float a = x[(size_t)m * K + k];                                                \
            const float* wr = w + (size_t)k * N;                                                 \
            for (int n = 0; n < N; ++n) o[n] = __builtin_fmaf(a, wr[n], o[n]);                   \
        }                                                                                        \
        for (int n = 0; n < N; ++n) {                                                            \
            float v = o[n];                                                                      \
            if (bias) v = v + bias[n];                                                           \
            if (relu) v = v > 0.0f ? v : 0.0f;                                                   \
            o[n] = v;                                                                            \
        }                                                                                        \
    }
#if defined(__x86_64__)
__attribute__((target("fma,avx2"))) static void oc_fc1_fma(const float* x, const float* w,
                                                           const float* bias, long M, int K, int N,
                                                           int relu, float* out) {
    OC_FC1_BODY
}
#endif
static void oc_fc1_libm(const float* x, const float* w, const float* bias, long M, int K, int N,
                        int relu, float* out) {
    OC_FC1_BODY
}
void oc_fc1(const float* x, const float* w, const float* bias, long M, int K, int N, int relu,
            float* out) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2")) {
        oc_fc1_fma(x, w, bias, M, K, N, relu, out);
        return;
    }
#endif
    oc_fc1_libm(x, w, bias, M, K, N, relu, out);
}
