// Fused bilinear RoI crop + max-pool (forward) and its atomic-free adjoint.
// Sampling rule and op order of tf.image.crop_and_resize's CPU kernel:
//   hs = ((y2-y1)*(H-1))/(ch-1);  in_y = y1*(H-1) + y*hs
//   rows/cols whose source coordinate is outside [0, dim-1] produce 0
//   top = TL + (TR-TL)*lx; bottom = BL + (BR-BL)*lx; out = top + (bottom-top)*ly
// followed by the pool x pool, stride pool maximum (first maximum wins).
// gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"

namespace frcnn {

constexpr int ROI_MAX_CROP = 64;

struct AxisGeom {
    int lo[ROI_MAX_CROP];      // floor index
    int hi[ROI_MAX_CROP];      // ceil index
    float frac[ROI_MAX_CROP];  // lerp weight
    int ok[ROI_MAX_CROP];      // 0 -> extrapolated (output 0)
};

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

__device__ __forceinline__ void fill_geom(AxisGeom& gy, AxisGeom& gx, float4 box, int H, int W,
                                          int ch, int cw) {
    const int t = threadIdx.x;
    if (t < ch) axis_sample(box.x, box.z, H, ch, t, gy.lo[t], gy.hi[t], gy.frac[t], gy.ok[t]);
    if (t >= 64 && t - 64 < cw) {
        const int s = t - 64;
        axis_sample(box.y, box.w, W, cw, s, gx.lo[s], gx.hi[s], gx.frac[s], gx.ok[s]);
    }
}

__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float w) {
    float4 r;
    r.x = a.x + (b.x - a.x) * w;
    r.y = a.y + (b.y - a.y) * w;
    r.z = a.z + (b.z - a.z) * w;
    r.w = a.w + (b.w - a.w) * w;
    return r;
}

// ---- channels-last fast path: one workgroup per RoI, one float4 of channels
// per thread, bins walked in row-major order.  Every load and store of a wave
// is 1 KiB contiguous.
template <int POOL>
__global__ void __launch_bounds__(128)
roi_fwd_cl_kernel(const float* __restrict__ fmap, int H, int W, int C, long long sb, long long sh,
                  long long sw, const float4* __restrict__ rois, int R, int ch, int cw,
                  float* __restrict__ out, uint8_t* __restrict__ winner) {
    __shared__ AxisGeom gy, gx;
    const int r = blockIdx.x, b = blockIdx.y;
    const float4 box = rois[(size_t)b * R + r];
    fill_geom(gy, gx, box, H, W, ch, cw);
    __syncthreads();
    const int ph = ch / POOL, pw = cw / POOL;
    const float* img = fmap + (size_t)b * sb;
    const size_t obase = ((size_t)b * R + r) * (size_t)(ph * pw) * C;
    const int C4 = C >> 2;
    for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
        const float* chan = img + c4 * 4;
        for (int p = 0; p < ph; ++p) {
            for (int q = 0; q < pw; ++q) {
                float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
                uchar4 arg = make_uchar4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < POOL * POOL; ++k) {
                    const int sy = p * POOL + k / POOL, sx = q * POOL + k % POOL;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gy.ok[sy] && gx.ok[sx]) {          // block-uniform
                        const float* rt = chan + (size_t)gy.lo[sy] * sh;
                        const float* rb = chan + (size_t)gy.hi[sy] * sh;
                        const size_t ol = (size_t)gx.lo[sx] * sw, orr = (size_t)gx.hi[sx] * sw;
                        const float4 tl = *reinterpret_cast<const float4*>(rt + ol);
                        const float4 tr = *reinterpret_cast<const float4*>(rt + orr);
                        const float4 bl = *reinterpret_cast<const float4*>(rb + ol);
                        const float4 br = *reinterpret_cast<const float4*>(rb + orr);
                        const float lx = gx.frac[sx], ly = gy.frac[sy];
                        const float4 top = lerp4(tl, tr, lx);
                        const float4 bot = lerp4(bl, br, lx);
                        v = lerp4(top, bot, ly);
                    }
                    if (k == 0) {
                        best = v;
                    } else {
                        if (v.x > best.x) { best.x = v.x; arg.x = k; }
                        if (v.y > best.y) { best.y = v.y; arg.y = k; }
                        if (v.z > best.z) { best.z = v.z; arg.z = k; }
                        if (v.w > best.w) { best.w = v.w; arg.w = k; }
                    }
                }
                const size_t o = obase + (size_t)(p * pw + q) * C + c4 * 4;
                *reinterpret_cast<float4*>(out + o) = best;
                if (winner) *reinterpret_cast<uchar4*>(winner + o) = arg;
            }
        }
    }
}

// ---- channels-last "row walker" (C % 64 == 0, crop <= 16x16).
// Workgroup = one RoI x one 64-channel slice (slice = blockIdx.x % nslice: with
// 8 slices the blocks dispatched together cover one RoI and each XCD keeps
// seeing the same eighth of the feature map); thread = (channel quad, pooled
// column q); no LDS, no barrier: every wave is self-contained.
//
// The sample rows of a box touch an ascending list of distinct feature rows.
// Each thread walks that list once: per feature row it loads the four cells
// its POOL sample columns need (the next row already in flight), lerps them
// horizontally, and keeps the last two lerped rows in registers; every sample
// row whose bottom row has arrived is then finished with one vertical lerp per
// sample column and folded into the running max of its pool window.  A
// feature row shared by several sample rows (small boxes) is fetched once.
//
// The row list and the per-sample-row (top, bottom, weight) live one entry per
// lane and are read with v_readlane by the uniform loop counters, so the loop
// control runs on the scalar unit.
constexpr int RW_CS = 64;
constexpr int RW_LANES = RW_CS / 4;
constexpr int RW_MAXCROP = 16;

struct RowWalk {
    int t_lo, t_hi;     // lane s: unique-row index of sample row s (top, bottom); -1 extrapolated
    float t_fr;         // lane s: vertical lerp weight of sample row s
    int t_row;          // lane j: feature row of unique-row j
    int nr;             // number of unique rows (uniform)
    bool walk;          // false: flipped box, t_lo/t_hi hold raw rows
};

// Every wave builds the table for itself.  Lane 2s / 2s+1 = floor / ceil row of
// sample s; in an ascending sequence a value can only repeat one of the
// previous two entries, so two shuffles find the first occurrences.
__device__ __forceinline__ RowWalk build_row_walk(float y1, float y2, int H, int ch) {
    const int lane = lane_id();
    const int s = lane >> 1, which = lane & 1;
    int lo = 0, hi = 0, ok = 0;
    float fr = 0.f;
    if (s < ch) axis_sample(y1, y2, H, ch, s, lo, hi, fr, ok);
    const bool valid = s < ch && ok;
    RowWalk w;
    w.walk = !(y2 < y1);
    const int v = valid ? (which ? hi : lo) : -1 - lane;
    const int v1 = __shfl_up(v, 1), v2 = __shfl_up(v, 2);
    const bool is_first = valid && (lane < 1 || v != v1) && (lane < 2 || v != v2);
    const unsigned long long fm = __ballot(is_first);
    const unsigned long long le = fm & (lanemask_lt() | (1ull << lane));
    const int cnt = __popcll(le);
    const int F = le ? 63 - __clzll((long long)le) : 0;
    const int vF = __shfl(v, F);
    int idx = (v == vF) ? cnt - 1 : cnt - 2;
    if (!w.walk) idx = v;
    const int src = (2 * lane) & 63;
    const int s_ok = __shfl((int)valid, src);
    w.t_lo = __shfl(idx, src);
    w.t_hi = __shfl(idx, src | 1);
    w.t_fr = __shfl(fr, src);
    if (!(lane < ch && s_ok)) { w.t_lo = -1; w.t_hi = -1; }
    // scatter: unique-row slot (cnt-1) <- v; lanes that are not a first occurrence aim at lane 63
    w.t_row = __builtin_amdgcn_ds_permute((is_first ? cnt - 1 : 63) << 2, v);
    w.nr = __popcll(fm);
    return w;
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rlf(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff,
                                           float4 v) {
    f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, soff, 0);
}
template <bool B> struct BoolTag { static constexpr bool value = B; };

template <int POOL, bool WIN>
__global__ void __launch_bounds__(256)
roi_fwd_walk_kernel(const float* __restrict__ fmap, int H, int W, int C, long long sb, long long sh,
                    long long sw, const float4* __restrict__ rois, int R, int ch, int cw,
                    int nslice, int nsub, float* __restrict__ out, uint8_t* __restrict__ winner) {
    const int tid = threadIdx.x;
    const int slice = blockIdx.x % nslice, r = blockIdx.x / nslice, b = blockIdx.y;
    const float4 box = rois[(size_t)b * R + r];
    const RowWalk rw = build_row_walk(box.x, box.z, H, ch);
    const int ph = ch / POOL, pw = cw / POOL;
    const int cq = tid & (RW_LANES - 1), q = tid / RW_LANES;
    // this thread's sample columns: byte offsets inside a feature row (32-bit)
    uint32_t xlo[POOL], xhi[POOL];
    float xfr[POOL];
    bool xok[POOL];
    bool xall = true;
    const uint32_t chan_off = (uint32_t)(slice * nsub * RW_CS + cq * 4) * 4u;
#pragma unroll
    for (int k = 0; k < POOL; ++k) {
        int lo = 0, hi = 0, ok = 0;
        float fr = 0.f;
        const int sx = q * POOL + k;
        if (sx < cw) axis_sample(box.y, box.w, W, cw, sx, lo, hi, fr, ok);
        xlo[k] = (uint32_t)lo * (uint32_t)sw * 4u + chan_off;
        xhi[k] = (uint32_t)hi * (uint32_t)sw * 4u + chan_off;
        xfr[k] = fr;
        xok[k] = sx < cw && ok;
        xall = xall && (xok[k] || sx >= cw);
    }
    const bool wave_xall = __all(xall);                   // no extrapolated column in this wave
    if (q >= pw) return;
    // uniform descriptors: this image's feature map; this RoI's output block
    const uint32_t row_bytes = (uint32_t)sh * 4u;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(fmap) + (size_t)b * sb, 0, (uint32_t)H * row_bytes, 0x00020000);
    const size_t oroi = ((size_t)b * R + r) * (size_t)(ph * pw) * C;
    const uint32_t orow_bytes = (uint32_t)pw * (uint32_t)C * 4u;
    const __amdgpu_buffer_rsrc_t obuf = __builtin_amdgcn_make_buffer_rsrc(
        out + oroi, 0, (uint32_t)ph * orow_bytes, 0x00020000);
    const uint32_t ooff = (uint32_t)q * (uint32_t)C * 4u + chan_off;                   // per thread
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nsy = ph * POOL;
    float4 best = zero;
    uchar4 arg = make_uchar4(0, 0, 0, 0);
    uint32_t sub_off = 0;                 // byte offset of the 64-channel sub-slice in progress

    auto fold = [&](int sy, const float4 (&v)[POOL]) {
        const int ky = sy % POOL;
        if (WIN) {
#pragma unroll
            for (int kx = 0; kx < POOL; ++kx) {
                const int k = ky * POOL + kx;
                if (k == 0) {
                    best = v[kx];
                    arg = make_uchar4(0, 0, 0, 0);
                } else {
                    if (v[kx].x > best.x) { best.x = v[kx].x; arg.x = k; }
                    if (v[kx].y > best.y) { best.y = v[kx].y; arg.y = k; }
                    if (v[kx].z > best.z) { best.z = v[kx].z; arg.z = k; }
                    if (v[kx].w > best.w) { best.w = v[kx].w; arg.w = k; }
                }
            }
        } else if (POOL == 1) {
            best = v[0];
        } else if (ky == 0) {
            best.x = vmax2(v[0].x, v[POOL - 1].x); best.y = vmax2(v[0].y, v[POOL - 1].y);
            best.z = vmax2(v[0].z, v[POOL - 1].z); best.w = vmax2(v[0].w, v[POOL - 1].w);
        } else {
            best.x = vmax3(best.x, v[0].x, v[POOL - 1].x); best.y = vmax3(best.y, v[0].y, v[POOL - 1].y);
            best.z = vmax3(best.z, v[0].z, v[POOL - 1].z); best.w = vmax3(best.w, v[0].w, v[POOL - 1].w);
        }
        if (ky == POOL - 1) {
            const uint32_t orow = (uint32_t)(sy / POOL) * orow_bytes + sub_off;        // uniform
            buf_store4(obuf, ooff, orow, best);
            if (WIN)
                *reinterpret_cast<uchar4*>(winner + oroi + ((orow + ooff) >> 2)) = arg;
        }
    };

    if (!rw.walk) {
        // flipped box: rows are not ascending; fetch every corner directly
        for (int sub = 0; sub < nsub; ++sub, sub_off += RW_CS * 4)
        for (int sy = 0; sy < nsy; ++sy) {
            const int lo = rl(rw.t_lo, sy), hi = rl(rw.t_hi, sy);
            const float fr = rlf(rw.t_fr, sy);
            float4 v[POOL];
#pragma unroll
            for (int k = 0; k < POOL; ++k) {
                v[k] = zero;
                if (lo >= 0 && xok[k]) {
                    const uint32_t rt = (uint32_t)lo * row_bytes + sub_off, rb = (uint32_t)hi * row_bytes + sub_off;
                    const float4 top = lerp4(buf_load4(img, xlo[k], rt), buf_load4(img, xhi[k], rt),
                                             xfr[k]);
                    const float4 bot = lerp4(buf_load4(img, xlo[k], rb), buf_load4(img, xhi[k], rb),
                                             xfr[k]);
                    v[k] = lerp4(top, bot, fr);
                }
            }
            fold(sy, v);
        }
        return;
    }

    const int NR = rw.nr;
    float4 hA[POOL], hB[POOL], lAl[POOL], lAr[POOL], lBl[POOL], lBr[POOL];
#pragma unroll
    for (int k = 0; k < POOL; ++k) { hA[k] = hB[k] = lAl[k] = lAr[k] = lBl[k] = lBr[k] = zero; }
    int sy = 0;

    auto load_row = [&](int j, float4 (&ll)[POOL], float4 (&lr)[POOL]) {
        const uint32_t roff = (uint32_t)rl(rw.t_row, j) * row_bytes + sub_off;          // uniform
#pragma unroll
        for (int k = 0; k < POOL; ++k) {
            ll[k] = buf_load4(img, xlo[k], roff);
            lr[k] = buf_load4(img, xhi[k], roff);
        }
    };
    // finish every sample row that needs nothing beyond unique row j
    auto drain = [&](int j, const float4 (&hc)[POOL], const float4 (&hp)[POOL], auto xtag) {
        constexpr bool XALL = decltype(xtag)::value;
        while (sy < nsy) {
            const int lo = rl(rw.t_lo, sy), hi = rl(rw.t_hi, sy);
            if (lo >= 0 && hi > j) break;
            const float fr = rlf(rw.t_fr, sy);
            if (lo < 0) {
                float4 v[POOL];
#pragma unroll
                for (int k = 0; k < POOL; ++k) v[k] = zero;
                fold(sy, v);
            } else if (lo == j) {
                float4 v[POOL];
#pragma unroll
                for (int k = 0; k < POOL; ++k) {
                    v[k] = lerp4(hc[k], hc[k], fr);
                    if (!XALL && !xok[k]) v[k] = zero;
                }
                fold(sy, v);
            } else {
                float4 v[POOL];
#pragma unroll
                for (int k = 0; k < POOL; ++k) {
                    v[k] = lerp4(hp[k], hc[k], fr);
                    if (!XALL && !xok[k]) v[k] = zero;
                }
                fold(sy, v);
            }
            ++sy;
        }
    };
    auto step = [&](int j, float4 (&hc)[POOL], const float4 (&hp)[POOL], const float4 (&cl)[POOL],
                    const float4 (&cr)[POOL], float4 (&nl)[POOL], float4 (&nr_)[POOL], auto xtag) {
        load_row(min(j + 1, NR - 1), nl, nr_);          // next row in flight (last row: reloaded)
#pragma unroll
        for (int k = 0; k < POOL; ++k) hc[k] = lerp4(cl[k], cr[k], xfr[k]);
        drain(j, hc, hp, xtag);
    };
    auto run = [&](auto xtag) {
        sy = 0;
        drain(-1, hA, hB, xtag);                         // leading extrapolated sample rows
        if (NR > 0) load_row(0, lAl, lAr);
#pragma unroll 1
        for (int j = 0; j < NR; j += 2) {
            step(j, hA, hB, lAl, lAr, lBl, lBr, xtag);
            if (j + 1 < NR) step(j + 1, hB, hA, lBl, lBr, lAl, lAr, xtag);
        }
        drain(NR, hA, hB, xtag);                         // trailing extrapolated sample rows
    };
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub, sub_off += RW_CS * 4) {
        if (wave_xall) run(BoolTag<true>{}); else run(BoolTag<false>{});
    }
}

// ---- any-stride fallback: one thread per channel.
template <int POOL>
__global__ void __launch_bounds__(128)
roi_fwd_generic_kernel(const float* __restrict__ fmap, int H, int W, int C, long long sb,
                       long long sh, long long sw, long long sc, const float4* __restrict__ rois,
                       int R, int ch, int cw, float* __restrict__ out,
                       uint8_t* __restrict__ winner) {
    __shared__ AxisGeom gy, gx;
    const int r = blockIdx.x, b = blockIdx.y;
    const float4 box = rois[(size_t)b * R + r];
    fill_geom(gy, gx, box, H, W, ch, cw);
    __syncthreads();
    const int ph = ch / POOL, pw = cw / POOL;
    const float* img = fmap + (size_t)b * sb;
    const size_t obase = ((size_t)b * R + r) * (size_t)(ph * pw) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float* chan = img + (size_t)c * sc;
        for (int bin = 0; bin < ph * pw; ++bin) {
            const int p = bin / pw, q = bin % pw;
            float best = 0.f;
            uint8_t arg = 0;
            for (int k = 0; k < POOL * POOL; ++k) {
                const int sy = p * POOL + k / POOL, sx = q * POOL + k % POOL;
                float v = 0.f;
                if (gy.ok[sy] && gx.ok[sx]) {
                    const float* rt = chan + (size_t)gy.lo[sy] * sh;
                    const float* rb = chan + (size_t)gy.hi[sy] * sh;
                    const size_t ol = (size_t)gx.lo[sx] * sw, orr = (size_t)gx.hi[sx] * sw;
                    const float tl = rt[ol], tr = rt[orr], bl = rb[ol], br = rb[orr];
                    const float lx = gx.frac[sx], ly = gy.frac[sy];
                    const float top = tl + (tr - tl) * lx;
                    const float bot = bl + (br - bl) * lx;
                    v = top + (bot - top) * ly;
                }
                if (k == 0 || v > best) { best = v; arg = (uint8_t)k; }
            }
            const size_t o = obase + (size_t)bin * C + c;
            out[o] = best;
            if (winner) winner[o] = arg;
        }
    }
}

// ---- backward: gather formulation.  Workgroup = one feature cell (b,y,x);
// thread = float4 of channels.  Every RoI of the image is visited in order;
// for each, the (at most a few) samples whose bilinear footprint touches the
// cell are found from the axis geometry, and the winner byte decides whether
// the bin's gradient flows through that sample.  Summation order per cell:
// RoI, bin (row-major), corner TL,TR,BL,BR.
template <int POOL>
__global__ void __launch_bounds__(128)
roi_bwd_cl_kernel(const float* __restrict__ dY, const uint8_t* __restrict__ winner,
                  const float4* __restrict__ rois, int R, int H, int W, int C, int ch, int cw,
                  float* __restrict__ dfmap) {
    __shared__ int s_sy[ROI_MAX_CROP], s_sx[ROI_MAX_CROP];
    __shared__ int s_ny, s_nx;
    const int x = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int ph = ch / POOL, pw = cw / POOL;
    const int C4 = C >> 2;
    // per-thread accumulators: this kernel supports C4 <= 4 * blockDim
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < R; ++r) {
        const float4 box = rois[(size_t)b * R + r];
        __syncthreads();
        // sample rows / cols of this RoI that touch (y, x): first wave scans rows,
        // second wave scans cols, ordered compaction by ballot.
        {
            const int t = threadIdx.x;
            if (t < 64) {
                int lo = 0, hi = 0, ok = 0; float fr = 0.f;
                if (t < ch) axis_sample(box.x, box.z, H, ch, t, lo, hi, fr, ok);
                const bool hit = t < ch && ok && (lo == y || hi == y);
                const unsigned long long m = __ballot(hit);
                if (hit) s_sy[__popcll(m & lanemask_lt())] = t;
                if (t == 0) s_ny = __popcll(m);
            } else {
                const int s = t - 64;
                int lo = 0, hi = 0, ok = 0; float fr = 0.f;
                if (s < cw) axis_sample(box.y, box.w, W, cw, s, lo, hi, fr, ok);
                const bool hit = s < cw && ok && (lo == x || hi == x);
                const unsigned long long m = __ballot(hit);
                if (hit) s_sx[__popcll(m & lanemask_lt())] = s;
                if (s == 0) s_nx = __popcll(m);
            }
        }
        __syncthreads();
        const int ny = s_ny, nx = s_nx;
        if (ny == 0 || nx == 0) continue;                 // block-uniform
        const size_t rbase = ((size_t)b * R + r) * (size_t)(ph * pw) * C;
        for (int a = 0; a < ny; ++a) {
            const int sy = s_sy[a];
            int ylo, yhi, yok; float ly;
            axis_sample(box.x, box.z, H, ch, sy, ylo, yhi, ly, yok);
            for (int e = 0; e < nx; ++e) {
                const int sx = s_sx[e];
                int xlo, xhi, xok; float lx;
                axis_sample(box.y, box.w, W, cw, sx, xlo, xhi, lx, xok);
                const int p = sy / POOL, q = sx / POOL;
                const int k = (sy % POOL) * POOL + (sx % POOL);
                const size_t o = rbase + (size_t)(p * pw + q) * C;
                const float wt = 1.0f - ly, wb = ly, wl = 1.0f - lx, wr = lx;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c4 = threadIdx.x + u * 128;
                    if (c4 >= C4) break;
                    uchar4 wv = make_uchar4(0, 0, 0, 0);
                    if (POOL > 1) wv = *reinterpret_cast<const uchar4*>(winner + o + c4 * 4);
                    const float4 g = *reinterpret_cast<const float4*>(dY + o + c4 * 4);
                    const bool mx = wv.x == k, my = wv.y == k, mz = wv.z == k, mw = wv.w == k;
                    float4 dtop, dbot;
                    dtop.x = wt * g.x; dtop.y = wt * g.y; dtop.z = wt * g.z; dtop.w = wt * g.w;
                    dbot.x = wb * g.x; dbot.y = wb * g.y; dbot.z = wb * g.z; dbot.w = wb * g.w;
                    // corners in the order TL, TR, BL, BR; a cell can be several of them
                    if (ylo == y && xlo == x) {
                        if (mx) acc[u].x += wl * dtop.x; if (my) acc[u].y += wl * dtop.y;
                        if (mz) acc[u].z += wl * dtop.z; if (mw) acc[u].w += wl * dtop.w;
                    }
                    if (ylo == y && xhi == x) {
                        if (mx) acc[u].x += wr * dtop.x; if (my) acc[u].y += wr * dtop.y;
                        if (mz) acc[u].z += wr * dtop.z; if (mw) acc[u].w += wr * dtop.w;
                    }
                    if (yhi == y && xlo == x) {
                        if (mx) acc[u].x += wl * dbot.x; if (my) acc[u].y += wl * dbot.y;
                        if (mz) acc[u].z += wl * dbot.z; if (mw) acc[u].w += wl * dbot.w;
                    }
                    if (yhi == y && xhi == x) {
                        if (mx) acc[u].x += wr * dbot.x; if (my) acc[u].y += wr * dbot.y;
                        if (mz) acc[u].z += wr * dbot.z; if (mw) acc[u].w += wr * dbot.w;
                    }
                }
            }
        }
    }
    float* dst = dfmap + (((size_t)b * H + y) * W + x) * C;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c4 = threadIdx.x + u * 128;
        if (c4 < C4) *reinterpret_cast<float4*>(dst + c4 * 4) = acc[u];
    }
}

// ---- backward, fast path: one wave per feature cell, no LDS, no barrier.
// Lane = RoI while candidates are detected (64 at a time, conservative reach
// test, ballot), then lane = channel quad while a candidate is applied.  For a
// candidate the wave evaluates all sample rows (lanes 0-31) and sample columns
// (lanes 32-63) at once; ballots give the sample rows / columns whose bilinear
// footprint contains this cell, and the (few) hits are walked by scalar loops.
// Summation order per cell: RoI ascending, sample (sy,sx) row-major, corner
// TL,TR,BL,BR.
__device__ __forceinline__ bool axis_reaches(float c1, float c2, int dim, int n, int target) {
    float first, last;
    if (n > 1) {
        const float scale = ((c2 - c1) * (float)(dim - 1)) / (float)(n - 1);
        first = c1 * (float)(dim - 1);
        last = c1 * (float)(dim - 1) + (float)(n - 1) * scale;
    } else {
        first = last = (float)(0.5 * (double)(c1 + c2) * (double)(dim - 1));
    }
    const float lo = floorf(fminf(first, last)), hi = ceilf(fmaxf(first, last));
    return (float)target >= lo && (float)target <= hi;      // NaN -> false
}

template <int POOL, int NQ>
__global__ void __launch_bounds__(64)
roi_bwd_wave_kernel(const float* __restrict__ dY, const uint8_t* __restrict__ winner,
                    const float4* __restrict__ rois, int R, int H, int W, int C, int ch, int cw,
                    float* __restrict__ dfmap) {
    const int x = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x;
    const int ph = ch / POOL, pw = cw / POOL;
    const int C4 = C >> 2;
    float4 acc[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* rb = rois + (size_t)b * R;
    const bool isx = lane >= 32;
    const int s = lane & 31;
    const int target = isx ? x : y;
    const int ns = isx ? cw : ch, dim = isx ? W : H;
#pragma unroll 1
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        bool cand = false;
        if (r < R) {
            box = rb[r];
            cand = axis_reaches(box.x, box.z, H, ch, y) && axis_reaches(box.y, box.w, W, cw, x);
        }
        unsigned long long m = __ballot(cand);
#pragma unroll 1
        while (m != 0ull) {                                   // wave-uniform
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const float by1 = rlf(box.x, l), bx1 = rlf(box.y, l);
            const float by2 = rlf(box.z, l), bx2 = rlf(box.w, l);
            int lo = 0, hi = 0, ok = 0;
            float fr = 0.f;
            if (s < ns) axis_sample(isx ? bx1 : by1, isx ? bx2 : by2, dim, ns, s, lo, hi, fr, ok);
            const bool v = s < ns && ok;
            const unsigned long long mlo = __ballot(v && lo == target);
            const unsigned long long mhi = __ballot(v && hi == target);
            const uint32_t ylo_m = (uint32_t)mlo, yhi_m = (uint32_t)mhi;
            const uint32_t xlo_m = (uint32_t)(mlo >> 32), xhi_m = (uint32_t)(mhi >> 32);
            uint32_t ym = ylo_m | yhi_m;
            const uint32_t xm_all = xlo_m | xhi_m;
            if (ym == 0u || xm_all == 0u) continue;
            const size_t rbase = ((size_t)b * R + (r0 + l)) * (size_t)(ph * pw) * C;
            while (ym != 0u) {
                const int sy = __ffs((int)ym) - 1;
                ym &= ym - 1u;
                const float ly = rlf(fr, sy);
                const bool top = (ylo_m >> sy) & 1u, bot = (yhi_m >> sy) & 1u;
                const float wt = 1.0f - ly, wb = ly;
                uint32_t xm = xm_all;
                while (xm != 0u) {
                    const int sx = __ffs((int)xm) - 1;
                    xm &= xm - 1u;
                    const float lx = rlf(fr, 32 + sx);
                    const bool left = (xlo_m >> sx) & 1u, right = (xhi_m >> sx) & 1u;
                    const float wl = 1.0f - lx, wr = lx;
                    const uint32_t k = (uint32_t)((sy % POOL) * POOL + (sx % POOL));
                    const size_t o = rbase + (size_t)((sy / POOL) * pw + (sx / POOL)) * C;
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        const int c4 = lane + 64 * u;
                        if (c4 >= C4) break;
                        float4 g = *reinterpret_cast<const float4*>(dY + o + c4 * 4);
                        if (POOL > 1) {
                            const uint32_t wv = *reinterpret_cast<const uint32_t*>(winner + o + c4 * 4);
                            if ((wv & 0xFFu) != k) g.x = 0.f;
                            if (((wv >> 8) & 0xFFu) != k) g.y = 0.f;
                            if (((wv >> 16) & 0xFFu) != k) g.z = 0.f;
                            if ((wv >> 24) != k) g.w = 0.f;
                        }
                        // a non-winner channel adds +0 (x + 0 == x): same value as skipping it
                        if (top) {
                            const float4 d = make_float4(wt * g.x, wt * g.y, wt * g.z, wt * g.w);
                            if (left) {
                                acc[u].x += wl * d.x; acc[u].y += wl * d.y;
                                acc[u].z += wl * d.z; acc[u].w += wl * d.w;
                            }
                            if (right) {
                                acc[u].x += wr * d.x; acc[u].y += wr * d.y;
                                acc[u].z += wr * d.z; acc[u].w += wr * d.w;
                            }
                        }
                        if (bot) {
                            const float4 d = make_float4(wb * g.x, wb * g.y, wb * g.z, wb * g.w);
                            if (left) {
                                acc[u].x += wl * d.x; acc[u].y += wl * d.y;
                                acc[u].z += wl * d.z; acc[u].w += wl * d.w;
                            }
                            if (right) {
                                acc[u].x += wr * d.x; acc[u].y += wr * d.y;
                                acc[u].z += wr * d.z; acc[u].w += wr * d.w;
                            }
                        }
                    }
                }
            }
        }
    }
    float* dst = dfmap + (((size_t)b * H + y) * W + x) * C;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int c4 = lane + 64 * u;
        if (c4 < C4) *reinterpret_cast<float4*>(dst + c4 * 4) = acc[u];
    }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_roi_crop_pool_fwd(const float* fmap, int B, int H, int W, int C, int64_t sb,
                                       int64_t sh, int64_t sw, int64_t sc, const float* rois,
                                       int R, int crop_h, int crop_w, int pool, float* out,
                                       uint8_t* winner, frcnn_stream_t stream) {
    if (!fmap || !rois || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || R <= 0 ||
        crop_h <= 0 || crop_w <= 0)
        return FRCNN_E_BADARG;
    if (pool != 1 && pool != 2) return FRCNN_E_BADARG;
    if (crop_h % pool || crop_w % pool) return FRCNN_E_BADARG;
    if (crop_h > ROI_MAX_CROP || crop_w > ROI_MAX_CROP || B > 65535) return FRCNN_E_UNSUPPORTED;
    const bool fast = sc == 1 && (C % 4) == 0 && (sb % 4) == 0 && (sh % 4) == 0 && (sw % 4) == 0 &&
                      (reinterpret_cast<uintptr_t>(fmap) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) % 16) == 0 &&
                      (!winner || (reinterpret_cast<uintptr_t>(winner) % 4) == 0);
    const dim3 grid(R, B), block(128);
    const float4* rb = reinterpret_cast<const float4*>(rois);
    hipStream_t st = (hipStream_t)stream;
    const bool walk_path = fast && (C % RW_CS) == 0 && crop_h <= RW_MAXCROP &&
                           crop_w <= RW_MAXCROP && (long long)R * (C / RW_CS) <= 0x7FFFFFFFll &&
                           sb > 0 && sh > 0 && sw > 0 && (long long)(W - 1) * sw + C <= sh &&
                           (long long)H * sh * 4 < 0x7FFFFFFFll &&
                           (long long)W * sw * 4 + (long long)C * 4 < 0x7FFFFFFFll &&
                           (long long)(crop_h / pool) * (crop_w / pool) * C * 4 < 0x7FFFFFFFll;
    if (walk_path) {
        // 64-channel sub-slices walked by one workgroup.  2, 4 and 8 measured no faster than 1
        // (the kernel is bound by the texture addresser, not by the per-RoI set-up).
        const int nsub = 1;
        const int nslice = C / (RW_CS * nsub);
        const int pw = crop_w / pool;
        const dim3 g3((unsigned)(R * nslice), B), b3(64 * ((pw + 3) / 4));
#define FRCNN_WALK(P, WN)                                                                       \
    FRCNN_LAUNCH((roi_fwd_walk_kernel<P, WN>), g3, b3, 0, st, fmap, H, W, C, (long long)sb,     \
                 (long long)sh, (long long)sw, rb, R, crop_h, crop_w, nslice, nsub, out, winner)
        if (pool == 2) { if (winner) FRCNN_WALK(2, true); else FRCNN_WALK(2, false); }
        else           { FRCNN_WALK(1, false); }
#undef FRCNN_WALK
    } else if (fast) {
        if (pool == 2)
            FRCNN_LAUNCH(roi_fwd_cl_kernel<2>, grid, block, 0, st, fmap, H, W, C,
                               (long long)sb, (long long)sh, (long long)sw, rb, R, crop_h, crop_w,
                               out, winner);
        else
            FRCNN_LAUNCH(roi_fwd_cl_kernel<1>, grid, block, 0, st, fmap, H, W, C,
                               (long long)sb, (long long)sh, (long long)sw, rb, R, crop_h, crop_w,
                               out, winner);
    } else {
        if (pool == 2)
            FRCNN_LAUNCH(roi_fwd_generic_kernel<2>, grid, block, 0, st, fmap, H, W, C,
                               (long long)sb, (long long)sh, (long long)sw, (long long)sc, rb, R,
                               crop_h, crop_w, out, winner);
        else
            FRCNN_LAUNCH(roi_fwd_generic_kernel<1>, grid, block, 0, st, fmap, H, W, C,
                               (long long)sb, (long long)sh, (long long)sw, (long long)sc, rb, R,
                               crop_h, crop_w, out, winner);
    }
    return FRCNN_OK;
}

extern "C" int frcnn_roi_crop_pool_bwd(const float* dY, const uint8_t* winner, const float* rois,
                                       int B, int R, int H, int W, int C, int crop_h, int crop_w,
                                       int pool, float* dfmap, frcnn_stream_t stream) {
    if (!dY || !rois || !dfmap || B <= 0 || R <= 0 || H <= 0 || W <= 0 || C <= 0 || crop_h <= 0 ||
        crop_w <= 0)
        return FRCNN_E_BADARG;
    if (pool != 1 && pool != 2) return FRCNN_E_BADARG;
    if (pool == 2 && !winner) return FRCNN_E_BADARG;
    if (crop_h % pool || crop_w % pool) return FRCNN_E_BADARG;
    if (crop_h > ROI_MAX_CROP || crop_w > ROI_MAX_CROP) return FRCNN_E_UNSUPPORTED;
    if ((C % 4) != 0 || C > 4 * 4 * 128 || H > 65535 || B > 65535) return FRCNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dY) % 16) || (reinterpret_cast<uintptr_t>(dfmap) % 16) ||
        (winner && (reinterpret_cast<uintptr_t>(winner) % 4)))
        return FRCNN_E_BADARG;
    const dim3 grid(W, H, B), block(128);
    const float4* rb = reinterpret_cast<const float4*>(rois);
    hipStream_t st = (hipStream_t)stream;
    if (crop_h <= 32 && crop_w <= 32) {
        const int nq = (C / 4 + 63) / 64;                 // channel quads per lane, <= 8
#define FRCNN_BWD(P, Q)                                                                          \
    FRCNN_LAUNCH((roi_bwd_wave_kernel<P, Q>), grid, dim3(64), 0, st, dY, winner, rb, R, H, W, C,  \
                 crop_h, crop_w, dfmap)
#define FRCNN_BWD_P(P)                                                                           \
    do {                                                                                         \
        if (nq <= 1) FRCNN_BWD(P, 1); else if (nq <= 2) FRCNN_BWD(P, 2);                          \
        else if (nq <= 4) FRCNN_BWD(P, 4); else FRCNN_BWD(P, 8);                                  \
    } while (0)
        if (pool == 2) FRCNN_BWD_P(2); else FRCNN_BWD_P(1);
#undef FRCNN_BWD_P
#undef FRCNN_BWD
        return FRCNN_OK;
    }
    if (pool == 2)
        FRCNN_LAUNCH(roi_bwd_cl_kernel<2>, grid, block, 0, st, dY, winner, rb, R, H, W, C,
                           crop_h, crop_w, dfmap);
    else
        FRCNN_LAUNCH(roi_bwd_cl_kernel<1>, grid, block, 0, st, dY, winner, rb, R, H, W, C,
                           crop_h, crop_w, dfmap);
    return FRCNN_OK;
}
