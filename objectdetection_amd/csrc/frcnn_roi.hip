// Fused bilinear RoI crop + max-pool (forward) and its atomic-free adjoint.
// Sampling rule and op order of tf.image.crop_and_resize's CPU kernel:
//   hs = ((y2-y1)*(H-1))/(ch-1);  in_y = y1*(H-1) + y*hs
//   rows/cols whose source coordinate is outside [0, dim-1] produce 0
//   top = TL + (TR-TL)*lx; bottom = BL + (BR-BL)*lx; out = top + (bottom-top)*ly
// followed by the pool x pool, stride pool maximum (first maximum wins).
// gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"
#include <cstdlib>

namespace frcnn {

constexpr int ROI_MAX_CROP = 64;

struct AxisGeom {
    int lo[ROI_MAX_CROP];      // floor index
    int hi[ROI_MAX_CROP];      // ceil index
    float frac[ROI_MAX_CROP];  // lerp weight
    int ok[ROI_MAX_CROP];      // 0 -> extrapolated (output 0)
};

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

// ---- channels-last "staged row walker" (C % 64 == 0, crop <= 16x16): shared pieces.
// One wave = one RoI x a few 64-channel slices; thread = (channel quad, pooled column);
// no barrier: every wave is self-contained.
//
// The sample rows of a box touch an ascending list of feature rows.  A wave walks that
// list once: per feature row it brings the cells its sample columns need, lerps them
// horizontally, keeps the last two lerped rows in registers; every sample row whose
// bottom row has arrived is then finished with one vertical lerp per sample column and
// folded into the running max of its pool window.  A feature row shared by several sample
// rows (small boxes) is fetched once.
//
// The row list and the per-sample-row entries live one entry per lane and are read with
// v_readlane by the uniform loop counters, so the loop control runs on the scalar unit.
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    const f32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
// 16-byte buffer store with a scalar row offset.  gfx9-family hardware needs a wait state
// between a vector-memory store of more than 64 bits and a vector instruction that overwrites
// the store's data registers.  hipcc (ROCm 7.2) inserts it only when the store's soffset is NOT
// a register (GCNHazardRecognizer::createsVALUHazard exempts the register form), but gfx950
// needs it with a scalar-register offset as well: the compiler-scheduled store below
// (FRCNN_STORE_VARIANT 1) is followed directly by `v_pk_mul_f32 v[10:11], ...` on its own data
// registers v[10:13], and the x / y components of some lanes reach memory with the NEXT row's
// values.  Measured with the stress test per variant: compiler store + scalar offset FAILS,
// compiler store with the offset added into the VGPR (the compiler then pads) passes, asm
// store without a wait state FAILS, with one (s_nop 0) or two (s_nop 1) wait states passes.
// The store is issued from asm together with two wait states (what the compiler uses for the
// non-register form on gfx940 and later), so nothing can be scheduled in between.
#ifndef FRCNN_STORE_VARIANT
#define FRCNN_STORE_VARIANT 0    // experiments on the store hazard: 1..5, see buf_store4
#endif
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff,
                                           float4 v) {
    f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
#if FRCNN_STORE_VARIANT == 1        // compiler-scheduled store, scalar row offset
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, soff, 0);
#elif FRCNN_STORE_VARIANT == 2      // compiler-scheduled store, row offset added into the VGPR
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff + soff, 0, 0);
#elif FRCNN_STORE_VARIANT == 3      // asm store, no wait state behind it
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen"
                 :: "v"(t), "v"(voff), "s"(rs), "s"(soff) : "memory");
#elif FRCNN_STORE_VARIANT == 4      // one wait state
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 0"
                 :: "v"(t), "v"(voff), "s"(rs), "s"(soff) : "memory");
#elif FRCNN_STORE_VARIANT == 5      // two wait states
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1"
                 :: "v"(t), "v"(voff), "s"(rs), "s"(soff) : "memory");
#else
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1"
                 :: "v"(t), "v"(voff), "s"(rs), "s"(soff) : "memory");
#endif
}
template <bool B> struct BoolTag { static constexpr bool value = B; };
template <int I> struct IntTag { static constexpr int value = I; };

// ---- the staging.
// A 16-byte vector load costs the texture path about 20 cycles per wave instruction
// whatever its lanes fetch (masked-off lanes are not cheaper), an LDS read of the same
// width about 5.  So a wave stages the DISTINCT cells that its sample columns touch in a
// feature row -- at most 16 cells, four per direct-to-LDS load (buffer_load ... lds: lane l
// brings channel quad l%16 of cell l/16, the wave's 1 KiB lands contiguously) -- into a
// private ring of row slots, one row ahead, and the threads read their cells from there.
// A slot is read only by the wave that filled it, behind a counted vmcnt wait.  Pooled
// columns are taken four at a time (one per 16-lane group); two such groups are walked
// together (two pooled columns per thread) when their cells fit a slot, which halves the
// scalar work per output.  Boxes with a sample outside the image take a general walk
// (zero rows / columns), flipped boxes a direct one; everything else the fast walk.
constexpr int ST_CELLS = 16;
constexpr int ST_SLOT_BYTES = ST_CELLS * RW_CS * 4;       // 4 KiB
#ifndef FRCNN_ST_NSLOT
#define FRCNN_ST_NSLOT 2         // row slots of the fast walk's ring (rows in flight + 1)
#endif
constexpr int ST_NSLOT = FRCNN_ST_NSLOT;
#ifndef FRCNN_ST_PAIR
#define FRCNN_ST_PAIR 1          // walk two groups of pooled columns together when they fit
#endif
#ifndef FRCNN_ST_NSUB
#define FRCNN_ST_NSUB 2          // consecutive 64-channel slices walked by one wave
#endif
#ifndef FRCNN_ST_WPE
#define FRCNN_ST_WPE 4           // waves per SIMD the staged kernel is compiled for
#endif

// distinct values of the (floor, ceil) indices of samples s0 .. s0+ns-1 of one axis
// (ns <= 32): lane k: position of sample s0+k's floor / ceil in the list (-1:
// extrapolated); lane u: the list's u-th value; count.  Ascending coordinates only.
struct AxisUnique { int t_lo, t_hi, t_val, n; float t_fr; };

__device__ __forceinline__ AxisUnique axis_unique(float c1, float c2, int dim, int n, int s0, int ns) {
    const int lane = lane_id();
    const int k = lane >> 1, which = lane & 1;
    int lo = 0, hi = 0, ok = 0;
    float fr = 0.f;
    const bool in_range = k < ns && s0 + k < n;
    if (in_range) axis_sample(c1, c2, dim, n, s0 + k, lo, hi, fr, ok);
    const bool valid = in_range && ok;
    const int v = valid ? (which ? hi : lo) : -1 - lane;
    const int v1 = __shfl_up(v, 1), v2 = __shfl_up(v, 2);
    const bool is_first = valid && (lane < 1 || v != v1) && (lane < 2 || v != v2);
    const unsigned long long fm = __ballot(is_first);
    const unsigned long long le = fm & (lanemask_lt() | (1ull << lane));
    const int cnt = __popcll(le);
    const int F = le ? 63 - __clzll((long long)le) : 0;
    const int vF = __shfl(v, F);
    const int idx = (v == vF) ? cnt - 1 : cnt - 2;
    const int src = (2 * lane) & 63;
    const int s_ok = __shfl((int)valid, src);
    AxisUnique u;
    u.t_lo = __shfl(idx, src);
    u.t_hi = __shfl(idx, src | 1);
    u.t_fr = __shfl(fr, src);
    if (!(lane < ns && s_ok)) { u.t_lo = -1; u.t_hi = -1; }
    u.t_val = __builtin_amdgcn_ds_permute((is_first ? cnt - 1 : 63) << 2, v);
    u.n = __popcll(fm);
    return u;
}

// wait until at most n (uniform) of this wave's vector memory operations are outstanding.
// Loads and stores of a wave retire in issue order on this counter (the compiler's own waits
// rely on the same), so "at most n outstanding" means everything but the last n issued is done;
// callers may under-count n (waits longer), never over-count;
// n above 8 waits as for 8
__device__ __forceinline__ void wait_vm_at_most(int n) {
    if (n < 4) {
        if (n < 2) {
            if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
            if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
    } else if (n < 6) {
        if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
        if (n == 6)      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else             asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
}

// The same wait with two outcomes instead of nine: nothing issued since the staged row ->
// vmcnt(0); anything -> vmcnt(k), k (compile time, <= 8) = the stores of ONE finished output
// row, the fewest a non-zero count can be.  Under-counting only waits for a few stores more;
// the nine-way branch tree cost some twenty scalar instructions per feature row.
template <int K>
__device__ __forceinline__ void wait_vm_zero_or(int n) {
    static_assert(K >= 1 && K <= 8, "vmcnt immediate");
    if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// np (1..4, uniform) direct-to-LDS loads: piece p's lane l brings the 16 bytes at
// (rsrc base + voff_p + soff) to LDS byte lds_addr + 1024 p + 16 l.  Issued from asm so that
// the compiler neither counts them nor orders LDS reads behind them: the caller waits with
// wait_vm_at_most() before it reads the slot.
__device__ __forceinline__ void lds_dma_pieces(i32x4 rsrc, uint32_t v0, uint32_t v1, uint32_t v2,
                                               uint32_t v3, uint32_t soff, uint32_t lds_addr, int np) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_cmp_lt_u32 %8, 2\n\t"
        "buffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_cmp_lt_u32 %8, 3\n\t"
        "buffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_cmp_lt_u32 %8, 4\n\t"
        "buffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %5, %6 offen lds\n"
        "1:\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rsrc), "s"(soff), "s"(lds_addr), "s"(np)
        : "memory", "scc");
}

struct StageCtx {
    RowWalk rw;
    float4 box;
    int W, C, cw, ph, pw, nsy, nsub;
    uint32_t row_bytes, sw_bytes, chan_off, orow_bytes, ring_addr;
    i32x4 img_dma;
    __amdgpu_buffer_rsrc_t obuf;
    unsigned char* ring;
    uint8_t* winner_roi;       // winner + this RoI's base, or nullptr
};

// Walks the rows for NQ groups of four pooled columns starting at group g0; this thread's
// pooled columns are (g0+t)*4 + lane/16, t < NQ.  ux covers the groups' sample columns.
template <int POOL, bool WIN, int NQ>
__device__ __forceinline__ void staged_walk(const StageCtx& c, const AxisUnique& ux, int g0) {
    const int lane = lane_id();
    const int cq = lane & (RW_LANES - 1), ql = lane / RW_LANES;
    constexpr int NS = NQ * POOL;                  // sample columns per thread
    uint32_t xlo[NS], xhi[NS];
    float xfr[NS];
    bool xok[NS];
    bool xall = true;
    bool q_live[NQ];
    uint32_t ooff[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int q = (g0 + t) * 4 + ql;
        q_live[t] = q < c.pw;
        ooff[t] = (uint32_t)q * (uint32_t)c.C * 4u + c.chan_off;
#pragma unroll
        for (int k = 0; k < POOL; ++k) {
            int lo = 0, hi = 0, ok = 0;
            float fr = 0.f;
            const int sx = q * POOL + k;
            if (sx < c.cw) axis_sample(c.box.y, c.box.w, c.W, c.cw, sx, lo, hi, fr, ok);
            const int e = t * POOL + k;
            xfr[e] = fr;
            xok[e] = sx < c.cw && ok;
            xall = xall && (xok[e] || sx >= c.cw);
            const int tl = (t * 4 + ql) * POOL + k;          // sample's place in ux's tables
            const int li = __shfl(ux.t_lo, tl), ri = __shfl(ux.t_hi, tl);
            xlo[e] = (uint32_t)max(li, 0) * (RW_CS * 4u) + (uint32_t)cq * 16u;
            xhi[e] = (uint32_t)max(ri, 0) * (RW_CS * 4u) + (uint32_t)cq * 16u;
        }
    }
    const bool wave_xall = __all(xall);
    const int ncell = ux.n;                                               // uniform, <= ST_CELLS
    const int npiece = (ncell + 3) >> 2;
    uint32_t voff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int u = min(4 * p + ql, max(ncell - 1, 0));
        voff[p] = (uint32_t)__shfl(ux.t_val, u) * c.sw_bytes + c.chan_off;
    }
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const RowWalk& rw = c.rw;
    const int NR = rw.nr, nsy = c.nsy;
    float4 best[NQ];
    uchar4 arg[NQ];
    float4 hA[NS], hB[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) hA[e] = hB[e] = zero;
#pragma unroll
    for (int t = 0; t < NQ; ++t) { best[t] = zero; arg[t] = make_uchar4(0, 0, 0, 0); }
    unsigned sy = 0;
    int nst = 0;                 // vector memory operations issued after the staged row in flight

    auto fold = [&](unsigned s, const float4 (&v)[NS]) {
        const unsigned ky = s % POOL;
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const float4* w = &v[t * POOL];
            if (WIN) {
#pragma unroll
                for (int kx = 0; kx < POOL; ++kx) {
                    const int k = ky * POOL + kx;
                    if (k == 0) {
                        best[t] = w[kx];
                        arg[t] = make_uchar4(0, 0, 0, 0);
                    } else {
                        if (w[kx].x > best[t].x) { best[t].x = w[kx].x; arg[t].x = k; }
                        if (w[kx].y > best[t].y) { best[t].y = w[kx].y; arg[t].y = k; }
                        if (w[kx].z > best[t].z) { best[t].z = w[kx].z; arg[t].z = k; }
                        if (w[kx].w > best[t].w) { best[t].w = w[kx].w; arg[t].w = k; }
                    }
                }
            } else if (POOL == 1) {
                best[t] = w[0];
            } else if (ky == 0) {
                best[t].x = vmax2(w[0].x, w[POOL - 1].x); best[t].y = vmax2(w[0].y, w[POOL - 1].y);
                best[t].z = vmax2(w[0].z, w[POOL - 1].z); best[t].w = vmax2(w[0].w, w[POOL - 1].w);
            } else {
                best[t].x = vmax3(best[t].x, w[0].x, w[POOL - 1].x);
                best[t].y = vmax3(best[t].y, w[0].y, w[POOL - 1].y);
                best[t].z = vmax3(best[t].z, w[0].z, w[POOL - 1].z);
                best[t].w = vmax3(best[t].w, w[0].w, w[POOL - 1].w);
            }
        }
        if (ky == POOL - 1) {
            const uint32_t orow = (s / POOL) * c.orow_bytes;                           // uniform
            nst = __builtin_amdgcn_readfirstlane(nst + NQ * (WIN ? 2 : 1));
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                if (q_live[t]) {
                    buf_store4(c.obuf, ooff[t], orow, best[t]);
                    if (WIN)
                        *reinterpret_cast<uchar4*>(c.winner_roi + ((orow + ooff[t]) >> 2)) = arg[t];
                }
            }
        }
    };
    auto stage_row = [&](int j, int slot) {
        if (npiece > 0) {
            const uint32_t roff = (uint32_t)rl(rw.t_row, j) * c.row_bytes;             // uniform
            lds_dma_pieces(c.img_dma, voff[0], voff[1], voff[2], voff[3], roff,
                           c.ring_addr + (uint32_t)slot * ST_SLOT_BYTES, npiece);
        }
        nst = 0;
    };
    // finish every sample row that needs nothing beyond unique row j
    auto drain = [&](int j, const float4 (&hc)[NS], const float4 (&hp)[NS], auto xtag) {
        constexpr bool XALL = decltype(xtag)::value;
        while (sy < (unsigned)nsy) {
            const int hi = rl(rw.t_hi, sy);               // -1: extrapolated row
            if (hi > j) break;
            const int lo = rl(rw.t_lo, sy);
            const float fr = rlf(rw.t_fr, sy);
            float4 v[NS];
            if (lo < 0) {
#pragma unroll
                for (int e = 0; e < NS; ++e) v[e] = zero;
            } else if (lo == j) {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    v[e] = lerp4(hc[e], hc[e], fr);
                    if (!XALL && !xok[e]) v[e] = zero;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    v[e] = lerp4(hp[e], hc[e], fr);
                    if (!XALL && !xok[e]) v[e] = zero;
                }
            }
            fold(sy, v);
            ++sy;
        }
    };
    auto step = [&](int j, int slot, float4 (&hc)[NS], const float4 (&hp)[NS], auto xtag) {
        wait_vm_at_most(nst);                              // row j has landed in its slot
        const unsigned char* sp = c.ring + slot * ST_SLOT_BYTES;
        float4 cl[NS], cr[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            cl[e] = *reinterpret_cast<const float4*>(sp + xlo[e]);
            cr[e] = *reinterpret_cast<const float4*>(sp + xhi[e]);
        }
        asm volatile("" ::: "memory");
        if (j + 1 < NR) stage_row(j + 1, slot ^ 1);        // the slot read one step ago
#pragma unroll
        for (int e = 0; e < NS; ++e) hc[e] = lerp4(cl[e], cr[e], xfr[e]);
        drain(j, hc, hp, xtag);
    };
    auto run = [&](auto xtag) {
        drain(-1, hA, hB, xtag);                           // leading extrapolated sample rows
        if (NR > 0) stage_row(0, 0);
#pragma unroll 1
        for (int j = 0; j < NR; j += 2) {
            step(j, 0, hA, hB, xtag);
            if (j + 1 < NR) step(j + 1, 1, hB, hA, xtag);
        }
        drain(NR, hA, hB, xtag);                           // trailing extrapolated sample rows
    };
#pragma unroll 1
    for (int sub = 0; sub < c.nsub; ++sub) {
        if (wave_xall) run(BoolTag<true>{}); else run(BoolTag<false>{});
#pragma unroll
        for (int p = 0; p < 4; ++p) voff[p] += RW_CS * 4;
#pragma unroll
        for (int t = 0; t < NQ; ++t) ooff[t] += RW_CS * 4;
        sy = 0;
    }
}

// The row list of the fast walk: every sample row's (floor, ceil) rows sit at CONSECUTIVE
// places p, p+1 of the list, so a sample row is always "previous lerped row, current lerped
// row".  A sample that falls exactly on a feature row lists that row twice (the walk then
// visits it twice and forms top + (bottom - top) * w from two equal rows, as the definition
// does).  lane s: place of sample row s's ceil row and its weight; lane j: feature row at
// place j; nr places.  Only meaningful when every sample row is inside the image.
struct RowDup { int t_hi, t_row, nr; float t_fr; bool all_valid; };

__device__ __forceinline__ RowDup build_row_dup(float y1, float y2, int H, int ch) {
    const int lane = lane_id();
    const int s = lane >> 1, which = lane & 1;
    int lo = 0, hi = 0, ok = 0;
    float fr = 0.f;
    if (s < ch) axis_sample(y1, y2, H, ch, s, lo, hi, fr, ok);
    const bool in_list = s < ch;
    const int plo = __shfl_up(lo, 2), phi = __shfl_up(hi, 2);
    const bool pair_eq = s > 0 && lo == plo && hi == phi;
    const bool shift = s > 0 && !pair_eq && lo == phi;
    const bool is_first = in_list && (which ? !pair_eq : !(pair_eq || shift));
    const unsigned long long fm = __ballot(is_first);
    const int cnt = __popcll(fm & (lanemask_lt() | (1ull << lane)));
    const int idx = is_first ? cnt - 1 : ((which == 0 && pair_eq) ? cnt - 2 : cnt - 1);
    RowDup d;
    d.all_valid = __all(!in_list || ok) && !(y2 < y1);
    d.t_hi = __shfl(idx, (2 * lane + 1) & 63);
    d.t_fr = __shfl(fr, (2 * lane) & 63);
    if (lane >= ch) d.t_hi = -1;
    d.t_row = __builtin_amdgcn_ds_permute((is_first ? cnt - 1 : 63) << 2, which ? hi : lo);
    d.nr = __popcll(fm);
    return d;
}

// Four channels as two float pairs: arithmetic on a pair is one packed instruction
// (v_pk_add_f32 / v_pk_mul_f32), the same issue cost as the scalar form for twice the work.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Quad { f32x2 a, b; };

__device__ __forceinline__ Quad quad_lerp(Quad p, Quad q, float w) {
    const f32x2 w2 = {w, w};
    Quad r;
    r.a = p.a + (q.a - p.a) * w2;
    r.b = p.b + (q.b - p.b) * w2;
    return r;
}
__device__ __forceinline__ Quad quad_load_lds(const unsigned char* p) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    Quad r;
    r.a = __builtin_shufflevector(t, t, 0, 1);
    r.b = __builtin_shufflevector(t, t, 2, 3);
    return r;
}

// Fast walk: every sample row and every sample column of these groups lies inside the image.
template <int POOL, bool WIN, int NQ>
__device__ __forceinline__ void staged_walk_hot(const StageCtx& c, const RowDup& rd,
                                                const AxisUnique& ux, int g0) {
    const int lane = lane_id();
    const int cq = lane & (RW_LANES - 1), ql = lane / RW_LANES;
    constexpr int NS = NQ * POOL;
    uint32_t xlo[NS], xhi[NS];
    float xfr[NS];
    bool q_live[NQ];
    uint32_t ooff[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int q = (g0 + t) * 4 + ql;
        q_live[t] = q < c.pw;
        // columns past the crop: an offset no buffer range holds, the store is dropped
        ooff[t] = q_live[t] ? (uint32_t)q * (uint32_t)c.C * 4u + c.chan_off : 0x80000000u;
#pragma unroll
        for (int k = 0; k < POOL; ++k) {
            const int e = t * POOL + k;
            const int tl = (t * 4 + ql) * POOL + k;          // sample's place in ux's tables
            const int li = __shfl(ux.t_lo, tl), ri = __shfl(ux.t_hi, tl);
            xfr[e] = __shfl(ux.t_fr, tl);
            xlo[e] = (uint32_t)max(li, 0) * (RW_CS * 4u) + (uint32_t)cq * 16u;
            xhi[e] = (uint32_t)max(ri, 0) * (RW_CS * 4u) + (uint32_t)cq * 16u;
        }
    }
    const int npiece = (ux.n + 3) >> 2;                                   // uniform, 1..4
    uint32_t voff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int u = min(4 * p + ql, ux.n - 1);
        voff[p] = (uint32_t)__shfl(ux.t_val, u) * c.sw_bytes + c.chan_off;
    }
    const int NR = rd.nr;
    const unsigned nsy = (unsigned)c.nsy;
    const f32x2 zero2 = {0.f, 0.f};
    Quad best[NQ];
    uchar4 arg[NQ];
    Quad hA[NS], hB[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) { hA[e].a = hA[e].b = zero2; hB[e] = hA[e]; }
#pragma unroll
    for (int t = 0; t < NQ; ++t) { best[t].a = best[t].b = zero2; arg[t] = make_uchar4(0, 0, 0, 0); }
    unsigned sy = 0;
    // vector memory operations issued so far, and that count right after each slot's staging
    // loads: their difference is what may still be outstanding when the slot is read
    int issued = 0;
    int mark[ST_NSLOT];

    auto fold = [&](unsigned s, const Quad (&v)[NS]) {
        const unsigned ky = s % POOL;
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const Quad* w = &v[t * POOL];
            if (WIN) {
#pragma unroll
                for (int kx = 0; kx < POOL; ++kx) {
                    const int k = ky * POOL + kx;
                    if (k == 0) {
                        best[t] = w[kx];
                        arg[t] = make_uchar4(0, 0, 0, 0);
                    } else {
                        if (w[kx].a.x > best[t].a.x) { best[t].a.x = w[kx].a.x; arg[t].x = k; }
                        if (w[kx].a.y > best[t].a.y) { best[t].a.y = w[kx].a.y; arg[t].y = k; }
                        if (w[kx].b.x > best[t].b.x) { best[t].b.x = w[kx].b.x; arg[t].z = k; }
                        if (w[kx].b.y > best[t].b.y) { best[t].b.y = w[kx].b.y; arg[t].w = k; }
                    }
                }
            } else if (POOL == 1) {
                best[t] = w[0];
            } else if (ky == 0) {
                best[t].a.x = vmax2(w[0].a.x, w[POOL - 1].a.x);
                best[t].a.y = vmax2(w[0].a.y, w[POOL - 1].a.y);
                best[t].b.x = vmax2(w[0].b.x, w[POOL - 1].b.x);
                best[t].b.y = vmax2(w[0].b.y, w[POOL - 1].b.y);
            } else {
                best[t].a.x = vmax3(best[t].a.x, w[0].a.x, w[POOL - 1].a.x);
                best[t].a.y = vmax3(best[t].a.y, w[0].a.y, w[POOL - 1].a.y);
                best[t].b.x = vmax3(best[t].b.x, w[0].b.x, w[POOL - 1].b.x);
                best[t].b.y = vmax3(best[t].b.y, w[0].b.y, w[POOL - 1].b.y);
            }
        }
        if (ky == POOL - 1) {
            const uint32_t orow = (s / POOL) * c.orow_bytes;                           // uniform
            issued = __builtin_amdgcn_readfirstlane(issued + NQ * (WIN ? 2 : 1));
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                buf_store4(c.obuf, ooff[t], orow,
                           make_float4(best[t].a.x, best[t].a.y, best[t].b.x, best[t].b.y));
                if (WIN && q_live[t])
                    *reinterpret_cast<uchar4*>(c.winner_roi + ((orow + ooff[t]) >> 2)) = arg[t];
            }
        }
    };
    auto stage_row = [&](int j, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const uint32_t roff = (uint32_t)rl(rd.t_row, j) * c.row_bytes;                 // uniform
        lds_dma_pieces(c.img_dma, voff[0], voff[1], voff[2], voff[3], roff,
                       c.ring_addr + (uint32_t)slot * ST_SLOT_BYTES, npiece);
        issued += npiece;
        mark[slot] = issued;
    };
    auto step = [&](int j, auto slot_c, Quad (&hc)[NS], const Quad (&hp)[NS]) {
        constexpr int slot = decltype(slot_c)::value;
        // Row j has landed in its slot.  With two slots the only vector-memory operations issued
        // between a row's staging and its read are output-row stores, so a non-zero count is at
        // least one row's stores; with more slots later rows' staging loads (1..4 pieces) sit in
        // between and only the exact count is safe.
        if constexpr (ST_NSLOT == 2) wait_vm_zero_or<NQ * (WIN ? 2 : 1)>(issued - mark[slot]);
        else                         wait_vm_at_most(issued - mark[slot]);
        const unsigned char* sp = c.ring + slot * ST_SLOT_BYTES;
        Quad cl[NS], cr[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            cl[e] = quad_load_lds(sp + xlo[e]);
            cr[e] = quad_load_lds(sp + xhi[e]);
        }
        asm volatile("" ::: "memory");
        if (j + ST_NSLOT - 1 < NR)                         // into the slot read one step ago
            stage_row(j + ST_NSLOT - 1, IntTag<(slot + ST_NSLOT - 1) % ST_NSLOT>{});
#pragma unroll
        for (int e = 0; e < NS; ++e) hc[e] = quad_lerp(cl[e], cr[e], xfr[e]);
        while (sy < nsy && rl(rd.t_hi, sy) == j) {        // sample rows whose ceil row this is
            const float fr = rlf(rd.t_fr, sy);
            Quad v[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) v[e] = quad_lerp(hp[e], hc[e], fr);
            fold(sy, v);
            ++sy;
        }
    };
    // steps j .. j+U-1 with compile-time slot and ping-pong parity
    constexpr int U = (ST_NSLOT % 2) ? 2 * ST_NSLOT : ST_NSLOT;
    auto chain = [&](auto self, auto uc, int j) -> void {
        constexpr int u = decltype(uc)::value;
        if constexpr (u < U) {
            if (j + u < NR) {
                if constexpr (u & 1) step(j + u, IntTag<u % ST_NSLOT>{}, hB, hA);
                else                 step(j + u, IntTag<u % ST_NSLOT>{}, hA, hB);
                self(self, IntTag<u + 1>{}, j);
            }
        }
    };
    auto prologue = [&](auto self, auto ic) -> void {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < ST_NSLOT - 1) {
            if (i < NR) stage_row(i, IntTag<i>{});
            self(self, IntTag<i + 1>{});
        }
    };
#pragma unroll 1
    for (int sub = 0; sub < c.nsub; ++sub) {
        sy = 0;
        issued = 0;
#pragma unroll
        for (int i = 0; i < ST_NSLOT; ++i) mark[i] = 0;
        prologue(prologue, IntTag<0>{});
#pragma unroll 1
        for (int j = 0; j < NR; j += U) chain(chain, IntTag<0>{}, j);
#pragma unroll
        for (int p = 0; p < 4; ++p) voff[p] += RW_CS * 4;
#pragma unroll
        for (int t = 0; t < NQ; ++t) ooff[t] += RW_CS * 4;
    }
}

template <int POOL, bool WIN>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FRCNN_ST_WPE, FRCNN_ST_WPE)))
roi_fwd_stage_kernel(const float* __restrict__ fmap, int H, int W, int C, long long sb, long long sh,
                     long long sw, const float4* __restrict__ rois, int R, int ch, int cw,
                     int ngrp, int nsub, float* __restrict__ out, uint8_t* __restrict__ winner) {
    // ngrp workgroups per RoI, each walks nsub consecutive 64-channel slices
    __shared__ __attribute__((aligned(1024))) unsigned char st_ring[ST_NSLOT * ST_SLOT_BYTES];
    const int lane = threadIdx.x;
    const int slice = (blockIdx.x % ngrp) * nsub, r = blockIdx.x / ngrp, b = blockIdx.y;
    StageCtx c;
    c.box = rois[(size_t)b * R + r];
    const bool flipped = c.box.z < c.box.x || c.box.w < c.box.y;
    c.nsub = nsub;
    c.W = W; c.C = C; c.cw = cw;
    c.ph = ch / POOL; c.pw = cw / POOL; c.nsy = c.ph * POOL;
    c.row_bytes = (uint32_t)sh * 4u;
    c.sw_bytes = (uint32_t)sw * 4u;
    const int cq = lane & (RW_LANES - 1), ql = lane / RW_LANES;
    c.chan_off = (uint32_t)(slice * RW_CS + cq * 4) * 4u;
    c.orow_bytes = (uint32_t)c.pw * (uint32_t)C * 4u;
    const size_t oroi = ((size_t)b * R + r) * (size_t)(c.ph * c.pw) * C;
#ifdef FRCNN_ST_NOSTORE      // timing experiment: zero records, every store is dropped
    c.obuf = __builtin_amdgcn_make_buffer_rsrc(out + oroi, 0, 0, 0x00020000);
#else
    c.obuf = __builtin_amdgcn_make_buffer_rsrc(out + oroi, 0, (uint32_t)c.ph * c.orow_bytes, 0x00020000);
#endif
    c.winner_roi = WIN ? winner + oroi : nullptr;
    c.ring = st_ring;
    c.ring_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)st_ring);
    const float* img_ptr = fmap + (size_t)b * sb;
    const uint64_t img_addr = (uint64_t)(uintptr_t)img_ptr;
    c.img_dma.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)img_addr);
    c.img_dma.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(img_addr >> 32) & 0xFFFFu));
#ifdef FRCNN_ST_NOLOAD       // timing experiment: zero records, every staging load is dropped
    c.img_dma.z = 0;
#else
    c.img_dma.z = __builtin_amdgcn_readfirstlane((int)((uint32_t)H * c.row_bytes));
#endif
    c.img_dma.w = 0x00020000;
    const int ngroup = (c.pw + 3) >> 2;

    if (flipped) {
        // descending rows or columns: fetch every corner directly, four pooled columns at a time
        c.rw = build_row_walk(c.box.x, c.box.z, H, ch);
        const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(img_ptr), 0, (uint32_t)H * c.row_bytes, 0x00020000);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sub = 0; sub < nsub; ++sub, c.chan_off += RW_CS * 4)
        for (int g = 0; g < ngroup; ++g) {
            const int q = g * 4 + ql;
            uint32_t xlo[POOL], xhi[POOL];
            float xfr[POOL];
            bool xok[POOL];
#pragma unroll
            for (int k = 0; k < POOL; ++k) {
                int lo = 0, hi = 0, ok = 0;
                float fr = 0.f;
                const int sx = q * POOL + k;
                if (sx < cw) axis_sample(c.box.y, c.box.w, W, cw, sx, lo, hi, fr, ok);
                xlo[k] = (uint32_t)lo * c.sw_bytes + c.chan_off;
                xhi[k] = (uint32_t)hi * c.sw_bytes + c.chan_off;
                xfr[k] = fr;
                xok[k] = sx < cw && ok;
            }
            float4 best = zero;
            uchar4 arg = make_uchar4(0, 0, 0, 0);
            for (int sy = 0; sy < c.nsy; ++sy) {
                const int lo = rl(c.rw.t_lo, sy), hi = rl(c.rw.t_hi, sy);
                const float fr = rlf(c.rw.t_fr, sy);
#pragma unroll
                for (int k = 0; k < POOL; ++k) {
                    float4 v = zero;
                    if (lo >= 0 && xok[k]) {
                        const int rlo = c.rw.walk ? rl(c.rw.t_row, lo) : lo;
                        const int rhi = c.rw.walk ? rl(c.rw.t_row, hi) : hi;
                        const uint32_t rt = (uint32_t)rlo * c.row_bytes, rb = (uint32_t)rhi * c.row_bytes;
                        const float4 top = lerp4(buf_load4(img, xlo[k], rt), buf_load4(img, xhi[k], rt),
                                                 xfr[k]);
                        const float4 bot = lerp4(buf_load4(img, xlo[k], rb), buf_load4(img, xhi[k], rb),
                                                 xfr[k]);
                        v = lerp4(top, bot, fr);
                    }
                    const int kk = (sy % POOL) * POOL + k;
                    if (kk == 0) {
                        best = v;
                        arg = make_uchar4(0, 0, 0, 0);
                    } else {
                        if (v.x > best.x) { best.x = v.x; arg.x = kk; }
                        if (v.y > best.y) { best.y = v.y; arg.y = kk; }
                        if (v.z > best.z) { best.z = v.z; arg.z = kk; }
                        if (v.w > best.w) { best.w = v.w; arg.w = kk; }
                    }
                }
                if (sy % POOL == POOL - 1 && q < c.pw) {
                    const uint32_t o = (uint32_t)(sy / POOL) * c.orow_bytes +
                                       (uint32_t)q * (uint32_t)C * 4u + c.chan_off;
                    *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(out + oroi) + o) = best;
                    if (WIN) *reinterpret_cast<uchar4*>(c.winner_roi + (o >> 2)) = arg;
                }
            }
        }
        return;
    }

    int g = 0;
#pragma unroll 1
    while (g < ngroup) {
        // rebuilt per walk rather than kept in registers across one
        const RowDup rd = build_row_dup(c.box.x, c.box.z, H, ch);
        if (FRCNN_ST_PAIR && rd.all_valid && g + 1 < ngroup) {
            const AxisUnique u2 = axis_unique(c.box.y, c.box.w, W, cw, g * 4 * POOL, 8 * POOL);
            const bool cols_in = __all(lane >= 8 * POOL || g * 4 * POOL + lane >= cw || u2.t_lo >= 0);
            if (cols_in && u2.n <= ST_CELLS) {            // uniform
                staged_walk_hot<POOL, WIN, 2>(c, rd, u2, g);
                g += 2;
                continue;
            }
        }
        const AxisUnique u1 = axis_unique(c.box.y, c.box.w, W, cw, g * 4 * POOL, 4 * POOL);
        const bool cols_in = __all(lane >= 4 * POOL || g * 4 * POOL + lane >= cw || u1.t_lo >= 0);
        if (rd.all_valid && cols_in) {
            staged_walk_hot<POOL, WIN, 1>(c, rd, u1, g);
        } else {
            c.rw = build_row_walk(c.box.x, c.box.z, H, ch);
            staged_walk<POOL, WIN, 1>(c, u1, g);
        }
        g += 1;
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

// winner bytes of a plain crop (pool 1: the only sample of each window is sample 0)
__global__ void __launch_bounds__(256) roi_zero_winner_kernel(uint32_t* __restrict__ p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = 0u;
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

// ---- backward, fast path: one wave per feature cell (or two neighbours of a row), no LDS,
// no barrier.  Lane = RoI while candidates are detected (64 at a time, conservative reach
// test, ballot), then lane = channel quad while a candidate is applied.  For a
// candidate the wave evaluates all sample rows (lanes 0-31) and sample columns
// (lanes 32-63) at once; ballots give the sample rows / columns whose bilinear
// footprint contains the wave's cells, and the (few) hits are walked by scalar loops.
// Summation order per cell: RoI ascending, sample (sy,sx) row-major, corner
// TL,TR,BL,BR.

// One candidate RoI applied to one cell.  ylo_m / yhi_m: sample rows whose floor / ceil row is
// the cell's row, xlo_m / xhi_m likewise for columns; `fr` holds the lerp fraction of sample
// row sy in lane ylane0 + sy and of sample column sx in lane xlane0 + sx.  Order: sample
// (sy,sx) row-major, corner TL,TR,BL,BR.  cq = the lane's first channel quad.
// The gradient and winner quads of the RoI come through buffer descriptors (base = the RoI's
// first bin, records = its ph * pw * C elements): a pair's address is one scalar offset
// (bin * C) next to the lane's fixed vector offset, and lanes beyond C carry an offset past
// the records, so they read 0 without a branch.
template <int POOL, int NQ, int CX>
__device__ __forceinline__ void bwd_apply_candidate(
    float4 (&acc)[CX][NQ], __amdgpu_buffer_rsrc_t gbuf, __amdgpu_buffer_rsrc_t wbuf, int pw, int C,
    const uint32_t (&voff_g)[NQ], const uint32_t (&voff_w)[NQ], uint32_t ylo_m, uint32_t yhi_m,
    const uint32_t (&xlo_m)[CX], const uint32_t (&xhi_m)[CX], float fr, int ylane0, int xlane0) {
    uint32_t ym = ylo_m | yhi_m;
    uint32_t xm_all = 0u;
#pragma unroll
    for (int c = 0; c < CX; ++c) xm_all |= xlo_m[c] | xhi_m[c];
    if (ym == 0u || xm_all == 0u) return;
    while (ym != 0u) {
        const int sy = __ffs((int)ym) - 1;
        ym &= ym - 1u;
        const float ly = rlf(fr, ylane0 + sy);
        const float wt = 1.0f - ly, wb = ly;
        // sample columns per corner of this sample row: one scalar bit test per corner and pair
        uint32_t tl_m[CX], tr_m[CX], bl_m[CX], br_m[CX];
#pragma unroll
        for (int c = 0; c < CX; ++c) {
            tl_m[c] = (ylo_m >> sy) & 1u ? xlo_m[c] : 0u; tr_m[c] = (ylo_m >> sy) & 1u ? xhi_m[c] : 0u;
            bl_m[c] = (yhi_m >> sy) & 1u ? xlo_m[c] : 0u; br_m[c] = (yhi_m >> sy) & 1u ? xhi_m[c] : 0u;
        }
        const uint32_t rowbin = (uint32_t)((sy / POOL) * pw);
        uint32_t xm = xm_all;
        while (xm != 0u) {
            const int sx = __ffs((int)xm) - 1;
            xm &= xm - 1u;
            const float lx = rlf(fr, xlane0 + sx);
            const float wl = 1.0f - lx, wr = lx;
            const uint32_t k = (uint32_t)((sy % POOL) * POOL + (sx % POOL));
            const uint32_t soff = (rowbin + (uint32_t)(sx / POOL)) * (uint32_t)C;   // elements
            // all loads of the pair go out before the first use
            float4 gq[NQ];
            uint32_t wq[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                gq[u] = buf_load4(gbuf, voff_g[u], soff * 4u);
                wq[u] = POOL > 1 ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wbuf, voff_w[u], soff, 0)
                                 : 0u;
            }
            if (POOL > 1) {
                // a non-winner channel adds +0 (x + 0 == x): same value as skipping it
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const uint32_t wv = wq[u];
                    if ((wv & 0xFFu) != k) gq[u].x = 0.f;
                    if (((wv >> 8) & 0xFFu) != k) gq[u].y = 0.f;
                    if (((wv >> 16) & 0xFFu) != k) gq[u].z = 0.f;
                    if ((wv >> 24) != k) gq[u].w = 0.f;
                }
            }
            // corners in the order TL, TR, BL, BR; most pairs have one.  The empty asm
            // keeps each corner a (wave-uniform) branch: as selects, every pair would
            // pay for all four
#define FRCNN_CORNER(M, wy, wx)                                                                  \
            _Pragma("unroll") for (int c = 0; c < CX; ++c)                               \
            if ((M[c] >> sx) & 1u) {                                                     \
                asm volatile("");                                                        \
                _Pragma("unroll") for (int u = 0; u < NQ; ++u) {                         \
                    const float4 g = gq[u];                                              \
                    const float4 d = make_float4(wy * g.x, wy * g.y, wy * g.z, wy * g.w); \
                    acc[c][u].x += wx * d.x; acc[c][u].y += wx * d.y;                    \
                    acc[c][u].z += wx * d.z; acc[c][u].w += wx * d.w;                    \
                }                                                                        \
            }
            FRCNN_CORNER(tl_m, wt, wl)
            FRCNN_CORNER(tr_m, wt, wr)
            FRCNN_CORNER(bl_m, wb, wl)
            FRCNN_CORNER(br_m, wb, wr)
#undef FRCNN_CORNER
        }
    }
}

template <int POOL, int NQ, int LW, int CX>
__global__ void __launch_bounds__(64)
roi_bwd_wave_kernel(const float* __restrict__ dY, const uint8_t* __restrict__ winner,
                    const float4* __restrict__ rois, int R, int r_begin, int r_end, int H, int W,
                    int C, int ch, int cw, int nitem, int nsl, float* __restrict__ dfmap) {
    // The sample geometry of a candidate takes LW lanes per axis (crop <= LW), so one
    // evaluation serves NC = 32 / LW candidates: lanes [2*LW*t, 2*LW*t + LW) hold the sample
    // rows of candidate t, the next LW lanes its sample columns.
    constexpr int NC = 32 / LW;
    constexpr uint32_t FIELD = LW == 32 ? 0xFFFFFFFFu : ((1u << (LW & 31)) - 1u);
    // A wave owns CX neighbouring cells of a row: a sample's left and right cell are neighbours,
    // so with CX = 2 a fetched pair serves both of them (three pairs per two cells where
    // one cell alone asks for two).
    const int ncx = (W + CX - 1) / CX;
    // Workgroup n runs on XCD n % 8, each XCD with its own L2.  The cells of one (image, channel
    // slice) all go to ONE XCD: a bin of pooled gradients is asked for by the handful of waves
    // around its samples, and only that XCD's L2 ever holds it (with an image spread over the
    // XCDs as bands of rows, the bins along the band edges went through two L2s, and every L2
    // streamed a share of every image in flight).
    int cell, b, slice;
    {
        const int ncell = H * ncx, n = (int)blockIdx.x, full = nitem & ~7;
        int item;
        if (n < full * ncell) {
            const int slot = n >> 3;
            item = 8 * (slot / ncell) + (n & 7);
            cell = slot % ncell;
        } else {
            const int m = n - full * ncell;
            item = full + m / ncell;
            cell = m % ncell;
        }
        b = item / nsl;
        slice = item - b * nsl;
    }
    const int x = (cell % ncx) * CX, y = cell / ncx;
    const int lane = threadIdx.x;
    const int cq = slice * 64 * NQ + lane;                   // a slice = 256*NQ channels
    const int ph = ch / POOL, pw = cw / POOL;
    const int C4 = C >> 2;
    // RoIs [r_begin, r_end) of every image: a launch per range keeps the cells of a band within
    // a few dozen RoIs of each other, so the pooled gradients they share are still in the L2
    // when the next cell asks for them.  Later ranges continue the sums of the earlier ones
    // (same order of additions as one pass over all RoIs).
    float* dst = dfmap + (((size_t)b * H + y) * W + x) * C;
    float4 acc[CX][NQ];
#pragma unroll
    for (int c = 0; c < CX; ++c)
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int c4 = cq + 64 * u;
        acc[c][u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r_begin > 0 && c4 < C4 && x + c < W)
            acc[c][u] = *reinterpret_cast<const float4*>(dst + (size_t)c * C + c4 * 4);
    }
    // this lane's byte offsets inside a bin; past every record for channels beyond C
    uint32_t voff_g[NQ], voff_w[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int c4 = cq + 64 * u;
        voff_g[u] = c4 < C4 ? (uint32_t)c4 * 16u : 0x7FFFFFF0u;
        voff_w[u] = c4 < C4 ? (uint32_t)c4 * 4u : 0x7FFFFFF0u;
    }
    const uint32_t roi_elems = (uint32_t)(ph * pw) * (uint32_t)C;
    const float4* rb = rois + (size_t)b * R;
    const int field = lane / LW;
    const int ci = field >> 1;                               // this lane's candidate of the group
    const bool isx = field & 1;
    const int s = lane & (LW - 1);
    const int target = isx ? x : y;
    const int ns = isx ? cw : ch, dim = isx ? W : H;
#pragma unroll 1
    for (int r0 = r_begin; r0 < r_end; r0 += 64) {
        const int r = r0 + lane;
        float4 line = make_float4(0.f, 0.f, 0.f, 0.f);        // y base, x base, y scale, x scale
        bool cand = false;
        if (r < r_end) {
            const float4 box = rb[r];
            axis_line(box.x, box.z, H, ch, line.x, line.z);
            axis_line(box.y, box.w, W, cw, line.y, line.w);
            cand = line_reaches(line.x, line.z, ch, y, y) &&
                   line_reaches(line.y, line.w, cw, x, min(x + CX, W) - 1);
        }
        unsigned long long m = __ballot(cand);
#pragma unroll 1
        while (m != 0ull) {                                   // wave-uniform
            // the next NC candidates (RoI ascending), their corners selected per lane
            int lc[NC];
            int nc = 0;
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                lc[t] = 0;
                if (m != 0ull) {                              // wave-uniform
                    lc[t] = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    nc = t + 1;
                    const float by1 = rlf(line.x, lc[t]), bx1 = rlf(line.y, lc[t]);
                    const float by2 = rlf(line.z, lc[t]), bx2 = rlf(line.w, lc[t]);
                    if (ci == t) { c1 = isx ? bx1 : by1; c2 = isx ? bx2 : by2; }
                }
            }
            int lo = 0, hi = 0, ok = 0;
            float fr = 0.f;
            const bool act = s < ns && ci < nc;
            if (act) line_sample(c1, c2, dim, ns, s, lo, hi, fr, ok);
            const bool v = act && ok;
            const unsigned long long mlo = __ballot(v && lo == target);
            const unsigned long long mhi = __ballot(v && hi == target);
            // the second cell's column (x + 1 < W whenever a sample has it as floor or ceil)
            const unsigned long long mlo1 = CX > 1 ? __ballot(v && lo == target + 1) : 0ull;
            const unsigned long long mhi1 = CX > 1 ? __ballot(v && hi == target + 1) : 0ull;
#pragma unroll 1
            for (int t = 0; t < nc; ++t) {
            const int sh = t * 2 * LW;                        // first lane of candidate t
            const int l = t == 0 ? lc[0] : lc[NC - 1];
            static_assert(NC <= 2, "candidate t is picked from lc[0] / lc[1]");
            const uint32_t ylo_m = (uint32_t)(mlo >> sh) & FIELD;
            const uint32_t yhi_m = (uint32_t)(mhi >> sh) & FIELD;
            uint32_t xlo_m[CX], xhi_m[CX];
            xlo_m[0] = (uint32_t)(mlo >> (sh + LW)) & FIELD;
            xhi_m[0] = (uint32_t)(mhi >> (sh + LW)) & FIELD;
            uint32_t xany = xlo_m[0] | xhi_m[0];
            if (CX > 1) {
                xlo_m[CX - 1] = (uint32_t)(mlo1 >> (sh + LW)) & FIELD;
                xhi_m[CX - 1] = (uint32_t)(mhi1 >> (sh + LW)) & FIELD;
                xany |= xlo_m[CX - 1] | xhi_m[CX - 1];
            }
            static_assert(CX <= 2, "one extra pair of ballots");
            if (((ylo_m | yhi_m) == 0u) || xany == 0u) continue;
            const size_t rbase = ((size_t)b * R + (r0 + l)) * (size_t)roi_elems;
            const __amdgpu_buffer_rsrc_t gbuf = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(dY + rbase), 0, roi_elems * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t wbuf = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(POOL > 1 ? winner + rbase : nullptr), 0, POOL > 1 ? roi_elems : 0u,
                0x00020000);
            bwd_apply_candidate<POOL, NQ, CX>(acc, gbuf, wbuf, pw, C, voff_g, voff_w, ylo_m, yhi_m,
                                          xlo_m, xhi_m, fr, sh, sh + LW);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CX; ++c)
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int c4 = cq + 64 * u;
        if (c4 < C4 && x + c < W) *reinterpret_cast<float4*>(dst + (size_t)c * C + c4 * 4) = acc[c][u];
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
        const int nslice = C / RW_CS;
        // a wave walks FRCNN_ST_NSUB slices of a RoI in turn (its geometry is worked out once),
        // unless one slice per wave is what it takes to give every SIMD a few waves
        const bool few = (long long)R * B * nslice <= 16384;
        const int nsub = !few && FRCNN_ST_NSUB <= nslice && nslice % FRCNN_ST_NSUB == 0 ? FRCNN_ST_NSUB : 1;
        const int ngrp = nslice / nsub;
        const dim3 g3((unsigned)(R * ngrp), B), b3(64);
#define FRCNN_WALK(P, WN)                                                                       \
    FRCNN_LAUNCH((roi_fwd_stage_kernel<P, WN>), g3, b3, 0, st, fmap, H, W, C, (long long)sb,    \
                 (long long)sh, (long long)sw, rb, R, crop_h, crop_w, ngrp, nsub, out, winner)
        if (pool == 2) { if (winner) FRCNN_WALK(2, true); else FRCNN_WALK(2, false); }
        else {
            FRCNN_WALK(1, false);
            if (winner) {       // C % 64 == 0 and winner is 4-byte aligned here: whole words
                const size_t n4 = (size_t)B * R * crop_h * crop_w * C / 4;
                const size_t want = (n4 + 255) / 256;
                FRCNN_LAUNCH(roi_zero_winner_kernel, dim3((unsigned)(want < 8192 ? want : 8192)),
                             dim3(256), 0, st, reinterpret_cast<uint32_t*>(winner), n4);
            }
        }
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
    if ((C % 4) != 0 || B > 65535 || (long long)H * W > 0x7FFFFFF0ll) return FRCNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dY) % 16) || (reinterpret_cast<uintptr_t>(dfmap) % 16) ||
        (winner && (reinterpret_cast<uintptr_t>(winner) % 4)))
        return FRCNN_E_BADARG;
    const dim3 grid(W, H, B), block(128);
    const float4* rb = reinterpret_cast<const float4*>(rois);
    hipStream_t st = (hipStream_t)stream;
    // patch-owner kernel (frcnn_roi_bwd.hip): pool 2, crops up to 16 x 16 (B = 64: 8.6 ms against
    // 11.1 ms for the gather kernel below, B = 8: 1.56 against 1.90; bit-identical results)
    if (pool == 2 && tuning(TUNE_BWD_BAND) != 0) {
        const int ps = launch_roi_bwd_patch(dY, winner, rois, B, R, H, W, C, crop_h, crop_w, dfmap, st);
        if (ps != FRCNN_E_UNSUPPORTED) return ps;
    }
    // the wave-per-cell kernel addresses a RoI's bins with 32-bit byte offsets
    const bool roi_fits = (long long)(crop_h / pool) * (crop_w / pool) * C * 4 < 0x7FFFFFF0ll;
    if (crop_h <= 32 && crop_w <= 32 && roi_fits) {
        const int nq = (C / 4 + 63) / 64;                 // channel quads per lane; above 8 the
                                                          // channels are split over blockIdx.y
        const int nqk = nq <= 1 ? 1 : nq <= 2 ? 2 : nq <= 4 ? 4 : 8;
        const int nsl = (C / 4 + 64 * nqk - 1) / (64 * nqk);
        const bool pack2 = crop_h <= 16 && crop_w <= 16;      // two candidates per evaluation
        // cells per wave: two neighbours of a row share the pairs they fetch (B = 64: 12.4 ms
        // against 14.3 ms with one cell per wave; B = 8: 2.09 against 2.14)
        int cx = 2;
        if (tuning(TUNE_BWD_CX) == 1) cx = 1;
        const long long nwave = (long long)H * ((W + cx - 1) / cx) * B * nsl;
        if (nwave > 0x7FFFFFF0ll) return FRCNN_E_UNSUPPORTED;
        const dim3 cgrid((unsigned)nwave);
        // RoIs per launch (a multiple of 64): a launch per range keeps the waves of an image
        // within a few hundred RoIs of each other, so more of the bins they share are still in
        // the L2 when the next wave asks.  B = 64: 11.1 ms at 512, 11.3 at 256, 12.2 in one
        // launch; B = 8 has too few waves to fill the chip twice and is best in one launch.
        int chunk = B >= 32 ? 512 : R;
        if (const long long t = tuning(TUNE_BWD_CHUNK); t != TUNE_UNSET && t > 0) chunk = (int)(t < R ? t : R);
        if (chunk <= 0 || chunk > R) chunk = R;
        chunk = (chunk + 63) / 64 * 64;
#define FRCNN_BWD(P, Q)                                                                          \
    do {                                                                                         \
        for (int rb0 = 0; rb0 < R; rb0 += chunk) {                                               \
            const int re = rb0 + chunk < R ? rb0 + chunk : R;                                    \
            if (pack2 && cx == 2)                                                                \
                FRCNN_LAUNCH((roi_bwd_wave_kernel<P, Q, 16, 2>), cgrid, dim3(64), 0, st, dY, winner, \
                             rb, R, rb0, re, H, W, C, crop_h, crop_w, B * nsl, nsl, dfmap);      \
            else if (pack2)                                                                      \
                FRCNN_LAUNCH((roi_bwd_wave_kernel<P, Q, 16, 1>), cgrid, dim3(64), 0, st, dY, winner, \
                             rb, R, rb0, re, H, W, C, crop_h, crop_w, B * nsl, nsl, dfmap);      \
            else if (cx == 2)                                                                    \
                FRCNN_LAUNCH((roi_bwd_wave_kernel<P, Q, 32, 2>), cgrid, dim3(64), 0, st, dY, winner, \
                             rb, R, rb0, re, H, W, C, crop_h, crop_w, B * nsl, nsl, dfmap);      \
            else                                                                                 \
                FRCNN_LAUNCH((roi_bwd_wave_kernel<P, Q, 32, 1>), cgrid, dim3(64), 0, st, dY, winner, \
                             rb, R, rb0, re, H, W, C, crop_h, crop_w, B * nsl, nsl, dfmap);      \
        }                                                                                        \
    } while (0)
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
    if (C > 4 * 4 * 128 || H > 65535) return FRCNN_E_UNSUPPORTED;   // large-crop kernel's limits
    if (pool == 2)
        FRCNN_LAUNCH(roi_bwd_cl_kernel<2>, grid, block, 0, st, dY, winner, rb, R, H, W, C,
                           crop_h, crop_w, dfmap);
    else
        FRCNN_LAUNCH(roi_bwd_cl_kernel<1>, grid, block, 0, st, dY, winner, rb, R, H, W, C,
                           crop_h, crop_w, dfmap);
    return FRCNN_OK;
}
