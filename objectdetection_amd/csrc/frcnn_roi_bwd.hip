// RoI crop + 2x2 max-pool, backward, patch-owner form (the adjoint of Faster_RCNN/module/model.py:
// 53-55 + :30; the reference stops the gradient there, model.py:71, so the summation order is
// this build's own: per feature cell and channel, RoI ascending, sample (sy,sx) row-major,
// corner TL,TR,BL,BR, each contribution wx * (wy * g) rounded as written and added on its own).
//
// One wave owns a PATCH of PY x PX feature cells of one image for 256 channels (lane = channel
// quad) and keeps their sums in registers for the whole launch: four registers per cell,
// addressed through the VGPR index mode (s_set_gpr_idx_on: the register number of an operand is
// its static number plus M0), so a sample's four corners are four `index, add, add` groups and
// the cell a corner lands in is a scalar, not a branch.  A corner outside the patch (it belongs
// to the neighbouring patch's wave) adds into a spare register group that is never stored:
// skipping such corners with branches was measured and costs more than the additions it saves.
//
// The wave walks the image's RoIs in order: lanes test 64 RoIs at a time for reaching the patch
// (ballot), a candidate's 2 x 16 sample rows / columns are evaluated by 32 lanes at once, and the
// walk is bin row -> sample row -> sample column with every loop bound a bit mask, so only the
// samples that touch the patch are visited.  The pooled gradients and winner bytes of a bin row's
// bins are fetched together, before the first is used, and serve both of its sample rows.
// Against one wave per two cells, a bin is fetched by the ~2 patches its samples touch instead of
// ~10 cell pairs, and the per-RoI geometry is evaluated per patch, not per cell.
#include "frcnn_common.h"

namespace frcnn {

typedef float bp_f2 __attribute__((ext_vector_type(2)));
typedef float bp_f4 __attribute__((ext_vector_type(4)));
typedef float bp_f16 __attribute__((ext_vector_type(16)));

constexpr int BP_MAXCROP = 16;                  // samples per axis the 32 geometry lanes hold

// The sums live in v[64 : 64 + 16 NG), NG = PY * PX / 4 groups of sixteen registers, the spare
// group in the four registers behind them.  Every asm statement that touches them names all of
// them as tied operands, so the compiler keeps them where they are between statements.
template <int NG> struct BpRegs { bp_f16 a[NG]; bp_f4 spare; };
#define BP_ACC6(r)                                                                               \
    "+{v[64:79]}"(r.a[0]), "+{v[80:95]}"(r.a[1]), "+{v[96:111]}"(r.a[2]), "+{v[112:127]}"(r.a[3]), \
    "+{v[128:143]}"(r.a[4]), "+{v[144:159]}"(r.a[5]), "+{v[160:163]}"(r.spare)
#define BP_ACC4(r)                                                                               \
    "+{v[64:79]}"(r.a[0]), "+{v[80:95]}"(r.a[1]), "+{v[96:111]}"(r.a[2]), "+{v[112:127]}"(r.a[3]), \
    "+{v[128:131]}"(r.spare)
#define BP_ACC3(r)                                                                               \
    "+{v[64:79]}"(r.a[0]), "+{v[80:95]}"(r.a[1]), "+{v[96:111]}"(r.a[2]), "+{v[112:115]}"(r.spare)

#define BP_ADD(i, lo, hi)                                                                        \
    "s_set_gpr_idx_idx %[" #i "]\n\t"                                                            \
    "v_pk_add_f32 v[64:65], v[64:65], %[" #lo "]\n\t"                                            \
    "v_pk_add_f32 v[66:67], v[66:67], %[" #hi "]\n\t"
#define BP_ADD_FIRST(i, lo, hi)                                                                  \
    "s_set_gpr_idx_on %[" #i "], 0x9\n\t"                                                        \
    "v_pk_add_f32 v[64:65], v[64:65], %[" #lo "]\n\t"                                            \
    "v_pk_add_f32 v[66:67], v[66:67], %[" #hi "]\n\t"

// One sample's four corners: index register i*, values = channel pairs 0-1 / 2-3.  M0 is
// written; nothing else in these kernels reads it.
#define BP_DEFINE_ADD4(NG, ACC)                                                                  \
    __device__ __forceinline__ void bp_add4(BpRegs<NG>& r, int i0, bp_f2 a0, bp_f2 a1, int i1,   \
                                            bp_f2 b0, bp_f2 b1, int i2, bp_f2 c0, bp_f2 c1,      \
                                            int i3, bp_f2 d0, bp_f2 d1) {                        \
        asm volatile(BP_ADD_FIRST(i0, a0, a1) BP_ADD(i1, b0, b1) BP_ADD(i2, c0, c1)              \
                     BP_ADD(i3, d0, d1) "s_set_gpr_idx_off"                                     \
                     : ACC(r)                                                                    \
                     : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [a0] "v"(a0),      \
                       [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c0), [c1] "v"(c1),      \
                       [d0] "v"(d0), [d1] "v"(d1));                                              \
    }
BP_DEFINE_ADD4(6, BP_ACC6)
BP_DEFINE_ADD4(4, BP_ACC4)
BP_DEFINE_ADD4(3, BP_ACC3)

// g where the channel's winner byte is K, else 0 (a non-winner channel adds +0: x + 0 == x)
#define BP_MASK(dst, BYTE, wv, k, gv)                                                            \
    asm("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:" #BYTE " src1_sel:DWORD\n\t"                    \
        "v_cndmask_b32 %0, 0, %3, vcc"                                                           \
        : "=v"(dst) : "v"(wv), "v"(k), "v"(gv) : "vcc")

// One sample of a bin (its winner value in every lane of kv): the channels whose winner it is
// carry the bin's gradient through its four corners.  r_* / c_*: register offsets of the
// sample's two rows / columns, or SPARE where that row / column is not the patch's.
template <int NG>
__device__ __forceinline__ void bp_sample(BpRegs<NG>& r, bp_f4 g, uint32_t wv, uint32_t kv, float wt,
                                          float wb, float wl, float wr, int r_lo, int r_hi, int c_lo,
                                          int c_hi) {
    constexpr int SPARE = 16 * NG;
    bp_f2 m01, m23;
    BP_MASK(m01.x, BYTE_0, wv, kv, g.x);
    BP_MASK(m01.y, BYTE_1, wv, kv, g.y);
    BP_MASK(m23.x, BYTE_2, wv, kv, g.z);
    BP_MASK(m23.y, BYTE_3, wv, kv, g.w);
    const bp_f2 t01 = wt * m01, t23 = wt * m23, b01 = wb * m01, b23 = wb * m23;
    const int iTL = min(r_lo + c_lo, SPARE), iTR = min(r_lo + c_hi, SPARE);
    const int iBL = min(r_hi + c_lo, SPARE), iBR = min(r_hi + c_hi, SPARE);
    bp_add4(r, iTL, wl * t01, wl * t23, iTR, wr * t01, wr * t23, iBL, wl * b01, wl * b23, iBR,
            wr * b01, wr * b23);
}

template <int PY, int PX, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
roi_bwd_patch_kernel(const float* __restrict__ dY, const uint8_t* __restrict__ winner,
                     const float4* __restrict__ rois, int R, int H, int W, int C, int ch, int cw,
                     int npx, int npatch, int nitem, int nsl, float* __restrict__ dfmap) {
    static_assert((PY * PX) % 4 == 0, "whole groups of sixteen registers");
    constexpr int NG = PY * PX / 4, SPARE = 16 * NG;
    // Workgroup n runs on XCD n % 8, each XCD with its own L2: the patches of one (image, channel
    // slice) all go to ONE XCD, so a bin that two patches ask for is found in that L2.
    int patch, b, slice;
    {
        const int n = (int)blockIdx.x, full = nitem & ~7;
        int item;
        if (n < full * npatch) {
            const int slot = n >> 3;
            item = 8 * (slot / npatch) + (n & 7);
            patch = slot % npatch;
        } else {
            const int m = n - full * npatch;
            item = full + m / npatch;
            patch = m % npatch;
        }
        b = item / nsl;
        slice = item - b * nsl;
    }
    const int py = patch / npx, px = patch - py * npx;
    const int y0 = py * PY, x0 = px * PX;
    const int lane = threadIdx.x;
    const int c4 = slice * 64 + lane;                        // this lane's channel quad
    const int C4 = C >> 2;
    const int ph = ch >> 1, pw = cw >> 1;

    BpRegs<NG> regs;
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) regs.a[i][j] = 0.f;
    regs.spare = bp_f4{0.f, 0.f, 0.f, 0.f};

    // this lane's byte offsets inside a bin; past every record for channels beyond C
    const uint32_t voff_g = c4 < C4 ? (uint32_t)c4 * 16u : 0x7FFFFFF0u;
    const uint32_t voff_w = c4 < C4 ? (uint32_t)c4 * 4u : 0x7FFFFFF0u;
    const uint32_t roi_elems = (uint32_t)(ph * pw) * (uint32_t)C;
    const float4* rb = rois + (size_t)b * R;
    // winner values of the four samples of a bin, for the byte compares
    uint32_t kv[4] = {0u, 1u, 2u, 3u};
    asm volatile("" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]));

    // geometry lanes: 0-15 sample rows, 16-31 sample columns of the candidate
    const int s = lane & 15;
    const bool isx = (lane & 16) != 0;
    const int ns = isx ? cw : ch, dim = isx ? W : H;
    const int p0 = isx ? x0 : y0, pn = isx ? PX : PY;
    const int stride = isx ? 4 : 4 * PX;
    const bool glane = lane < 32 && s < ns;
    const int ylast = min(y0 + PY, H) - 1, xlast = min(x0 + PX, W) - 1;

#pragma unroll 1
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int r = r0 + lane;
        float4 line = make_float4(0.f, 0.f, 0.f, 0.f);        // y base, x base, y scale, x scale
        bool cand = false;
        if (r < R) {
            const float4 box = rb[r];
            axis_line(box.x, box.z, H, ch, line.x, line.z);
            axis_line(box.y, box.w, W, cw, line.y, line.w);
            cand = line_reaches(line.x, line.z, ch, y0, ylast) &&
                   line_reaches(line.y, line.w, cw, x0, xlast);
        }
        unsigned long long m = __ballot(cand);
#pragma unroll 1
        while (m != 0ull) {                                   // wave-uniform, RoI ascending
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const float by = rlf(line.x, l), bx = rlf(line.y, l);
            const float sy_ = rlf(line.z, l), sx_ = rlf(line.w, l);
            int lo = 0, hi = 0, ok = 0;
            float fr = 0.f;
            if (glane) line_sample(isx ? bx : by, isx ? sx_ : sy_, dim, ns, s, lo, hi, fr, ok);
            const bool v = glane && ok;
            const bool lo_in = v && lo >= p0 && lo < p0 + pn;
            const bool hi_in = v && hi >= p0 && hi < p0 + pn;
            // register offsets of the sample's two rows (columns) inside the patch, or the spare
            const int off_lo = lo_in ? (lo - p0) * stride : SPARE;
            const int off_hi = hi_in ? (hi - p0) * stride : SPARE;
            const float w_lo = 1.0f - fr, w_hi = fr;
            const unsigned long long any = __ballot(lo_in || hi_in);
            const uint32_t row_any = (uint32_t)any & 0xFFFFu, col_any = (uint32_t)(any >> 16) & 0xFFFFu;
            if (row_any == 0u || col_any == 0u) continue;
            const size_t rbase = ((size_t)b * R + (r0 + l)) * (size_t)roi_elems;
            const __amdgpu_buffer_rsrc_t gbuf = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(dY + rbase), 0, roi_elems * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t wbuf = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(winner + rbase), 0, roi_elems, 0x00020000);
#pragma unroll 1
            for (int p = 0; p < ph; ++p) {
                if (((row_any >> (2 * p)) & 3u) == 0u) continue;
                // the bin row's bins whose sample columns touch the patch: all loads go out first
                bp_f4 g[BP_MAXCROP / 2];
                uint32_t wv[BP_MAXCROP / 2];
                const uint32_t rowbin = (uint32_t)(p * pw) * (uint32_t)C;
#pragma unroll
                for (int q = 0; q < BP_MAXCROP / 2; ++q) {
                    g[q] = bp_f4{0.f, 0.f, 0.f, 0.f};
                    wv[q] = 0u;
                    if ((col_any >> (2 * q)) & 3u) {
                        const uint32_t soff = rowbin + (uint32_t)q * (uint32_t)C;   // elements
                        g[q] = __builtin_amdgcn_raw_buffer_load_b128(gbuf, voff_g, soff * 4u, 0);
                        wv[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wbuf, voff_w, soff, 0);
                    }
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int sy = 2 * p + dy;
                    if (((row_any >> sy) & 1u) == 0u) continue;
                    const int r_lo = rl(off_lo, sy), r_hi = rl(off_hi, sy);
                    const float wt = rlf(w_lo, sy), wb = rlf(w_hi, sy);
#pragma unroll
                    for (int sx = 0; sx < BP_MAXCROP; ++sx) {
                        if ((col_any >> sx) & 1u) {           // wave-uniform
                            asm volatile("");                 // keep each sample a uniform branch
                            const int c_lo = rl(off_lo, 16 + sx), c_hi = rl(off_hi, 16 + sx);
                            const float wl = rlf(w_lo, 16 + sx), wr = rlf(w_hi, 16 + sx);
                            bp_sample<NG>(regs, g[sx / 2], wv[sx / 2], kv[dy * 2 + (sx & 1)], wt, wb, wl,
                                          wr, r_lo, r_hi, c_lo, c_hi);
                        }
                    }
                }
            }
        }
    }
    // the patch's cells that exist
    if (c4 < C4) {
#pragma unroll
        for (int cell = 0; cell < PY * PX; ++cell) {
            const int y = y0 + cell / PX, x = x0 + cell % PX;
            if (y < H && x < W) {
                const bp_f16 a = regs.a[cell / 4];
                const int e = (cell % 4) * 4;
                float* dst = dfmap + (((size_t)b * H + y) * W + x) * C + (size_t)c4 * 4;
                *reinterpret_cast<float4*>(dst) = make_float4(a[e], a[e + 1], a[e + 2], a[e + 3]);
            }
        }
    }
}

int launch_roi_bwd_patch(const float* dY, const uint8_t* winner, const float* rois, int B, int R, int H,
                         int W, int C, int crop_h, int crop_w, float* dfmap, hipStream_t stream) {
    if (!winner || crop_h > BP_MAXCROP || crop_w > BP_MAXCROP || (crop_h & 1) || (crop_w & 1) || (C % 4) != 0)
        return FRCNN_E_UNSUPPORTED;
    if ((long long)(crop_h / 2) * (crop_w / 2) * C * 4 >= 0x7FFFFFF0ll) return FRCNN_E_UNSUPPORTED;
    const int nsl = (C / 4 + 63) / 64;
    // patch shape: 4 x 4 cells when the launch fills the chip several times over, 3 x 4 (a third
    // more waves, each shorter) when it does not: B = 64: 8.6 / 9.1 ms, B = 8: 1.67 / 1.56 ms
    // for the two shapes (37 x 62 x 512 map, 2,000 RoIs)
    const long long w44 = (long long)((H + 3) / 4) * ((W + 3) / 4) * B * nsl;
    int shape = w44 >= 8192 ? 44 : 34;
    if (const long long t = tuning(TUNE_BWD_CX); t > 10) shape = (int)t;
    if (shape != 44 && shape != 34 && shape != 46 && shape != 26) return FRCNN_E_BADARG;
    const int PYr = shape / 10, PXr = shape % 10;
    const int npy = (H + PYr - 1) / PYr, npx = (W + PXr - 1) / PXr;
    const long long nwave = (long long)npy * npx * B * nsl;
    if (nwave > 0x7FFFFFF0ll) return FRCNN_E_UNSUPPORTED;
#define FRCNN_BP(PY, PX, WPE)                                                                    \
    FRCNN_LAUNCH((roi_bwd_patch_kernel<PY, PX, WPE>), dim3((unsigned)nwave), dim3(64), 0, stream, \
                 dY, winner, reinterpret_cast<const float4*>(rois), R, H, W, C, crop_h, crop_w,   \
                 npx, npy * npx, B * nsl, nsl, dfmap)
    switch (shape) {
        case 46: FRCNN_BP(4, 6, 2); break;
        case 44: FRCNN_BP(4, 4, 3); break;
        case 34: FRCNN_BP(3, 4, 3); break;
        default: FRCNN_BP(2, 6, 3); break;
    }
#undef FRCNN_BP
    return FRCNN_OK;
}

}  // namespace frcnn
