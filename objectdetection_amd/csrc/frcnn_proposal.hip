// Proposal creator: per-image score top-k (LDS radix select + LDS bitonic
// sort, fused box decode) and greedy NMS (wave-ballot bitmask kernel followed
// by a one-wave scan per image).  gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"
#include <atomic>

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
// Candidates are walked in score order in chunks of NMS_CW column blocks of 64.
// A candidate can only be suppressed by a KEPT box, so for each chunk the mask
// kernel tests the chunk's columns against
//   (a) the boxes kept so far, 64 per "kept tile" (their list is in keep_idx), and
//   (b) the chunk's own candidates (upper triangle of the chunk's tiles),
// one wave per 64x64 tile: lane = row box, the 64 column boxes in LDS, the wave
// ballot of the suppress predicate = the column's mask word (bit r = row r of
// the tile suppresses the column).  Of a kept tile only one word per column block
// is stored (bit c = some row suppresses column c); the in-chunk tiles keep a word
// per column, and as they depend on the boxes only, those of chunk c+1 are computed
// beside the scan of chunk c.  The scan (one workgroup per image)
// then walks the chunk's column blocks: a column is dead if a kept tile's word
// has its bit or an in-chunk word meets that block's keep bits; the block's own
// triangle is resolved serially by one wave.  An image that has post_n boxes
// raises `done` and every later block of it returns at once.
// Pair tests executed: (kept so far) x (columns examined) + the chunk triangles,
// instead of K^2/2; mask storage is three small per-image buffers reused by
// every chunk.
constexpr int NMS_CW = 16;                      // column blocks per chunk
#ifndef NMS_SCAN_WAVES
#define NMS_SCAN_WAVES 4                        // waves of the scan workgroup (1, 2 or 4)
#endif
#ifndef NMS_SCAN_AHEAD
#define NMS_SCAN_AHEAD 1                         // column blocks the scan's loads run ahead (1 or 2).
                                                // 2 is a little faster alone but needs 140 VGPRs:
                                                // beside a kernel whose waves hold 128 each such a
                                                // wave waits for two of them to end on one SIMD
#endif
constexpr int NMS_COLS = NMS_CW * 64;           // columns per chunk

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

// One 64x64 tile.  Per column: a 3-instruction conservative filter
//     inter*(1+thr) > thr*(1-4e-6)*(area_i+area_j)
// (implied by iou > thr with a margin far above float rounding), and only when
// some lane passes it the exact division.
template <bool DIAG>
__device__ __forceinline__ unsigned long long mask_tile(const NBox& a, float c1, const NBox* cols,
                                                        float thr, int lane) {
    unsigned long long mine = 0ull;
    // columns in groups of 8: the group's boxes are read from LDS up front, so the reads'
    // latency overlaps the previous columns' arithmetic instead of preceding each column
#pragma unroll 1
    for (int c0 = 0; c0 < 64; c0 += 8) {
        NBox grp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) grp[u] = cols[c0 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            const NBox cb = grp[u];
            const float iy0 = vmax2(a.y0, cb.y0), ix0 = vmax2(a.x0, cb.x0);
            const float iy1 = vmin2(a.y1, cb.y1), ix1 = vmin2(a.x1, cb.x1);
            // the filter: one clamp is enough (a zero height zeroes the product, a negative width
            // makes it negative: neither passes), the product and the column's term in one fma
            const float fil = vmax2(iy1 - iy0, 0.0f) * (ix1 - ix0);
            unsigned long long w = __ballot(__builtin_fmaf(fil, c1, -cb.t) > a.t);
            if (DIAG) w &= (1ull << c) - 1ull;                   // rows before the column only (lane < c)
            if (w != 0ull) {                                                // uniform
                const float inter = vmax2(iy1 - iy0, 0.0f) * vmax2(ix1 - ix0, 0.0f);
                const float u2 = a.area + cb.area - inter;
                w &= __ballot((a.area > 0.0f) && (cb.area > 0.0f) && (inter / u2 > thr));
            }
            if (lane == c) mine = w;
        }
    }
    return mine;
}

// The same tile when all that matters is WHETHER some row suppresses the column (the rows are
// kept boxes): bit c of the result.  The ballot words never leave the scalar registers.
__device__ __forceinline__ unsigned long long mask_tile_any(const NBox& a, float c1, const NBox* cols,
                                                            float thr) {
    unsigned long long dead = 0ull;
#pragma unroll 1
    for (int c0 = 0; c0 < 64; c0 += 8) {
        NBox grp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) grp[u] = cols[c0 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const NBox cb = grp[u];
            const float iy0 = vmax2(a.y0, cb.y0), ix0 = vmax2(a.x0, cb.x0);
            const float iy1 = vmin2(a.y1, cb.y1), ix1 = vmin2(a.x1, cb.x1);
            const float fil = vmax2(iy1 - iy0, 0.0f) * (ix1 - ix0);
            unsigned long long w = __ballot(__builtin_fmaf(fil, c1, -cb.t) > a.t);
            if (w != 0ull) {                                                // uniform
                const float inter = vmax2(iy1 - iy0, 0.0f) * vmax2(ix1 - ix0, 0.0f);
                const float u2 = a.area + cb.area - inter;
                w = __ballot((a.area > 0.0f) && (cb.area > 0.0f) && (inter / u2 > thr));
                if (w != 0ull) dead |= 1ull << (c0 + u);
            }
        }
    }
    return dead;
}

// The in-chunk tiles of a chunk: column block jbl has the row blocks 0..jbl, and blocks p and
// NMS_CW-1-p together always have NMS_CW+1, so the triangle folds into NMS_CW/2 rows of
// NMS_CW+1 tiles, t = 0 .. NMS_TRI-1, with none left empty.  They depend on the boxes only, not
// on what has been kept: the tiles of chunk c+1 are computed by spare workgroups of the launch
// that scans chunk c (the scan itself keeps one workgroup per image busy and leaves the rest of
// the chip idle), into the other of two maskD buffers.
constexpr int NMS_TRI = (NMS_CW / 2) * (NMS_CW + 1);
static_assert(NMS_CW % 2 == 0, "the triangle folds in pairs of column blocks");

// tile t of the chunk whose first column block is cb0; md = the image's maskD
// [NMS_CW][NMS_COLS] (a word per column: bit r = row r of the row block suppresses it);
// s_col: 64 entries of LDS private to the calling wave
__device__ __forceinline__ void nms_tri_tile(const float4* __restrict__ bx, int K, int cb0, int t,
                                             NBox* s_col, float thr, float c1, float thr_lo,
                                             unsigned long long* __restrict__ md, int lane) {
    const int p = t / (NMS_CW + 1), q = t - p * (NMS_CW + 1);
    int jbl, ibl;
    if (q <= p) { jbl = p; ibl = q; }
    else { jbl = NMS_CW - 1 - p; ibl = q - p - 1; }
    const int j = (cb0 + jbl) * 64 + lane;
    if ((cb0 + jbl) * 64 >= K) return;                  // wave-uniform
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    s_col[lane] = norm_box(j < K ? bx[j] : zero4, thr_lo);
    __builtin_amdgcn_wave_barrier();                    // one wave: its LDS operations stay in order
    const int i = (cb0 + ibl) * 64 + lane;
    const NBox a = norm_box(i < K ? bx[i] : zero4, thr_lo);
    // a tiny row area could underflow the filter's products: +inf makes the filter pass
    // whenever the boxes intersect, and the exact test decides
    const float c1l = a.area < NMS_TINY_AREA ? __builtin_inff() : c1;
    const unsigned long long mine = ibl == jbl ? mask_tile<true>(a, c1l, s_col, thr, lane)
                                               : mask_tile<false>(a, c1l, s_col, thr, lane);
    md[(size_t)ibl * NMS_COLS + jbl * 64 + lane] = mine;
}

// grid (NMS_CW * gk [+ NMS_TRI], 1, B), one wave per workgroup (one-wave workgroups also get
// wave slots while another stream's kernel fills the chip).  NMS_CW * gk workgroups take the
// KEPT row tiles: gk per column block, which share the block's tiles [0,nkt) with stride gk
// (how many boxes are kept is only known on the device).  The first chunk's launch has no kept
// tiles yet and NMS_TRI more workgroups for its own in-chunk tiles.  (One workgroup per
// conceivable tile -- 48 per column block -- left most of them empty whenever few boxes are
// kept, and an empty workgroup still holds a wave slot for the two dependent loads that tell
// it so.)
// Capped at 64 VGPRs: two such waves fit the registers one 128-VGPR wave of a kernel running
// beside it frees.
// deadK [B][NMS_CW]: ONE word per column block, bit c = some kept box suppresses column c:
// every kept tile ORs its word in (the scan hands the words back as zeros);
// maskD [B][NMS_CW][NMS_COLS].
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8)))
nms_mask_kernel(const float4* __restrict__ boxes, int K, int cb0, int post_n, int gk,
                float thr, float c1, float thr_lo, const int* __restrict__ done,
                const int* __restrict__ count_g, const int32_t* __restrict__ keep_idx,
                unsigned long long* __restrict__ deadK, unsigned long long* __restrict__ maskD) {
    const int b = blockIdx.z;
    // the first chunk starts from an empty keep list and never reads count / done: the
    // workspace needs no clearing between calls
    if (cb0 != 0 && done[b]) return;
    const int lane = lane_id();
    __shared__ NBox s_col[64];
    const float4* bx = boxes + (size_t)b * K;
    const int n = (int)blockIdx.x;
    if (n >= NMS_CW * gk) {
        nms_tri_tile(bx, K, cb0, n - NMS_CW * gk, s_col, thr, c1, thr_lo,
                     maskD + (size_t)b * NMS_CW * NMS_COLS, lane);
        return;
    }
    // Workgroup n runs on XCD n % 8: the column block is the fast index, so that the workgroups
    // that find a kept tile when only a few boxes are kept (part 0) spread over every XCD; with
    // part as the fast index they all met on one
    const int part = n / NMS_CW, jbl = n - part * NMS_CW;
    const int jb = cb0 + jbl;
    if (jb * 64 >= K) return;
    const int count = cb0 != 0 ? count_g[b] : 0;        // boxes kept before this chunk
    const int nkt = (count + 63) >> 6;
    if (part >= nkt) return;                            // no kept tile for this workgroup
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int j = jb * 64 + lane;
        s_col[lane] = norm_box(j < K ? bx[j] : zero4, thr_lo);
    }
    __builtin_amdgcn_wave_barrier();                    // one wave: its LDS operations stay in order
#pragma unroll 1
    for (int rt = part; rt < nkt; rt += gk) {
        const int slot = rt * 64 + lane;
        const NBox a = norm_box(slot < count ? bx[keep_idx[(size_t)b * post_n + slot]] : zero4, thr_lo);
        const float c1l = a.area < NMS_TINY_AREA ? __builtin_inff() : c1;
        const unsigned long long dead = mask_tile_any(a, c1l, s_col, thr);
        if (lane == 0 && dead != 0ull) atomicOr(&deadK[(size_t)b * NMS_CW + jbl], dead);
    }
}

// Workgroups 0 .. nimg-1 (NMS_SCAN_WAVES waves each): one per image walks the chunk's column
// blocks in order.  With maskD_next, nimg * NMS_TRI_WGS more: wave = one in-chunk tile of the
// NEXT chunk (nms_tri_tile), which needs nothing from this scan.
constexpr int NMS_TRI_WGS = (NMS_TRI + NMS_SCAN_WAVES - 1) / NMS_SCAN_WAVES;

__global__ void __launch_bounds__(64 * NMS_SCAN_WAVES)
nms_scan_kernel(const float4* __restrict__ boxes, unsigned long long* __restrict__ deadK,
                const unsigned long long* __restrict__ maskD,
                unsigned long long* __restrict__ maskD_next, int nimg, int K, int cb0,
                int post_n, int last_chunk, float thr, float c1, float thr_lo,
                int* __restrict__ count_g, int* __restrict__ done,
                int32_t* __restrict__ keep_idx, float4* __restrict__ proposed,
                int32_t* __restrict__ keep_pos, int32_t* __restrict__ keep_count) {
    __shared__ unsigned long long keptw[NMS_CW];        // keep bits of this chunk's blocks
    __shared__ unsigned long long deadw[NMS_SCAN_WAVES];
    __shared__ unsigned long long deadk_s[NMS_CW];      // columns the kept boxes suppress, per block
    __shared__ int s_cnt;
    __shared__ NBox s_tri[NMS_SCAN_WAVES][64];
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    if ((int)blockIdx.x >= nimg) {
        const int n = (int)blockIdx.x - nimg, bt = n / NMS_TRI_WGS;
        const int t = (n - bt * NMS_TRI_WGS) * NMS_SCAN_WAVES + wave;
        // an image finished by an earlier chunk has no next chunk (the first chunk's launch
        // finds `done` as the previous call left it and does not look; an image that this
        // very launch finishes may or may not be seen as done: its tiles are never read)
        if (t >= NMS_TRI || (cb0 != 0 && done[bt])) return;
        nms_tri_tile(boxes + (size_t)bt * K, K, cb0 + NMS_CW, t, s_tri[wave], thr, c1, thr_lo,
                     maskD_next + (size_t)bt * NMS_CW * NMS_COLS, lane);
        return;
    }
    const int b = blockIdx.x;
    if (cb0 != 0 && done[b]) return;                    // block-uniform
    // the walk is one dependent chain: its waves go first where they share a CU with tile waves
    __builtin_amdgcn_s_setprio(3);
    const float4* bx = boxes + (size_t)b * K;
    const unsigned long long* md = maskD + (size_t)b * NMS_CW * NMS_COLS;
    float4* prop = proposed + (size_t)b * post_n;
    int32_t* kidx = keep_idx + (size_t)b * post_n;
    int32_t* kp = keep_pos ? keep_pos + (size_t)b * post_n : nullptr;
    const int count0 = cb0 != 0 ? count_g[b] : 0;
    if (tid == 0) s_cnt = count0;
    // the kept tiles' words of this chunk, handed back as zeros for the next chunk's tiles (the
    // first chunk has no kept tiles: whatever an earlier call left there is dropped)
    if (tid < NMS_CW) {
        unsigned long long* dk = deadK + (size_t)b * NMS_CW + tid;
        const unsigned long long v = *dk;
        *dk = 0ull;
        deadk_s[tid] = cb0 != 0 ? v : 0ull;
    }
    __syncthreads();
    const int nblk = min(NMS_CW, (K + 63) / 64 - cb0);
    // Everything a column block needs from memory is independent of the scan's own results
    // (the raw in-chunk words, the diagonal word, the box), so the loads of a later block are
    // issued before block jbl is resolved and the serial part sees no load latency.  A wave
    // owns the in-chunk rows wave, wave+SW, ...; rows at or beyond the block itself have
    // no tile: their words are read (whatever they hold) and dropped, which keeps branches out
    // of the walk.
    constexpr int SW = NMS_SCAN_WAVES;
    struct Pre { unsigned long long diag, mdw[NMS_CW / SW]; float4 box; };
    auto prefetch = [&](int jbl) {
        Pre p;
        const int col = jbl * 64 + lane;
        p.diag = md[(size_t)jbl * NMS_COLS + col];
#pragma unroll
        for (int t = 0; t < NMS_CW / SW; ++t) {
            const int ibl = wave + SW * t;
            p.mdw[t] = md[(size_t)ibl * NMS_COLS + col];
        }
        const int j = (cb0 + jbl) * 64 + lane;
        p.box = bx[j < K ? j : K - 1];
        return p;
    };
    Pre cur = prefetch(0);
#if NMS_SCAN_AHEAD == 2
    Pre nxt = cur;
    if (1 < nblk) nxt = prefetch(1);
#endif
#pragma unroll 1
    for (int jbl = 0; jbl < nblk; ++jbl) {
        const int count = s_cnt;
        if (count >= post_n) break;                     // block-uniform
#if NMS_SCAN_AHEAD == 2
        Pre nn = nxt;
        if (jbl + 2 < nblk) nn = prefetch(jbl + 2);     // block-uniform, two blocks ahead
#else
        Pre nxt = cur;
        if (jbl + 1 < nblk) nxt = prefetch(jbl + 1);    // block-uniform, one block ahead
#endif
        const int j0 = (cb0 + jbl) * 64;
        unsigned long long hit = 0ull;
#pragma unroll
        for (int t = 0; t < NMS_CW / SW; ++t) {
            const int ibl = wave + SW * t;
            if (ibl < jbl) hit |= cur.mdw[t] & keptw[ibl];
        }
        const unsigned long long diag = cur.diag;
        const unsigned long long dead = __ballot(hit != 0ull);
        if (lane == 0) deadw[wave] = dead;
        __syncthreads();
        if (wave == 0) {
            const int nvalid = K - j0 < 64 ? K - j0 : 64;
            const unsigned long long vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
            // Greedy order inside the block, resolved by relaxation instead of one kept box at
            // a time: lane c's word `diag` lists the rows r < c that would suppress column c.
            // A column whose suppressors are all dead is kept; one with a kept suppressor is
            // dead; the lowest undecided column is always decidable, so the loop ends, and its
            // fixed point is the greedy result.  Typical depth: a handful of rounds.
            unsigned long long any_dead = deadk_s[jbl];
#pragma unroll
            for (int w = 0; w < SW; ++w) any_dead |= deadw[w];
            unsigned long long gone = uniform64(any_dead | ~vmask);
            unsigned long long keptmask = 0ull;
            const unsigned long long me = 1ull << lane;
            while (~(gone | keptmask) != 0ull) {        // wave-uniform
                const bool und = ((gone | keptmask) & me) == 0ull;
                const unsigned long long k2 = __ballot(und && (diag & ~gone) == 0ull);
                const unsigned long long g2 = __ballot(und && (diag & keptmask) != 0ull);
                keptmask |= k2;
                gone |= g2;
            }
            // only the first `room` kept boxes fit
            const int room = post_n - count;
            keptmask = __ballot(((keptmask & me) != 0ull) && __popcll(keptmask & lanemask_lt()) < room);
            if ((keptmask >> lane) & 1ull) {
                const int slot = count + __popcll(keptmask & lanemask_lt());
                float4 v = cur.box;
                v.x = clip01(v.x); v.y = clip01(v.y); v.z = clip01(v.z); v.w = clip01(v.w);
                prop[slot] = v;
                kidx[slot] = j0 + lane;
                if (kp) kp[slot] = j0 + lane;
            }
            if (lane == 0) {
                keptw[jbl] = keptmask;
                s_cnt = count + __popcll(keptmask);
            }
        }
        __syncthreads();
        cur = nxt;
#if NMS_SCAN_AHEAD == 2
        nxt = nn;
#endif
    }
    const int count = s_cnt;
    const bool finished = count >= post_n || last_chunk;
    if (finished) {
        // zero padding (tf.image.combined_non_max_suppression pads boxes with 0)
        for (int s = count + tid; s < post_n; s += 64 * NMS_SCAN_WAVES) {
            prop[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kp) kp[s] = -1;
        }
        if (tid == 0 && keep_count) keep_count[b] = count;
    }
    if (tid == 0) {
        done[b] = finished ? 1 : 0;
        count_g[b] = count;
    }
}


// =========================================================================
// NMS in ONE launch, for a few images (latency)
// =========================================================================
// The chunked formulation above costs two dependent launches per 1,024 candidates and a serial
// walk of 16 column blocks per chunk: for one image that chain is the whole cost (24 launches,
// ~190 blocks at > 1 us each).  Here one persistent grid does it all:
//   * workgroups 0 .. B-1 are RESOLVERS, one per image: thread = one column of the current chunk
//     (wave = column block).  A column's suppressors inside the chunk (maskT: in-chunk triangle)
//     and in the chunk before (maskR: a full 16 x 16 rectangle of tiles) are mask words that
//     depend on the boxes only; columns the boxes kept in still earlier chunks suppress arrive as
//     one word per column block (deadK).  The chunk is then resolved by relaxation over all 1,024
//     columns at once: a column whose suppressors are all gone is kept, one with a kept
//     suppressor is gone -- both are final facts, so stale reads of other waves' words only delay
//     a decision, and the fixed point is the greedy result (at most 16 rounds: a column block is
//     final once the blocks before it are).
//   * the other workgroups are WORKERS: stage s = every tile that chunk s needs (triangle,
//     rectangle against chunk s-1, kept tiles against what was kept up to chunk s-2), written
//     into buffers of parity s & 1.  Stage s may start once chunk s-2 is resolved, so it runs
//     BESIDE the resolution of chunk s-1: the resolver's chain is relaxation after relaxation,
//     the tiles arrive off its critical path.
// Hand-off is by counters in the workspace (agent-scope release / acquire: each XCD has its own
// L2).  Work is assigned statically to worker waves, so progress needs every workgroup of the
// grid to become resident: the grid is at most 256 workgroups (one per compute unit), none of
// them waits for anything but the counters, and the resolvers are dispatched first.
constexpr int NF_THREADS = 1024;
constexpr int NF_WAVES = NF_THREADS / 64;
static_assert(NF_WAVES == NMS_CW, "the resolver gives every column block of a chunk one wave");
constexpr int NF_MAX_B = 8;
constexpr int NF_DONE = 0x7FFFFFFF;
constexpr int NF_RECT = NMS_CW * NMS_CW;

struct NfImage { int scan_done; int abort; int pad[2]; };   // chunks resolved (NF_DONE: finished)

struct NfWorkspace {
    NfImage* img;                   // [B]
    int* stage_done;                // [B][nchunk] worker workgroups that finished stage s
    int* count_after;               // [B][nchunk] boxes kept after chunk c
    unsigned long long* deadK;      // [B][2][NMS_CW]
    unsigned long long* maskT;      // [B][2][NMS_CW][NMS_COLS]
    unsigned long long* maskR;      // [B][2][NMS_CW][NMS_COLS]
    unsigned long long* sumT;       // [B][2][NMS_CW][NMS_CW] bit l: column l of the tile has a non-zero word
    unsigned long long* sumR;       // [B][2][NMS_CW][NMS_CW]
    int32_t* keep_idx;              // [B][post_n]
    unsigned long long* trace;      // timing experiments only (FRCNN_NF_TRACE builds)
};

#ifdef FRCNN_NF_TRACE
#define NF_TRACE(slot, extra)                                                                   \
    do { if (tid == 0) ws.trace[(slot)] = (__builtin_amdgcn_s_memrealtime() << 8) | (unsigned long long)(extra); } while (0)
#else
#define NF_TRACE(slot, extra) do { } while (0)
#endif

// Everything one XCD writes for another to read inside this launch goes through agent-scope
// atomic loads and stores (they are served at the coherent point, past the per-XCD L2s), ordered
// by "all my stores have completed" (vmcnt) before the counter that announces them.  Release /
// acquire fences would do the same by writing back and invalidating the whole L2 of an XCD: with
// hundreds of workgroups fencing per stage, no load of anyone hit a cache any more.
__device__ __forceinline__ int nf_load(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void nf_store(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long nf_load64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void nf_store64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void nf_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Polls *p until it reaches `want`.  Every wait in the grid is bounded: after about a second
// without progress (a grid that cannot become resident as a whole) the waiter raises the abort
// word of image 0, which every other wait also watches, and the grid drains; the call then
// reports keep_count = -1.  Returns the value seen, or -1 on abort.
constexpr int NF_SPINS = 1 << 21;            // x ~0.5 us
__device__ __forceinline__ int nf_wait(const int* p, int want, int* abort_word) {
    for (int i = 0; i < NF_SPINS; ++i) {
        const int v = nf_load(p);
        if (v >= want) return v;
        if ((i & 63) == 63 && nf_load(abort_word) != 0) return -1;
        __builtin_amdgcn_s_sleep(16);               // ~0.5 us: hundreds of waiters poll one word
    }
    __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return -1;
}

__global__ void nms_fused_init_kernel(int* words, int n, int* more, int m) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) words[i] = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) more[i] = 0;
}

// tile t = (row block ibl of chunk s-1, column block jbl of chunk s): a full 64 x 64 tile
__device__ __forceinline__ void nms_rect_tile(const float4* __restrict__ bx, int K, int cb0, int t,
                                              NBox* s_col, float thr, float c1, float thr_lo,
                                              unsigned long long* __restrict__ mr,
                                              unsigned long long* __restrict__ sum, int lane) {
    const int ibl = t / NMS_CW, jbl = t - ibl * NMS_CW;
    if ((cb0 + jbl) * 64 >= K) return;                  // wave-uniform
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = (cb0 + jbl) * 64 + lane;
    s_col[lane] = norm_box(j < K ? bx[j] : zero4, thr_lo);
    __builtin_amdgcn_wave_barrier();
    const int i = (cb0 - NMS_CW + ibl) * 64 + lane;     // always < K: the chunk before is complete
    const NBox a = norm_box(bx[i], thr_lo);
    const float c1l = a.area < NMS_TINY_AREA ? __builtin_inff() : c1;
    const unsigned long long mine = mask_tile<false>(a, c1l, s_col, thr, lane);
    // most columns have no suppressor in a given row block: only non-zero words are stored, and
    // a word per tile says which (the resolver fetches nothing else)
    const unsigned long long nz = __ballot(mine != 0ull);
    if (mine != 0ull) nf_store64(&mr[(size_t)ibl * NMS_COLS + jbl * 64 + lane], mine);
    if (lane == 0) nf_store64(&sum[ibl * NMS_CW + jbl], nz);
    __builtin_amdgcn_wave_barrier();                    // the next tile of this wave rewrites s_col
}

// nms_tri_tile for the one-launch kernel: the same tile, stored sparsely as above
__device__ __forceinline__ void nms_tri_tile_sparse(const float4* __restrict__ bx, int K, int cb0, int t,
                                                    NBox* s_col, float thr, float c1, float thr_lo,
                                                    unsigned long long* __restrict__ md,
                                                    unsigned long long* __restrict__ sum, int lane) {
    const int p = t / (NMS_CW + 1), q = t - p * (NMS_CW + 1);
    int jbl, ibl;
    if (q <= p) { jbl = p; ibl = q; }
    else { jbl = NMS_CW - 1 - p; ibl = q - p - 1; }
    if ((cb0 + jbl) * 64 >= K) return;                  // wave-uniform
    const int j = (cb0 + jbl) * 64 + lane;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    s_col[lane] = norm_box(j < K ? bx[j] : zero4, thr_lo);
    __builtin_amdgcn_wave_barrier();
    const int i = (cb0 + ibl) * 64 + lane;
    const NBox a = norm_box(i < K ? bx[i] : zero4, thr_lo);
    const float c1l = a.area < NMS_TINY_AREA ? __builtin_inff() : c1;
    const unsigned long long mine = ibl == jbl ? mask_tile<true>(a, c1l, s_col, thr, lane)
                                               : mask_tile<false>(a, c1l, s_col, thr, lane);
    const unsigned long long nz = __ballot(mine != 0ull);
    if (mine != 0ull) nf_store64(&md[(size_t)ibl * NMS_COLS + jbl * 64 + lane], mine);
    if (lane == 0) nf_store64(&sum[ibl * NMS_CW + jbl], nz);
    __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(NF_THREADS)
nms_fused_kernel(const float4* __restrict__ boxes, int B, int K, int post_n, int nchunk, int NW,
                 float thr, float c1, float thr_lo, NfWorkspace ws,
                 float4* __restrict__ proposed, int32_t* __restrict__ keep_pos,
                 int32_t* __restrict__ keep_count) {
    __shared__ NBox s_col[NF_WAVES][64];
    __shared__ unsigned long long kept_w[NMS_CW], gone_w[NMS_CW], prevk[NMS_CW];
    __shared__ int und_w[2][NMS_CW];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= B) {
        // ------------------------------------------------------------------ worker
        // jobs are dealt to the workgroups first and to the waves of a workgroup second: a stage
        // has fewer tiles than the grid has waves, and sixteen busy waves on one compute unit
        // run at a quarter of the speed of four on each of four
        const int TW = NW * NF_WAVES;
        const int gw0 = wave * NW + ((int)blockIdx.x - B);
        for (int s = 0; s < nchunk; ++s) {
            const int cb0 = s * NMS_CW;
            for (int b = 0; b < B; ++b) {
                NfImage* im = ws.img + b;
                // stage s rewrites the buffers chunk s-2 was resolved from and needs what was
                // kept up to chunk s-2
                if ((int)blockIdx.x == B && b == 0) NF_TRACE(1024 + s * 4 + 0, 0);
                if (tid == 0) s_flag = nf_wait(&im->scan_done, s - 1, &ws.img[0].abort);
                __syncthreads();
                if ((int)blockIdx.x == B && b == 0) NF_TRACE(1024 + s * 4 + 1, 0);
                const int sd = s_flag;
                __syncthreads();                        // s_flag is rewritten for the next image
                if (sd < 0) return;                     // workgroup-uniform: the grid is draining
                if (sd == NF_DONE) continue;            // workgroup-uniform: the image is finished
#ifdef FRCNN_NF_TRACE
                if ((int)blockIdx.x == B && b == 0) NF_TRACE(1536 + s * 4 + 0, 0);
#endif
                const float4* bx = boxes + (size_t)b * K;
                const int par = s & 1;
                unsigned long long* dk = ws.deadK + ((size_t)b * 2 + par) * NMS_CW;
                unsigned long long* mt = ws.maskT + ((size_t)b * 2 + par) * NMS_CW * NMS_COLS;
                unsigned long long* mr = ws.maskR + ((size_t)b * 2 + par) * NMS_CW * NMS_COLS;
                unsigned long long* st_ = ws.sumT + ((size_t)b * 2 + par) * NMS_CW * NMS_CW;
                unsigned long long* sr_ = ws.sumR + ((size_t)b * 2 + par) * NMS_CW * NMS_CW;
                const int count = s >= 2 ? nf_load(&ws.count_after[(size_t)b * nchunk + s - 2]) : 0;
                const int nkt = (count + 63) >> 6;
                const int nkept = nkt * NMS_CW, nrect = s >= 1 ? NF_RECT : 0;
                const int njobs = nkept + nrect + NMS_TRI;
                const int* kidx = ws.keep_idx + (size_t)b * post_n;
                const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                int gw = gw0 + b * 131;
                gw -= (gw / TW) * TW;
#pragma unroll 1
                for (int j = gw; j < njobs; j += TW) {                 // wave-uniform
                    if (j < nkept) {
                        const int kt = j / NMS_CW, jbl = j - kt * NMS_CW;
                        const int jb = cb0 + jbl;
                        if (jb * 64 >= K) continue;
                        const int col = jb * 64 + lane;
                        float4 cbx = bx[col < K ? col : K - 1];
                        if (col >= K) cbx = zero4;
                        s_col[wave][lane] = norm_box(cbx, thr_lo);
                        __builtin_amdgcn_wave_barrier();
                        const int slot = kt * 64 + lane;
                        float4 rbx = bx[slot < count ? nf_load(&kidx[slot]) : 0];
                        if (slot >= count) rbx = zero4;
                        const NBox a = norm_box(rbx, thr_lo);
                        const float c1l = a.area < NMS_TINY_AREA ? __builtin_inff() : c1;
                        const unsigned long long dead = mask_tile_any(a, c1l, s_col[wave], thr);
                        if (lane == 0 && dead != 0ull)
                            __hip_atomic_fetch_or(&dk[jbl], dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_amdgcn_wave_barrier();
                    } else if (j < nkept + nrect) {
                        nms_rect_tile(bx, K, cb0, j - nkept, s_col[wave], thr, c1, thr_lo, mr, sr_, lane);
                    } else {
#ifdef FRCNN_NF_TRACE
                        {
                            const float4 probe = bx[(cb0 * 64 + lane) < K ? cb0 * 64 + lane : 0];
                            asm volatile("s_waitcnt vmcnt(0)" :: "v"(probe.x));
                            if ((int)blockIdx.x == B && b == 0) NF_TRACE(1536 + s * 4 + 1, 0);
                        }
#endif
                        nms_tri_tile_sparse(bx, K, cb0, j - nkept - nrect, s_col[wave], thr, c1, thr_lo, mt,
                                            st_, lane);
#ifdef FRCNN_NF_TRACE
                        asm volatile("s_waitcnt vmcnt(0)");
                        if ((int)blockIdx.x == B && b == 0) NF_TRACE(1536 + s * 4 + 2, 0);
#endif
                    }
                }
                nf_stores_done();
                __syncthreads();                        // every wave's stores have completed
                if ((int)blockIdx.x == B && b == 0) NF_TRACE(1024 + s * 4 + 2, njobs);
                if (tid == 0) {
                    __hip_atomic_fetch_add(&ws.stage_done[(size_t)b * nchunk + s], 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((int)blockIdx.x == B && b == 0) NF_TRACE(1024 + s * 4 + 3, 0);
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- resolver
    const int b = blockIdx.x;
    const float4* bx = boxes + (size_t)b * K;
    float4* prop = proposed + (size_t)b * post_n;
    int* kidx = ws.keep_idx + (size_t)b * post_n;
    int32_t* kp = keep_pos ? keep_pos + (size_t)b * post_n : nullptr;
    NfImage* im = ws.img + b;
    __builtin_amdgcn_s_setprio(3);
    int count = 0;
    if (tid < NMS_CW) prevk[tid] = 0ull;
#ifdef FRCNN_NF_TRACE
    if (tid == 0 && b == 0) { ws.trace[2040] = __builtin_amdgcn_s_memrealtime(); ws.trace[2041] = __builtin_amdgcn_s_memtime(); }
#endif
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        if (b == 0) NF_TRACE(c * 8 + 0, 0);
        if (tid == 0) s_flag = nf_wait(&ws.stage_done[(size_t)b * nchunk + c], NW, &ws.img[0].abort);
        __syncthreads();
        if (b == 0) NF_TRACE(c * 8 + 1, 0);
        if (s_flag < 0) {                               // workgroup-uniform: give up, tell the caller
            if (tid == 0) {
                if (keep_count) keep_count[b] = -1;
                nf_store(&im->scan_done, NF_DONE);
            }
            return;
        }
        const int par = c & 1;
        const int cb0 = c * NMS_CW;
        const int colb = wave * 64 + lane;              // column inside the chunk
        const int col = cb0 * 64 + colb;
        const bool valid = col < K;
        const unsigned long long* mt = ws.maskT + ((size_t)b * 2 + par) * NMS_CW * NMS_COLS;
        const unsigned long long* mr = ws.maskR + ((size_t)b * 2 + par) * NMS_CW * NMS_COLS;
        unsigned long long* dkp = ws.deadK + ((size_t)b * 2 + par) * NMS_CW;
        // suppressor words: rows of earlier blocks of this chunk (T), of this block (diag), of the
        // chunk before (R).  Words of nonexistent columns are never written and never used.
        const unsigned long long* st_ = ws.sumT + ((size_t)b * 2 + par) * NMS_CW * NMS_CW;
        const unsigned long long* sr_ = ws.sumR + ((size_t)b * 2 + par) * NMS_CW * NMS_CW;
        // lane i < 16: the summary words of (row block i, this column block)
        const bool blk = (cb0 + wave) * 64 < K;         // wave-uniform: the column block exists
        const unsigned long long sT = (lane <= wave && blk) ? nf_load64(&st_[lane * NMS_CW + wave]) : 0ull;
        const unsigned long long sR = (lane < NMS_CW && c >= 1 && blk) ? nf_load64(&sr_[lane * NMS_CW + wave]) : 0ull;
        unsigned long long T[NMS_CW], R[NMS_CW];
        unsigned long long diag = 0ull;
#pragma unroll
        for (int i = 0; i < NMS_CW; ++i) {
            const unsigned long long nzT = uniform64(__shfl(sT, i)), nzR = uniform64(__shfl(sR, i));
            T[i] = 0ull; R[i] = 0ull;
            if (i < wave) { if ((nzT >> lane) & 1ull) T[i] = nf_load64(&mt[(size_t)i * NMS_COLS + colb]); }
            else if (i == wave) { if ((nzT >> lane) & 1ull) diag = nf_load64(&mt[(size_t)i * NMS_COLS + colb]); }
            if ((nzR >> lane) & 1ull) R[i] = nf_load64(&mr[(size_t)i * NMS_COLS + colb]);
        }
        const unsigned long long dk = c >= 2 ? uniform64(nf_load64(&dkp[wave])) : 0ull;
        const float4 box = bx[valid ? col : K - 1];
        bool gone0 = !valid || ((dk >> lane) & 1ull);
        {
            unsigned long long h = 0ull;
#pragma unroll
            for (int i = 0; i < NMS_CW; ++i) h |= R[i] & prevk[i];
            gone0 = gone0 || h != 0ull;
        }
        unsigned long long my_gone = __ballot(gone0), my_kept = 0ull;
        if (b == 0) NF_TRACE(c * 8 + 2, 0);
        if (lane == 0) {
            gone_w[wave] = my_gone;
            kept_w[wave] = 0ull;
            nf_store64(&dkp[wave], 0ull);               // handed back as zeros for stage c + 2
        }
        __syncthreads();
        const unsigned long long me = 1ull << lane;
#pragma unroll 1
        for (int r = 0;; ++r) {
            unsigned long long hit = 0ull, alive = 0ull;
            const bool open = ~(my_kept | my_gone) != 0ull;         // wave-uniform
            if (open) {
#pragma unroll
                for (int i = 0; i < NMS_CW; ++i) {
                    if (i < wave) {                     // wave-uniform
                        const unsigned long long kw = kept_w[i], gw_ = gone_w[i];
                        hit |= T[i] & kw;
                        alive |= T[i] & ~gw_;
                    }
                }
            }
            // this block's own triangle to a local fixed point with what is known of the others
            while (open) {
                const bool und = ((my_gone | my_kept) & me) == 0ull;
                const unsigned long long k2 = __ballot(und && alive == 0ull && (diag & ~my_gone) == 0ull);
                const unsigned long long g2 = __ballot(und && (hit != 0ull || (diag & my_kept) != 0ull));
                if ((k2 | g2) == 0ull) break;           // wave-uniform
                my_kept |= k2;
                my_gone |= g2;
            }
            if (lane == 0) {
                kept_w[wave] = my_kept;
                gone_w[wave] = my_gone;
                und_w[r & 1][wave] = ~(my_kept | my_gone) != 0ull ? 1 : 0;
            }
            __syncthreads();
            const int u = lane < NMS_CW ? und_w[r & 1][lane] : 0;
            if (__ballot(u != 0) == 0ull) { if (b == 0) NF_TRACE(c * 8 + 3, r + 1); break; }   // uniform
        }
        // kept columns in order, as far as there is room
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < NMS_CW; ++i) {
            const int n = __popcll(kept_w[i]);
            if (i < wave) before += n;
            total += n;
        }
        const int slot = count + before + __popcll(my_kept & lanemask_lt());
        const bool keep = (my_kept & me) != 0ull && slot < post_n;
        const unsigned long long capped = __ballot(keep);
        if (keep) {
            float4 v = box;
            v.x = clip01(v.x); v.y = clip01(v.y); v.z = clip01(v.z); v.w = clip01(v.w);
            prop[slot] = v;
            nf_store(&kidx[slot], col);
            if (kp) kp[slot] = col;
        }
        __syncthreads();                                // every wave has read kept_w / prevk
        if (lane == 0) prevk[wave] = capped;
        const int newcount = min(count + total, post_n);
        const bool finished = newcount >= post_n || c == nchunk - 1;
        if (finished) {
            for (int sl = newcount + tid; sl < post_n; sl += NF_THREADS) {
                prop[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kp) kp[sl] = -1;
            }
            if (tid == 0 && keep_count) keep_count[b] = newcount;
        }
        if (tid == 0) nf_store(&ws.count_after[(size_t)b * nchunk + c], newcount);
        nf_stores_done();
        __syncthreads();                                // all stores of this chunk have completed
        if (tid == 0) nf_store(&im->scan_done, finished ? NF_DONE : c + 1);
        if (b == 0) NF_TRACE(c * 8 + 4, newcount);
#ifdef FRCNN_NF_TRACE
        if (tid == 0 && b == 0) { ws.trace[2042] = __builtin_amdgcn_s_memrealtime(); ws.trace[2043] = __builtin_amdgcn_s_memtime(); }
#endif
        count = newcount;
        if (finished) break;
    }
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

// workspace = deadK [B][NMS_CW] u64 | maskD [2][B][NMS_CW][NMS_COLS] u64 |
//             keep_idx [B][post_n] i32 | count [B] i32 | done [B] i32
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t nms_deadk_bytes(int B) { return align256((size_t)B * NMS_CW * 8); }
static size_t nms_maskd_bytes(int B) { return align256((size_t)B * NMS_CW * NMS_COLS * 8); }

extern "C" size_t frcnn_nms_workspace_bytes(int B, int K, int post_n) {
    if (B <= 0 || K <= 0 || post_n <= 0) return 0;
    return nms_deadk_bytes(B) + 2 * nms_maskd_bytes(B) + align256((size_t)B * post_n * 4) +
           align256((size_t)B * 8);
}

// the chunk loop both entry points share; `ws` holds frcnn_nms_workspace_bytes(B,K,post_n) bytes
static int nms_launch(const float* boxes, int B, int K, float iou_threshold, int post_n,
                      float* proposed, int32_t* keep_pos, int32_t* keep_count, char* ws,
                      hipStream_t st) {
    const int CB = (K + 63) / 64;
    const int KT = (post_n + 63) / 64;
    if (B > 65535) return FRCNN_E_UNSUPPORTED;
    unsigned long long* deadK = reinterpret_cast<unsigned long long*>(ws);
    ws += nms_deadk_bytes(B);
    unsigned long long* maskD[2];
    maskD[0] = reinterpret_cast<unsigned long long*>(ws);
    ws += nms_maskd_bytes(B);
    maskD[1] = reinterpret_cast<unsigned long long*>(ws);
    ws += nms_maskd_bytes(B);
    int32_t* keep_idx = reinterpret_cast<int32_t*>(ws);
    ws += align256((size_t)B * post_n * 4);
    int* count = reinterpret_cast<int*>(ws);
    int* done = count + B;
    // filter constants: iou > thr  =>  inter*(1+thr) > thr*(1-4e-6)*(area_i+area_j)
    const float c1 = 1.0f + iou_threshold;
    const float thr_lo = iou_threshold * 0.999996f;
    // workgroups per column block for the kept row tiles: one per tile with few images (every
    // tile in parallel), fewer -- each taking several tiles in turn -- with many
    int gk = (16384 + NMS_CW * B - 1) / (NMS_CW * B);
    if (const long long t = tuning(TUNE_NMS_GK); t != TUNE_UNSET && t > 0) gk = (int)(t < 65536 ? t : 65536);
    gk = gk < 1 ? 1 : gk > KT ? KT : gk;
    if ((long long)B * (1 + NMS_TRI_WGS) > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    for (int cb0 = 0, ci = 0; cb0 < CB; cb0 += NMS_CW, ++ci) {
        const int last = cb0 + NMS_CW >= CB;
        // chunk ci's in-chunk tiles are in maskD[ci & 1]: written by this launch for the first
        // chunk, by the spare workgroups of the previous chunk's scan for the others
        FRCNN_LAUNCH(nms_mask_kernel, dim3(NMS_CW * gk + (ci == 0 ? NMS_TRI : 0), 1, B), dim3(64), 0,
                     st, reinterpret_cast<const float4*>(boxes), K, cb0, post_n, gk,
                     iou_threshold, c1, thr_lo, done, count, keep_idx, deadK, maskD[ci & 1]);
        FRCNN_LAUNCH(nms_scan_kernel, dim3(last ? B : B * (1 + NMS_TRI_WGS)),
                     dim3(64 * NMS_SCAN_WAVES), 0, st, reinterpret_cast<const float4*>(boxes), deadK,
                     maskD[ci & 1], last ? nullptr : maskD[(ci + 1) & 1], B, K, cb0, post_n, last,
                     iou_threshold, c1, thr_lo, count, done, keep_idx,
                     reinterpret_cast<float4*>(proposed), keep_pos, keep_count);
    }
    return FRCNN_OK;
}

extern "C" int frcnn_nms(const float* boxes, int B, int K, float iou_threshold, int post_n,
                         float* proposed, int32_t* keep_pos, int32_t* keep_count, void* workspace,
                         size_t workspace_bytes, frcnn_stream_t stream) {
    if (!boxes || !proposed || B <= 0 || K <= 0 || post_n <= 0) return FRCNN_E_BADARG;
    if (!(iou_threshold >= 0.0f)) return FRCNN_E_BADARG;
    if (!workspace || workspace_bytes < frcnn_nms_workspace_bytes(B, K, post_n))
        return FRCNN_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 7u) != 0) return FRCNN_E_BADARG;
    return nms_launch(boxes, B, K, iou_threshold, post_n, proposed, keep_pos, keep_count,
                      reinterpret_cast<char*>(workspace), (hipStream_t)stream);
}


// ---- the one-launch variant for a few images --------------------------------------------------
static int nf_nchunk(int K) { return ((K + 63) / 64 + NMS_CW - 1) / NMS_CW; }
static size_t nf_ints(int B, int K) { return (size_t)B * (4 + 2 * (size_t)nf_nchunk(K)); }

extern "C" size_t frcnn_nms_fused_workspace_bytes(int B, int K, int post_n) {
    if (B <= 0 || K <= 0 || post_n <= 0) return 0;
    size_t n = align256(nf_ints(B, K) * 4) + align256((size_t)B * 2 * NMS_CW * 8) +
               2 * align256((size_t)B * 2 * NMS_CW * NMS_COLS * 8) +
               2 * align256((size_t)B * 2 * NMS_CW * NMS_CW * 8) + align256((size_t)B * post_n * 4);
#ifdef FRCNN_NF_TRACE
    n += 16384;                                 // 2,048 time stamps behind the kept index list
#endif
    return n;
}

extern "C" int frcnn_nms_fused(const float* boxes, int B, int K, float iou_threshold, int post_n,
                               float* proposed, int32_t* keep_pos, int32_t* keep_count,
                               void* workspace, size_t workspace_bytes, frcnn_stream_t stream) {
    if (!boxes || !proposed || B <= 0 || K <= 0 || post_n <= 0) return FRCNN_E_BADARG;
    if (!(iou_threshold >= 0.0f)) return FRCNN_E_BADARG;
    if (B > NF_MAX_B) return FRCNN_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < frcnn_nms_fused_workspace_bytes(B, K, post_n))
        return FRCNN_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return FRCNN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = nf_nchunk(K);
    char* p = reinterpret_cast<char*>(workspace);
    NfWorkspace ws;
    int* ints = reinterpret_cast<int*>(p);
    ws.img = reinterpret_cast<NfImage*>(ints);
    ws.stage_done = ints + 4 * B;
    ws.count_after = ws.stage_done + (size_t)B * nchunk;
    p += align256(nf_ints(B, K) * 4);
    ws.deadK = reinterpret_cast<unsigned long long*>(p);
    p += align256((size_t)B * 2 * NMS_CW * 8);
    ws.maskT = reinterpret_cast<unsigned long long*>(p);
    p += align256((size_t)B * 2 * NMS_CW * NMS_COLS * 8);
    ws.maskR = reinterpret_cast<unsigned long long*>(p);
    p += align256((size_t)B * 2 * NMS_CW * NMS_COLS * 8);
    ws.sumT = reinterpret_cast<unsigned long long*>(p);
    p += align256((size_t)B * 2 * NMS_CW * NMS_CW * 8);
    ws.sumR = reinterpret_cast<unsigned long long*>(p);
    p += align256((size_t)B * 2 * NMS_CW * NMS_CW * 8);
    ws.keep_idx = reinterpret_cast<int32_t*>(p);
    p += align256((size_t)B * post_n * 4);
    ws.trace = reinterpret_cast<unsigned long long*>(p);
    // worker workgroups: one per compute unit together with the resolvers (a stage of one image
    // has up to 904 tiles for these 4,080 waves: one or two per SIMD)
    int NW = 128 * B;
    NW = NW > 256 - B ? 256 - B : NW;
    // the counters and the kept words start from zero (one small launch; nothing else is read
    // before it is written)
    static_assert(sizeof(unsigned long long) == 8, "deadK words are zeroed as pairs of ints");
    FRCNN_LAUNCH(nms_fused_init_kernel, dim3(1), dim3(1024), 0, st, ints, (int)nf_ints(B, K),
                 reinterpret_cast<int*>(ws.deadK), B * 2 * NMS_CW * 2);
    FRCNN_LAUNCH(nms_fused_kernel, dim3(B + NW), dim3(NF_THREADS), 0, st,
                 reinterpret_cast<const float4*>(boxes), B, K, post_n, nchunk, NW, iou_threshold,
                 1.0f + iou_threshold, iou_threshold * 0.999996f, ws,
                 reinterpret_cast<float4*>(proposed), keep_pos, keep_count);
    return FRCNN_OK;
}

// ---- the allocation-free variant: the NMS workspace is a device array of the code object.
// Each GPU the library runs on holds its own copy of the array at its own address, found once
// per device with hipGetSymbolAddress (in host code the array's name is only a stand-in for
// that lookup, never a device address).
static __device__ __align__(256) unsigned char nms_static_scratch[FRCNN_NMS_STATIC_BYTES];

constexpr int NMS_STATIC_MAX_DEVICES = 64;
static std::atomic<void*> nms_static_ptr[NMS_STATIC_MAX_DEVICES];

extern "C" int frcnn_nms_static(const float* boxes, int B, int K, float iou_threshold, int post_n,
                                float* proposed, int32_t* keep_pos, int32_t* keep_count,
                                frcnn_stream_t stream) {
    if (!boxes || !proposed || B <= 0 || K <= 0 || post_n <= 0) return FRCNN_E_BADARG;
    if (!(iou_threshold >= 0.0f)) return FRCNN_E_BADARG;
    if (frcnn_nms_workspace_bytes(B, K, post_n) > (size_t)FRCNN_NMS_STATIC_BYTES)
        return FRCNN_E_STATIC_LIMIT;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NMS_STATIC_MAX_DEVICES)
        return FRCNN_E_LAUNCH;
    void* ws = nms_static_ptr[dev].load(std::memory_order_acquire);
    if (!ws) {
        // first call on this device (the device is current): two threads racing here look up
        // the same address
        if (hipGetSymbolAddress(&ws, HIP_SYMBOL(nms_static_scratch)) != hipSuccess || !ws)
            return FRCNN_E_LAUNCH;
        nms_static_ptr[dev].store(ws, std::memory_order_release);
    }
    return nms_launch(boxes, B, K, iou_threshold, post_n, proposed, keep_pos, keep_count,
                      reinterpret_cast<char*>(ws), (hipStream_t)stream);
}
