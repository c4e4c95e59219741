// Score top-k by a stable LSD radix sort through global memory, built from ONE-WAVE
// workgroups (64 threads, 1 KB of LDS): such workgroups get wave slots one at a time, so the
// sort runs beside a bandwidth-bound kernel of another stream whose waves fill the chip
// (workgroups of several waves wait there until the other grid has drained), and K has no
// limit.  Same result as frcnn_topk: descending score, equal scores -> lower index first
// (-0 == +0), fused decode of the K winners.
// Reference: Faster_RCNN/module/model.py:45-48 (= module/data.py:203-206).
//
// Four passes over 8-bit digits of the ascending key desc_key(score); per pass
//   hist    : per 1024-element block, digit histogram (one LDS atomic per element) [B][nblk][256]
//   scatter : every block first derives the output place of its first element of every digit
//             from the image's histograms (exclusive scan over (digit, block): a few hundred
//             instructions per block, cheaper than a launch of its own between the two
//             kernels while another stream's kernel fills the chip), then every element
//             goes to that place + its rank among the block's elements of that digit (rank in
//             element order = stable)
// and a final gather of the first K: nine launches.  gfx950, wave64.
#include "frcnn_common.h"

namespace frcnn {

constexpr int TS_ITEMS = 16;
constexpr int TS_BLK = 64 * TS_ITEMS;                   // elements per workgroup (one wave)

template <int PASS>
__device__ __forceinline__ uint32_t ts_key(const float* __restrict__ scores,
                                           const uint32_t* __restrict__ keys_in, size_t i) {
    return PASS == 0 ? desc_key(scores[i]) : keys_in[i];
}

// lanes of the wave that hold the same 8-bit digit as this lane (among `valid` lanes)
__device__ __forceinline__ unsigned long long same_digit(uint32_t dg, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long m = __ballot((dg >> bit) & 1u);
        peers &= ((dg >> bit) & 1u) ? m : ~m;
    }
    return peers;
}

// Workgroup n runs on XCD n % 8.  All blocks of an image go to one XCD (consecutive workgroups
// of that XCD), so the scattered 4-byte writes of a pass meet in ONE L2 and leave it as whole
// lines; spread over the eight L2s they left as partial lines, which the RoI kernel running
// beside the sort paid for at the memory side.
__device__ __forceinline__ void ts_block_of(int n, int nblk, int B, int& blk, int& b) {
    const int full = B & ~7;                            // images that fill whole rounds of 8 XCDs
    if (n < full * nblk) {
        const int slot = n >> 3;
        b = 8 * (slot / nblk) + (n & 7);
        blk = slot % nblk;
    } else {
        const int m = n - full * nblk;
        b = full + m / nblk;
        blk = m % nblk;
    }
}

template <int PASS>
__global__ void __launch_bounds__(64)
ts_hist_kernel(const float* __restrict__ scores, const uint32_t* __restrict__ keys_in, int N,
               int nblk, int B, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    const int lane = threadIdx.x;
    int blk, b;
    ts_block_of((int)blockIdx.x, nblk, B, blk, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) h[t * 64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const size_t base = (size_t)b * N;
#pragma unroll
    for (int it = 0; it < TS_ITEMS; ++it) {
        const int i = blk * TS_BLK + it * 64 + lane;
        const bool valid = i < N;
        const uint32_t dg = valid ? (ts_key<PASS>(scores, keys_in, base + i) >> (8 * PASS)) & 255u : 0u;
#ifdef FRCNN_TS_HIST_BALLOT
        const unsigned long long peers = same_digit(dg, valid);
        if (valid && lane == __ffsll((long long)peers) - 1) {
            volatile uint32_t* p = &h[dg];
            *p = *p + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_wave_barrier();
#else
        // one LDS atomic per lane: the hardware serialises lanes that meet in a bin, which costs
        // less than finding them with eight ballots first (the counts do not depend on the order)
        if (valid) __hip_atomic_fetch_add(&h[dg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the atomics have landed
#pragma unroll
    for (int t = 0; t < 4; ++t)
        hist[((size_t)b * nblk + blk) * 256 + t * 64 + lane] =
            *(volatile uint32_t*)&h[t * 64 + lane];
}

template <int PASS>
__global__ void __launch_bounds__(64)
ts_scatter_kernel(const float* __restrict__ scores, const uint32_t* __restrict__ keys_in,
                  const int32_t* __restrict__ idx_in, int N, int nblk, int B,
                  const uint32_t* __restrict__ hist, uint32_t* __restrict__ keys_out,
                  int32_t* __restrict__ idx_out) {
    __shared__ __attribute__((aligned(16))) uint32_t nextpos[256];   // next output place of each digit
    const int lane = threadIdx.x;
    int blk, b;
    ts_block_of((int)blockIdx.x, nblk, B, blk, b);
    {
        // lane = four consecutive digits.  Elements of digit d in blocks before this one, and
        // in all blocks; the first place of (this block, d) = elements of smaller digits
        // anywhere + elements of d in earlier blocks.
        const uint4* h = reinterpret_cast<const uint4*>(hist + (size_t)b * nblk * 256) + lane;
        uint4 total = make_uint4(0, 0, 0, 0), before = make_uint4(0, 0, 0, 0);
#pragma unroll 4
        for (int k = 0; k < nblk; ++k) {
            const uint4 v = h[(size_t)k * 64];
            total.x += v.x; total.y += v.y; total.z += v.z; total.w += v.w;
            if (k < blk) { before.x += v.x; before.y += v.y; before.z += v.z; before.w += v.w; }
        }
        const uint32_t mine = total.x + total.y + total.z + total.w;
        uint32_t incl = mine;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = __shfl_up(incl, s);
            if (lane >= s) incl += o;
        }
        uint4 first;
        first.x = incl - mine;
        first.y = first.x + total.x;
        first.z = first.y + total.y;
        first.w = first.z + total.z;
        first.x += before.x; first.y += before.y; first.z += before.z; first.w += before.w;
        *reinterpret_cast<uint4*>(&nextpos[4 * lane]) = first;
    }
    __builtin_amdgcn_wave_barrier();
    const size_t base = (size_t)b * N;
    // the wave takes its 1024 elements 64 at a time, so (item, lane) order is element order
    uint32_t key[TS_ITEMS];
    int32_t src[TS_ITEMS];
#pragma unroll
    for (int it = 0; it < TS_ITEMS; ++it) {
        const int i = blk * TS_BLK + it * 64 + lane;
        key[it] = i < N ? ts_key<PASS>(scores, keys_in, base + i) : 0xFFFFFFFFu;
        src[it] = i < N ? (PASS == 0 ? i : idx_in[base + i]) : 0;
    }
#pragma unroll
    for (int it = 0; it < TS_ITEMS; ++it) {
        const int i = blk * TS_BLK + it * 64 + lane;
        const bool valid = i < N;
        const uint32_t dg = (key[it] >> (8 * PASS)) & 255u;
        const unsigned long long peers = same_digit(dg, valid);
        if (valid) {
            const int leader = __ffsll((long long)peers) - 1;
            uint32_t first = 0;
            volatile uint32_t* p = &nextpos[dg];        // other lanes read it at the next item
            if (lane == leader) {
                first = *p;
                *p = first + (uint32_t)__popcll(peers);
            }
            first = __shfl(first, leader);
            const uint32_t pos = first + (uint32_t)__popcll(peers & lanemask_lt());
            keys_out[base + pos] = key[it];
            idx_out[base + pos] = src[it];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(64)
ts_gather_kernel(const float* __restrict__ scores, const int32_t* __restrict__ idx_sorted, int N,
                 int K, const float4* __restrict__ anchors, const float4* __restrict__ deltas,
                 float* __restrict__ top_scores, int32_t* __restrict__ top_idx,
                 float4* __restrict__ cand) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (r >= K) return;
    const int i = idx_sorted[(size_t)b * N + r];
    top_idx[(size_t)b * K + r] = i;
    top_scores[(size_t)b * K + r] = scores[(size_t)b * N + i];
    if (cand) cand[(size_t)b * K + r] = decode_box(anchors[i], deltas[(size_t)b * N + i], false);
}

}  // namespace frcnn

using namespace frcnn;

static size_t ts_align(size_t v) { return (v + 255) & ~(size_t)255; }

// workspace = keys [2][B][N] u32 | idx [2][B][N] i32 | hist [B][nblk][256] u32
extern "C" size_t frcnn_topk_sort_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    const size_t nblk = ((size_t)N + TS_BLK - 1) / TS_BLK;
    return 4 * ts_align((size_t)B * N * 4) + ts_align((size_t)B * nblk * 256 * 4);
}

extern "C" int frcnn_topk_sort(const float* scores, int B, int N, int K, const float* anchors,
                               const float* deltas, float* top_scores, int32_t* top_idx,
                               float* cand_boxes, void* workspace, size_t workspace_bytes,
                               frcnn_stream_t stream) {
    if (!scores || !top_scores || !top_idx || B <= 0 || N <= 0 || K <= 0 || K > N)
        return FRCNN_E_BADARG;
    const bool fuse = anchors && deltas;
    if ((anchors != nullptr) != (deltas != nullptr)) return FRCNN_E_BADARG;
    if (fuse && !cand_boxes) return FRCNN_E_BADARG;
    if (!workspace || workspace_bytes < frcnn_topk_sort_workspace_bytes(B, N)) return FRCNN_E_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 3u) return FRCNN_E_BADARG;
    if (B > 65535 || (long long)B * N > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    const int nblk = (N + TS_BLK - 1) / TS_BLK;
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t plane = ts_align((size_t)B * N * 4);
    uint32_t* keys[2] = {reinterpret_cast<uint32_t*>(ws), reinterpret_cast<uint32_t*>(ws + plane)};
    int32_t* idx[2] = {reinterpret_cast<int32_t*>(ws + 2 * plane),
                       reinterpret_cast<int32_t*>(ws + 3 * plane)};
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 4 * plane);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long long)nblk * B)), block(64);
#define TS_PASS(P, KIN, IIN, KOUT, IOUT)                                                          \
    FRCNN_LAUNCH(ts_hist_kernel<P>, grid, block, 0, st, scores, KIN, N, nblk, B, hist);           \
    FRCNN_LAUNCH(ts_scatter_kernel<P>, grid, block, 0, st, scores, KIN, IIN, N, nblk, B, hist, KOUT, \
                 IOUT)
    TS_PASS(0, (const uint32_t*)nullptr, (const int32_t*)nullptr, keys[0], idx[0]);
    TS_PASS(1, keys[0], idx[0], keys[1], idx[1]);
    TS_PASS(2, keys[1], idx[1], keys[0], idx[0]);
    TS_PASS(3, keys[0], idx[0], keys[1], idx[1]);
#undef TS_PASS
    FRCNN_LAUNCH(ts_gather_kernel, dim3((K + 63) / 64, B), dim3(64), 0, st, scores, idx[1], N, K,
                 reinterpret_cast<const float4*>(anchors), reinterpret_cast<const float4*>(deltas),
                 top_scores, top_idx, fuse ? reinterpret_cast<float4*>(cand_boxes) : nullptr);
    return FRCNN_OK;
}
