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
//             element order = stable).  The last scatter writes the first K places straight
//             into the outputs (score, index, decoded box) and drops the rest.
// Eight launches.  (Counting an element into the NEXT pass's histogram from the scatter, by a
// global atomic at its new place, removes three launches and was slower: 97 us against 86 for
// one image, 207 against 132 for 64 -- one memory-side atomic per element and pass.)
// gfx950, wave64.
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

// The first K places of the last pass, written where the caller wants them.
struct TsOut {
    int K;
    const float4* anchors;
    const float4* deltas;
    float* top_scores;
    int32_t* top_idx;
    float4* cand;
};

template <int PASS>
__global__ void __launch_bounds__(64)
ts_scatter_kernel(const float* __restrict__ scores, const uint32_t* __restrict__ keys_in,
                  const int32_t* __restrict__ idx_in, int N, int nblk, int B,
                  const uint32_t* __restrict__ hist, uint32_t* __restrict__ keys_out,
                  int32_t* __restrict__ idx_out, TsOut out) {
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
            if (PASS < 3) {
                keys_out[base + pos] = key[it];
                idx_out[base + pos] = src[it];
            } else if ((int)pos < out.K) {
                const size_t o = (size_t)b * out.K + pos;
                out.top_idx[o] = src[it];
                out.top_scores[o] = scores[base + src[it]];
                if (out.cand)
                    out.cand[o] = decode_box(out.anchors[src[it]], out.deltas[base + src[it]], false);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- few images: sorted runs + ranks ----------------------------------------------------------
// The radix passes above are eight dependent launches of a few microseconds each: for one image
// that chain IS the cost.  When the whole call is a few hundred blocks, two launches do it:
//   runs : every 1,024-element block is sorted by one 256-thread workgroup (bitonic network, four
//          elements per thread; element = key << 32 | index, so every element is distinct and
//          the order of equal scores is the index order the radix sort keeps); keys and indices
//          are stored apart
//   rank : an element's place in the image = its place in its own run + the number of elements
//          of every other run that come before it.  A run holds consecutive indices, so among
//          equal keys a whole earlier run comes before and a whole later run after: the count is
//          a binary search on the 32-bit keys alone (<= for earlier runs, < for later ones).
//          The other runs of a run are shared out over `nsplit` workgroups and their waves
//          (one or two runs per wave, staged through LDS, the sixteen elements of a lane
//          searched in step so that sixteen LDS reads are in flight); a workgroup leaves its
//          partial counts in memory, and the one that arrives last adds them up and writes the
//          first K places straight into the outputs.
// Same result, bit for bit.  Work grows with runs-per-image squared, so the radix sort stays
// the path for batches.
constexpr int TM_RUN = 1024;
constexpr int TM_THREADS = 256;
constexpr int TM_WAVES = TM_THREADS / 64;
constexpr int TM_PER_THREAD = TM_RUN / TM_THREADS;      // 4
constexpr int TM_PER_LANE = TM_RUN / 64;                // 16: a wave of the rank kernel holds the run
constexpr int TM_MAX_RUNS_PER_IMAGE = 64;
constexpr int TM_MAX_SPLIT = 8;
constexpr long long TM_DEFAULT_MAX_RUNS = 256;           // blocks in the call (images x blocks)

struct TmWorkspace {
    uint32_t* keys;      // [B][nblk][TM_RUN] sorted keys of every run
    int32_t* idx;        // [B][nblk][TM_RUN] their score indices
    uint32_t* part;      // [B][nblk][nsplit][TM_RUN] partial counts
    int* arrived;        // [B][nblk] workgroups of the run that have left their counts
};

__device__ __forceinline__ unsigned long long tm_xor_lane(unsigned long long v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
    return ((unsigned long long)hi << 32) | lo;
}

// ascending bitonic sort of TM_RUN elements by TM_THREADS threads; element e sits in thread
// e / TM_PER_THREAD, register e % TM_PER_THREAD.  Partners in the same wave are exchanged by
// lane permutes, partners in another wave through `s` (TM_RUN words).
__device__ __forceinline__ void tm_sort_block(unsigned long long (&v)[TM_PER_THREAD], int tid,
                                              unsigned long long* s) {
    constexpr int T = TM_PER_THREAD;
#pragma unroll
    for (int k = 2; k <= TM_RUN; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= T) {                               // partner: same register, thread tid ^ (j / T)
                const int tj = j / T;
                const bool asc = ((tid * T) & k) == 0;
                const bool keep_min = ((tid & tj) == 0) == asc;
                unsigned long long o[T];
                if (tj < 64) {
#pragma unroll
                    for (int it = 0; it < T; ++it) o[it] = tm_xor_lane(v[it], tj);
                } else {
                    __syncthreads();                    // the words of the exchange before are read
#pragma unroll
                    for (int it = 0; it < T; ++it) s[tid * T + it] = v[it];
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < T; ++it) o[it] = s[(tid ^ tj) * T + it];
                }
#pragma unroll
                for (int it = 0; it < T; ++it) {
                    // one compare and no branch: elements are distinct (equal fillers swap for
                    // themselves)
                    const bool take = (o[it] < v[it]) == keep_min;
                    v[it] = take ? o[it] : v[it];
                }
            } else {                                    // partner: register it ^ j of this thread
#pragma unroll
                for (int it = 0; it < T; ++it) {
                    if ((it & j) != 0) continue;
                    const int p = it | j;
                    const bool asc = (((tid * T) | it) & k) == 0;
                    const unsigned long long a = v[it], c = v[p];
                    const bool sw = (a > c) == asc;
                    v[it] = sw ? c : a;
                    v[p] = sw ? a : c;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(TM_THREADS)
tm_runs_kernel(const float* __restrict__ scores, int N, int nblk, TmWorkspace ws) {
    __shared__ unsigned long long s[TM_RUN];
    const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    const size_t base = (size_t)b * N;
    unsigned long long v[TM_PER_THREAD];
#pragma unroll
    for (int it = 0; it < TM_PER_THREAD; ++it) {
        const int i = blk * TM_RUN + it * TM_THREADS + tid;     // coalesced; any start order will do
        v[it] = i < N ? ((unsigned long long)desc_key(scores[base + i]) << 32) | (uint32_t)i
                      : ~0ull;                          // filler: behind every element
    }
    tm_sort_block(v, tid, s);
    const size_t run = (size_t)b * nblk + blk;
    uint4 k4, i4;
    k4.x = (uint32_t)(v[0] >> 32); k4.y = (uint32_t)(v[1] >> 32);
    k4.z = (uint32_t)(v[2] >> 32); k4.w = (uint32_t)(v[3] >> 32);
    i4.x = (uint32_t)v[0]; i4.y = (uint32_t)v[1]; i4.z = (uint32_t)v[2]; i4.w = (uint32_t)v[3];
    static_assert(TM_PER_THREAD == 4, "a thread stores its four keys as one 16-byte word");
    reinterpret_cast<uint4*>(ws.keys + run * TM_RUN)[tid] = k4;
    reinterpret_cast<uint4*>(ws.idx + run * TM_RUN)[tid] = i4;
    if (tid == 0) ws.arrived[run] = 0;
}

// For the sixteen keys of a lane: how many of the sorted keys s[0 .. TM_RUN) come before each.
// TIES: equal keys of s come before (s is an earlier run).
template <bool TIES>
__device__ __forceinline__ void tm_count(const uint32_t* s, const uint32_t (&own)[TM_PER_LANE],
                                         uint32_t (&cnt)[TM_PER_LANE]) {
    constexpr int L = TM_PER_LANE;
    uint32_t pos[L];
#pragma unroll
    for (int it = 0; it < L; ++it) pos[it] = 0;
#pragma unroll 1
    for (int h = TM_RUN / 2; h >= 1; h >>= 1) {
        uint32_t w[L];
#pragma unroll
        for (int it = 0; it < L; ++it) w[it] = s[pos[it] + h - 1];
#pragma unroll
        for (int it = 0; it < L; ++it) pos[it] += (TIES ? w[it] <= own[it] : w[it] < own[it]) ? h : 0;
    }
    uint32_t w[L];
#pragma unroll
    for (int it = 0; it < L; ++it) w[it] = s[pos[it]];
#pragma unroll
    for (int it = 0; it < L; ++it)
        cnt[it] += pos[it] + ((TIES ? w[it] <= own[it] : w[it] < own[it]) ? 1u : 0u);
}

__global__ void __launch_bounds__(TM_THREADS)
tm_rank_kernel(const float* __restrict__ scores, int N, int nblk, int nsplit, TmWorkspace ws,
               TsOut out) {
    constexpr int L = TM_PER_LANE;
    // one staging buffer per wave: LDS serves a wave's accesses in order, so the next run's words
    // land behind the last search's reads.  16 KB: fits beside a band-resident RoI workgroup.
    __shared__ __attribute__((aligned(16))) uint32_t s_run[TM_WAVES][TM_RUN];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const size_t run0 = (size_t)b * nblk;
    const uint32_t* keys = ws.keys + run0 * TM_RUN;
    // every wave holds the keys of the whole run, sixteen consecutive ones per lane
    uint32_t own[L], cnt[L];
    {
        const uint4* src = reinterpret_cast<const uint4*>(keys + (size_t)blk * TM_RUN + lane * L);
#pragma unroll
        for (int t = 0; t < L / 4; ++t) {
            const uint4 v = src[t];
            own[4 * t] = v.x; own[4 * t + 1] = v.y; own[4 * t + 2] = v.z; own[4 * t + 3] = v.w;
        }
#pragma unroll
        for (int it = 0; it < L; ++it) cnt[it] = 0;
    }
    // this wave's share of the other runs: every (TM_WAVES * nsplit)-th of the runs that are not
    // blk.  The next one is on its way from memory while this one is searched.
    const int stride = TM_WAVES * nsplit;
    static_assert(L == 16, "a lane moves a run as four 16-byte words");
    uint4 n0, n1, n2, n3;
    n0 = n1 = n2 = n3 = make_uint4(0, 0, 0, 0);
    int q = split * TM_WAVES + wave;                    // position in the list of other runs
    int r = q < blk ? q : q + 1;
    if (r < nblk) {
        const uint4* src = reinterpret_cast<const uint4*>(keys + (size_t)r * TM_RUN) + lane;
        n0 = src[0]; n1 = src[64]; n2 = src[128]; n3 = src[192];
    }
    while (r < nblk) {                                  // wave-uniform
        uint32_t* s = s_run[wave];
        uint4* dst = reinterpret_cast<uint4*>(s) + lane;
        dst[0] = n0; dst[64] = n1; dst[128] = n2; dst[192] = n3;
        q += stride;
        const int rn = q < blk ? q : q + 1;
        if (rn < nblk) {
            const uint4* src = reinterpret_cast<const uint4*>(keys + (size_t)rn * TM_RUN) + lane;
            n0 = src[0]; n1 = src[64]; n2 = src[128]; n3 = src[192];
        }
        __builtin_amdgcn_wave_barrier();
        if (r < blk) tm_count<true>(s, own, cnt);       // wave-uniform
        else tm_count<false>(s, own, cnt);
        __builtin_amdgcn_wave_barrier();
        r = rn;
    }
    // the four waves' counts of an element meet in LDS; thread t then speaks for elements 4t .. 4t+3
    __syncthreads();
    uint32_t* part = &s_run[0][0];                      // [TM_WAVES][TM_RUN]
#pragma unroll
    for (int it = 0; it < L; ++it) part[wave * TM_RUN + lane * L + it] = cnt[it];
    __syncthreads();
    uint32_t place[TM_PER_THREAD];
#pragma unroll
    for (int it = 0; it < TM_PER_THREAD; ++it) {
        const int e = tid * TM_PER_THREAD + it;
        place[it] = 0;
#pragma unroll
        for (int w = 0; w < TM_WAVES; ++w) place[it] += part[w * TM_RUN + e];
    }
    if (nsplit > 1) {
        // Counts go to memory as device-scope atomic stores (served at the coherent point, past
        // the per-XCD L2s) and have completed (vmcnt) before this workgroup is counted as
        // arrived; the last one reads all of them the same way.
        int* mine = reinterpret_cast<int*>(ws.part + ((run0 + blk) * nsplit + split) * TM_RUN);
#pragma unroll
        for (int it = 0; it < TM_PER_THREAD; ++it)
            __hip_atomic_store(mine + tid * TM_PER_THREAD + it, (int)place[it], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            s_last = __hip_atomic_fetch_add(ws.arrived + run0 + blk, 1, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1;
        __syncthreads();
        if (!s_last) return;                            // workgroup-uniform
        const int* all = reinterpret_cast<const int*>(ws.part + (run0 + blk) * nsplit * TM_RUN);
#pragma unroll
        for (int it = 0; it < TM_PER_THREAD; ++it) {
            place[it] = 0;
            for (int sp = 0; sp < nsplit; ++sp)
                place[it] += (uint32_t)__hip_atomic_load(all + sp * TM_RUN + tid * TM_PER_THREAD + it,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const size_t base = (size_t)b * N;
    const uint4 i4 = reinterpret_cast<const uint4*>(ws.idx + (run0 + blk) * TM_RUN)[tid];
    const uint32_t mine_idx[TM_PER_THREAD] = {i4.x, i4.y, i4.z, i4.w};
    int idx[TM_PER_THREAD];
    bool live[TM_PER_THREAD];
    float sc[TM_PER_THREAD];
    float4 an[TM_PER_THREAD], dl[TM_PER_THREAD];
#pragma unroll
    for (int it = 0; it < TM_PER_THREAD; ++it) {
        place[it] += tid * TM_PER_THREAD + it;
        // a filler carries index 0xFFFFFFFF (no score has it: N < 2^31)
        live[it] = (int)place[it] < out.K && mine_idx[it] != 0xFFFFFFFFu;
        idx[it] = live[it] ? (int)mine_idx[it] : 0;
        sc[it] = scores[base + idx[it]];
        if (out.cand) {
            an[it] = out.anchors[idx[it]];
            dl[it] = out.deltas[base + idx[it]];
        }
    }
#pragma unroll
    for (int it = 0; it < TM_PER_THREAD; ++it) {
        if (!live[it]) continue;
        const size_t o = (size_t)b * out.K + place[it];
        out.top_idx[o] = idx[it];
        out.top_scores[o] = sc[it];
        if (out.cand) out.cand[o] = decode_box(an[it], dl[it], false);
    }
}

}  // namespace frcnn

using namespace frcnn;

static size_t ts_align(size_t v) { return (v + 255) & ~(size_t)255; }

// sorted-runs form: workgroups that share the other runs of a run (every wave gets one or two)
static int tm_nsplit(int nblk) {
    const int want = (nblk - 1 + TM_WAVES - 1) / TM_WAVES;
    return want < 1 ? 1 : want > TM_MAX_SPLIT ? TM_MAX_SPLIT : want;
}
static bool tm_wanted(int B, int nblk) {
    long long max_runs = tuning(TUNE_TOPK_MERGE_MAX_RUNS);
    if (max_runs == TUNE_UNSET) max_runs = TM_DEFAULT_MAX_RUNS;
    return nblk <= TM_MAX_RUNS_PER_IMAGE && (long long)B * nblk <= max_runs && B <= 65535;
}
// keys | idx | partial counts | arrival counters
static size_t tm_bytes(int B, int nblk) {
    const size_t runs = (size_t)B * nblk;
    return 2 * ts_align(runs * TM_RUN * 4) + ts_align(runs * tm_nsplit(nblk) * TM_RUN * 4) +
           ts_align(runs * 4);
}

// radix passes: keys [2][B][N] u32 | idx [2][B][N] i32 | hist [B][nblk][256] u32; the sorted-runs
// form lays its own buffers over the same bytes
static size_t ts_radix_bytes(int B, int N) {
    const size_t nblk = ((size_t)N + TS_BLK - 1) / TS_BLK;
    return 4 * ts_align((size_t)B * N * 4) + ts_align((size_t)B * nblk * 256 * 4);
}

extern "C" size_t frcnn_topk_sort_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    const size_t nblk = ((size_t)N + TS_BLK - 1) / TS_BLK;
    const size_t radix = ts_radix_bytes(B, N);
    const size_t runs = nblk <= (size_t)TM_MAX_RUNS_PER_IMAGE && tm_wanted(B, (int)nblk)
                            ? tm_bytes(B, (int)nblk) : 0;
    return radix > runs ? radix : runs;
}

extern "C" int frcnn_topk_sort_plan(int B, int N, size_t workspace_bytes) {
    if (B <= 0 || N <= 0) return 0;
    const int nblk = (N + TS_BLK - 1) / TS_BLK;
    return tm_wanted(B, nblk) && workspace_bytes >= tm_bytes(B, nblk) ? 1 : 0;
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
    // what the radix passes need always does; the sorted-runs form runs when there is room for it
    // (a tuning value changed after the caller sized the workspace does not fail the call)
    if (!workspace || workspace_bytes < ts_radix_bytes(B, N)) return FRCNN_E_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 15u) return FRCNN_E_BADARG;
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
    const TsOut out{K, reinterpret_cast<const float4*>(anchors), reinterpret_cast<const float4*>(deltas),
                    top_scores, top_idx, fuse ? reinterpret_cast<float4*>(cand_boxes) : nullptr};
    if (frcnn_topk_sort_plan(B, N, workspace_bytes)) {
        const size_t runs = (size_t)B * nblk;
        const int nsplit = tm_nsplit(nblk);
        TmWorkspace tw;
        char* p = ws;
        tw.keys = reinterpret_cast<uint32_t*>(p);
        p += ts_align(runs * TM_RUN * 4);
        tw.idx = reinterpret_cast<int32_t*>(p);
        p += ts_align(runs * TM_RUN * 4);
        tw.part = reinterpret_cast<uint32_t*>(p);
        p += ts_align(runs * nsplit * TM_RUN * 4);
        tw.arrived = reinterpret_cast<int*>(p);
        FRCNN_LAUNCH(tm_runs_kernel, dim3(nblk, B), dim3(TM_THREADS), 0, st, scores, N, nblk, tw);
        FRCNN_LAUNCH(tm_rank_kernel, dim3(nblk, nsplit, B), dim3(TM_THREADS), 0, st, scores, N, nblk,
                     nsplit, tw, out);
        return FRCNN_OK;
    }
#define TS_PASS(P, KIN, IIN, KOUT, IOUT)                                                          \
    FRCNN_LAUNCH(ts_hist_kernel<P>, grid, block, 0, st, scores, KIN, N, nblk, B, hist);           \
    FRCNN_LAUNCH(ts_scatter_kernel<P>, grid, block, 0, st, scores, KIN, IIN, N, nblk, B, hist, KOUT, \
                 IOUT, out)
    TS_PASS(0, (const uint32_t*)nullptr, (const int32_t*)nullptr, keys[0], idx[0]);
    TS_PASS(1, keys[0], idx[0], keys[1], idx[1]);
    TS_PASS(2, keys[1], idx[1], keys[0], idx[0]);
    TS_PASS(3, keys[0], idx[0], (uint32_t*)nullptr, (int32_t*)nullptr);
#undef TS_PASS
    return FRCNN_OK;
}
