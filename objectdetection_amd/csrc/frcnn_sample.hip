// Mini-batch sampling for the RPN (anchors) and for the head (proposals),
// plus proposal <-> GT labelling.  One workgroup per image: ordered ballot
// compaction, LDS radix select of the smallest shuffle keys, LDS bitonic sort.
// The workgroup is 1024 threads for small batches (shortest time for one image) and
// one wave for large ones (a one-wave workgroup gets a wave slot while a
// bandwidth-bound kernel of another stream fills the chip; larger ones wait for it).
// gfx950, wave64.  -ffp-contract=off.
#include "frcnn_common.h"

namespace frcnn {

// threads of the one-workgroup-per-image sampling kernels: SBLOCK_BIG below
// SBLOCK_SWITCH images, FRCNN_TARGET_BLOCK from there on
constexpr int SBLOCK_BIG = 1024;
constexpr int SBLOCK_SWITCH = 16;
constexpr int SBLOCK_MID = 256;                // one wave per SIMD: still fits beside a RoI workgroup
constexpr int SMAX_BATCH = 1024;
// mini-batches up to SMALL_BATCH use the small instance of the LDS block (8.4 KB instead of
// 29.9 KB): a workgroup that small still fits on a compute unit whose LDS is all but taken by
// the band-resident RoI kernel of another stream (154 KB of 160), so it does not keep that
// kernel's workgroups off the unit for as long as it runs
constexpr int SMALL_BATCH = 256;

template <int SMAX>
struct SampleShared {
    unsigned long long sel[SMAX];          // composites being selected / sorted
    unsigned long long inv[SMAX];          // (list position << 32 | slot), sorted by position
    uint32_t slot_item[SMAX];              // payload per slot (anchor id / proposal id)
    int32_t slot_gt[SMAX];                 // matched GT per slot
    uint32_t order[SMAX];                  // final shuffle: order[r] = slot
    uint32_t hist[256];
    uint32_t sh[4];
    uint32_t wcnt[3 * (SBLOCK_BIG / 64)];
    uint32_t ctr[2];
    int32_t tot[4];
};

// smallest `m` of key(0..M-1) by (key, position) -> s.sel[0..m) sorted ascending.
template <int SBLOCK, typename Shared, typename KeyFn>
__device__ void select_sorted(Shared& s, KeyFn key, int M, int m) {
    if (m <= 0) return;                    // block-uniform
    uint32_t T;
    int less;
    block_radix_select<SBLOCK>(key, M, m, s.hist, s.sh, T, less);
    block_collect_smallest<SBLOCK>(key, M, m, T, less, s.sel, s.wcnt, s.ctr);
    const int n = next_pow2(m);
    for (int t = threadIdx.x + m; t < n; t += SBLOCK) s.sel[t] = ~0ull;
    __syncthreads();
    block_bitonic_sort<SBLOCK>(s.sel, n);
}

// binary search of list position `pos` among inv[lo..hi) (sorted by position);
// returns slot or -1.
__device__ __forceinline__ int find_slot(const unsigned long long* inv, int lo, int hi,
                                         uint32_t pos) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t p = (uint32_t)(inv[mid] >> 32);
        if (p < pos) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Shared body.  flag(i) -> bit0 positive, bit1 negative, bit2 counts toward the
// negative payload rank ("valid").  Positives carry payload i; negatives carry
// their rank among bit2 items when NEG_RANK (the reference's RPN behaviour:
// tf.where over the valid-anchor sub-list, module/data.py:108-110) else i.
// On return (all threads): s.slot_item/slot_gt hold [positives | negatives] in
// shuffled order, s.order the final permutation, s.tot = {P, Nneg, nop, status}.
template <int SBLOCK, bool NEG_RANK, typename Shared, typename FlagFn, typename GtFn>
__device__ void sample_two_lists(Shared& s, FlagFn flag, GtFn pos_gt, int M, int neg_gt,
                                 const uint32_t* kp, const uint32_t* kn, const uint32_t* ka,
                                 uint64_t seed, uint32_t image, uint32_t stream_base, int batch,
                                 int max_pos) {
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    constexpr int NW = SBLOCK / 64;
    // ---- pass A: list lengths
    int cp = 0, cn = 0;
    for (int i = tid; i < M; i += SBLOCK) {
        const uint32_t f = flag(i);
        cp += (f & 1u) ? 1 : 0;
        cn += (f & 2u) ? 1 : 0;
    }
    for (int d = 32; d > 0; d >>= 1) { cp += __shfl_down(cp, d); cn += __shfl_down(cn, d); }
    if (lane == 0) { s.wcnt[wave] = (uint32_t)cp; s.wcnt[NW + wave] = (uint32_t)cn; }
    __syncthreads();
    if (tid == 0) {
        int P = 0, Nn = 0;
        for (int w = 0; w < NW; ++w) { P += (int)s.wcnt[w]; Nn += (int)s.wcnt[NW + w]; }
        const int nop = P < max_pos ? P : max_pos;
        s.tot[0] = P; s.tot[1] = Nn; s.tot[2] = nop;
        s.tot[3] = (Nn < batch - nop) ? FRCNN_S_SHORT_BATCH : FRCNN_S_OK;
    }
    __syncthreads();
    const int P = s.tot[0], Nn = s.tot[1], nop = s.tot[2];
    const int non = batch - nop;
    if (s.tot[3] != FRCNN_S_OK) return;     // block-uniform

    // ---- shuffled prefix of the positive list -> inv[0..nop)
    auto key_pos = [&](int i) -> uint32_t {
        return kp ? kp[i] : counter_key((uint32_t)i, stream_base + 0u, image, seed);
    };
    select_sorted<SBLOCK>(s, key_pos, P, nop);
    for (int t = tid; t < nop; t += SBLOCK)
        s.inv[t] = ((unsigned long long)(uint32_t)s.sel[t] << 32) | (uint32_t)t;
    __syncthreads();
    // ---- shuffled prefix of the negative list -> inv[nop..batch)
    auto key_neg = [&](int i) -> uint32_t {
        return kn ? kn[i] : counter_key((uint32_t)i, stream_base + 1u, image, seed);
    };
    select_sorted<SBLOCK>(s, key_neg, Nn, non);
    for (int t = tid; t < non; t += SBLOCK)
        s.inv[nop + t] = ((unsigned long long)(uint32_t)s.sel[t] << 32) | (uint32_t)(nop + t);
    __syncthreads();
    // sort each half of inv by list position (slot rides in the low word)
    {
        const int np = next_pow2(nop > 0 ? nop : 1), nn = next_pow2(non > 0 ? non : 1);
        // positives: copy to sel, pad, sort, copy back
        for (int t = tid; t < np; t += SBLOCK) s.sel[t] = t < nop ? s.inv[t] : ~0ull;
        __syncthreads();
        block_bitonic_sort<SBLOCK>(s.sel, np);
        for (int t = tid; t < nop; t += SBLOCK) s.inv[t] = s.sel[t];
        __syncthreads();
        for (int t = tid; t < nn; t += SBLOCK) s.sel[t] = t < non ? s.inv[nop + t] : ~0ull;
        __syncthreads();
        block_bitonic_sort<SBLOCK>(s.sel, nn);
        for (int t = tid; t < non; t += SBLOCK) s.inv[nop + t] = s.sel[t];
        __syncthreads();
    }
    // ---- pass B: ordered ranks -> slots
    int base_p = 0, base_n = 0, base_v = 0;
    const int iters = (M + SBLOCK - 1) / SBLOCK;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int i = it * SBLOCK + tid;
        const uint32_t f = i < M ? flag(i) : 0u;
        const unsigned long long bp = __ballot((f & 1u) != 0);
        const unsigned long long bn = __ballot((f & 2u) != 0);
        const unsigned long long bv = __ballot((f & 4u) != 0);
        if (lane == 0) {
            s.wcnt[wave] = (uint32_t)__popcll(bp);
            s.wcnt[NW + wave] = (uint32_t)__popcll(bn);
            s.wcnt[2 * NW + wave] = (uint32_t)__popcll(bv);
        }
        __syncthreads();
        int bef_p = 0, bef_n = 0, bef_v = 0, tp = 0, tn = 0, tv = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int a = (int)s.wcnt[w], b = (int)s.wcnt[NW + w], c = (int)s.wcnt[2 * NW + w];
            if (w < wave) { bef_p += a; bef_n += b; bef_v += c; }
            tp += a; tn += b; tv += c;
        }
        const unsigned long long lt = lanemask_lt();
        if (f & 1u) {
            const uint32_t rank = (uint32_t)(base_p + bef_p + __popcll(bp & lt));
            const int k = find_slot(s.inv, 0, nop, rank);
            if (k < nop && (uint32_t)(s.inv[k] >> 32) == rank) {
                const uint32_t slot = (uint32_t)s.inv[k];
                s.slot_item[slot] = (uint32_t)i;
                s.slot_gt[slot] = pos_gt(i);
            }
        }
        if (f & 2u) {
            const uint32_t rank = (uint32_t)(base_n + bef_n + __popcll(bn & lt));
            const int k = find_slot(s.inv, nop, batch, rank);
            if (k < batch && (uint32_t)(s.inv[k] >> 32) == rank) {
                const uint32_t slot = (uint32_t)s.inv[k];
                const uint32_t vrank = (uint32_t)(base_v + bef_v + __popcll(bv & lt));
                s.slot_item[slot] = NEG_RANK ? vrank : (uint32_t)i;
                s.slot_gt[slot] = neg_gt;
            }
        }
        base_p += tp; base_n += tn; base_v += tv;
        __syncthreads();
    }
    // ---- final shuffle of range(batch)
    auto key_all = [&](int i) -> uint32_t {
        return ka ? ka[i] : counter_key((uint32_t)i, stream_base + 2u, image, seed);
    };
    const int nb = next_pow2(batch);
    for (int t = tid; t < nb; t += SBLOCK)
        s.sel[t] = t < batch ? (((unsigned long long)key_all(t) << 32) | (uint32_t)t) : ~0ull;
    __syncthreads();
    block_bitonic_sort<SBLOCK>(s.sel, nb);
    for (int t = tid; t < batch; t += SBLOCK) s.order[t] = (uint32_t)s.sel[t];
    __syncthreads();
}

// ---------------------------------------------------------------- RPN sampler ----
template <int SBLOCK, int SMAX>
__global__ void __launch_bounds__(SBLOCK)
rpn_sample_kernel(const float4* __restrict__ anchors, int N, int W, int A,
                  const float4* __restrict__ gt, int G, const int32_t* __restrict__ best_gt,
                  const uint8_t* __restrict__ flags, const uint32_t* __restrict__ keys_pos,
                  const uint32_t* __restrict__ keys_neg, const uint32_t* __restrict__ keys_all,
                  uint64_t seed, int image_base, int batch, int max_pos,
                  float* __restrict__ batch_label, float4* __restrict__ batch_anchor,
                  float4* __restrict__ batch_gt, long long* __restrict__ gt_list,
                  long long* __restrict__ batch_pos, float4* __restrict__ batch_reg_gt,
                  int32_t* __restrict__ counts) {
    __shared__ SampleShared<SMAX> s;
    const int b = blockIdx.x;
    const uint8_t* fl = flags + (size_t)b * N;
    const int32_t* bg = best_gt + (size_t)b * N;
    auto flag = [&](int i) -> uint32_t {
        const uint32_t f = fl[i];
        return ((f & FRCNN_F_POSITIVE) ? 1u : 0u) | ((f & FRCNN_F_NEGATIVE) ? 2u : 0u) |
               ((f & FRCNN_F_VALID) ? 4u : 0u);
    };
    auto pos_gt = [&](int i) -> int { return bg[i]; };
    sample_two_lists<SBLOCK, true>(s, flag, pos_gt, N, G - 1,
                           keys_pos ? keys_pos + (size_t)b * N : nullptr,
                           keys_neg ? keys_neg + (size_t)b * N : nullptr,
                           keys_all ? keys_all + (size_t)b * batch : nullptr, seed,
                           (uint32_t)(image_base + b), 0u, batch, max_pos);
    const int tid = threadIdx.x;
    if (tid < 4) counts[b * 4 + tid] = s.tot[tid];
    const bool ok = s.tot[3] == FRCNN_S_OK;
    const int nop = s.tot[2];
    const float4* g_img = gt + (size_t)b * G;
    for (int r = tid; r < batch; r += SBLOCK) {
        const size_t o = (size_t)b * batch + r;
        if (!ok) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            batch_label[o] = 0.f; batch_anchor[o] = z; batch_gt[o] = z; batch_reg_gt[o] = z;
            gt_list[o] = 0; batch_pos[o * 3] = 0; batch_pos[o * 3 + 1] = 0; batch_pos[o * 3 + 2] = 0;
            continue;
        }
        const uint32_t slot = s.order[r];
        const uint32_t aid = s.slot_item[slot];
        const int g = s.slot_gt[slot];
        const float4 an = anchors[aid];
        const float4 gb = g_img[g];
        batch_label[o] = slot < (uint32_t)nop ? 1.0f : 0.0f;
        batch_anchor[o] = an;
        batch_gt[o] = gb;
        gt_list[o] = g;
        batch_pos[o * 3 + 0] = (aid / A) / W;      // module/data.py:117-118
        batch_pos[o * 3 + 1] = (aid / A) % W;
        batch_pos[o * 3 + 2] = aid % A;
        batch_reg_gt[o] = encode_box(an, gb, false);
    }
}

// --------------------------------------------------------------- head sampler ----
template <int SBLOCK, int SMAX>
__global__ void __launch_bounds__(SBLOCK)
sample_proposals_kernel(const long long* __restrict__ p_label, int R, int bg_label,
                        const uint32_t* __restrict__ keys_pos, const uint32_t* __restrict__ keys_neg,
                        const uint32_t* __restrict__ keys_all, uint64_t seed, int image_base,
                        int batch, int max_pos, long long* __restrict__ tindex,
                        int32_t* __restrict__ counts) {
    __shared__ SampleShared<SMAX> s;
    const int b = blockIdx.x;
    const long long* pl = p_label + (size_t)b * R;
    auto flag = [&](int i) -> uint32_t { return pl[i] != (long long)bg_label ? 1u : 2u; };
    auto pos_gt = [&](int) -> int { return 0; };
    sample_two_lists<SBLOCK, false>(s, flag, pos_gt, R, 0,
                            keys_pos ? keys_pos + (size_t)b * R : nullptr,
                            keys_neg ? keys_neg + (size_t)b * R : nullptr,
                            keys_all ? keys_all + (size_t)b * batch : nullptr, seed,
                            (uint32_t)(image_base + b), 3u, batch, max_pos);
    const int tid = threadIdx.x;
    if (tid < 4) counts[b * 4 + tid] = s.tot[tid];
    const bool ok = s.tot[3] == FRCNN_S_OK;
    for (int r = tid; r < batch; r += SBLOCK)
        tindex[(size_t)b * batch + r] = ok ? (long long)s.slot_item[s.order[r]] : 0ll;
}

// --------------------------------------------------------- proposal labelling ----
// One thread per proposal; the image's GT boxes, areas and labels sit in LDS.
// Op order of module/data.py:215-259.
__global__ void __launch_bounds__(FRCNN_TARGET_BLOCK > 256 ? FRCNN_TARGET_BLOCK : 256)
label_proposals_kernel(const float4* __restrict__ proposed, const float4* __restrict__ gt,
                       const long long* __restrict__ gt_label_in, int R, int G, float fg_thr,
                       float iou_floor, int bg_label, int num_classes,
                       long long* __restrict__ p_label, int32_t* __restrict__ gt_mask,
                       float* __restrict__ gt_onehot, float4* __restrict__ gt_coord,
                       int32_t* __restrict__ best_out, float* __restrict__ max_iou) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_gt = reinterpret_cast<float4*>(smem);
    long long* s_lab = reinterpret_cast<long long*>(s_gt + G);
    float* s_size = reinterpret_cast<float*>(s_lab + G);
    const int b = blockIdx.y;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float4 v = gt[(size_t)b * G + g];
        s_gt[g] = v;
        s_size[g] = (v.z - v.x) * (v.w - v.y);
        s_lab[g] = gt_label_in[(size_t)b * G + g];
    }
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const size_t o = (size_t)b * R + r;
    const float4 a = proposed[o];
    const float a_size = (a.z - a.x) * (a.w - a.y);
    float best = 0.0f;
    int arg = 0;
    for (int g = 0; g < G; ++g) {
        const float4 q = s_gt[g];
        const float xmin = fmaxf(a.y, q.y);
        const float ymin = fmaxf(a.x, q.x);
        const float xmax = fminf(a.w, q.w);
        const float ymax = fminf(a.z, q.z);
        const float inter = (xmax - xmin) * (ymax - ymin);
        const float inter2 = (xmax > xmin && ymax > ymin) ? inter : 0.0f;
        const float inter3 = inter2 > 0.0f ? inter2 : 1e-5f;         // data.py:228
        const float uni = s_size[g] + a_size - inter3;
        const float iou = inter3 / uni;
        float iou4 = iou >= iou_floor ? iou : 0.0f;                   // data.py:233
        iou4 = fminf(fmaxf(iou4, 0.0f), 1.0f);                        // data.py:234
        if (g == 0 || iou4 > best) { best = iou4; arg = g; }
    }
    const long long plabel = s_lab[arg];
    long long lab = best < fg_thr ? (long long)bg_label : plabel;     // data.py:242
    lab = best >= fg_thr ? plabel : lab;                              // data.py:243
    p_label[o] = lab;
    gt_mask[o] = lab != (long long)bg_label ? 1 : 0;
    if (gt_onehot) {
        float* oh = gt_onehot + o * (size_t)num_classes;
        for (int c = 0; c < num_classes; ++c) oh[c] = ((long long)c == lab) ? 1.0f : 0.0f;
    }
    gt_coord[o] = encode_box(a, s_gt[arg], true);                     // data.py:252-259
    if (best_out) best_out[o] = arg;
    if (max_iou) max_iou[o] = best;
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_rpn_sample(const float* anchors, int H, int W, int A, const float* gt, int B,
                                int G, const int32_t* best_gt, const uint8_t* flags,
                                const uint32_t* keys_pos, const uint32_t* keys_neg,
                                const uint32_t* keys_all, uint64_t seed, int image_base, int batch,
                                int max_pos, float* batch_label, float* batch_anchor,
                                float* batch_gt, int64_t* gt_list, int64_t* batch_pos,
                                float* batch_reg_gt, int32_t* counts, frcnn_stream_t stream) {
    if (!anchors || !gt || !best_gt || !flags || !batch_label || !batch_anchor || !batch_gt ||
        !gt_list || !batch_pos || !batch_reg_gt || !counts)
        return FRCNN_E_BADARG;
    if (H <= 0 || W <= 0 || A <= 0 || B <= 0 || G <= 0 || batch <= 0 || max_pos < 0 ||
        max_pos > batch)
        return FRCNN_E_BADARG;
    const bool any = keys_pos || keys_neg || keys_all;
    const bool all = keys_pos && keys_neg && keys_all;
    if (any && !all) return FRCNN_E_BADARG;
    const long long N = (long long)H * W * A;
    if (N > (1ll << 24) || batch > SMAX_BATCH) return FRCNN_E_UNSUPPORTED;
#define FRCNN_RPN_SAMPLE(BLK, SMX)                                                                    \
    FRCNN_LAUNCH((rpn_sample_kernel<BLK, SMX>), dim3(B), dim3(BLK), 0, (hipStream_t)stream,             \
                 reinterpret_cast<const float4*>(anchors), (int)N, W, A,                         \
                 reinterpret_cast<const float4*>(gt), G, best_gt, flags, keys_pos, keys_neg,     \
                 keys_all, seed, image_base, batch, max_pos, batch_label,                        \
                 reinterpret_cast<float4*>(batch_anchor), reinterpret_cast<float4*>(batch_gt),   \
                 reinterpret_cast<long long*>(gt_list), reinterpret_cast<long long*>(batch_pos), \
                 reinterpret_cast<float4*>(batch_reg_gt), counts)
    if (B < SBLOCK_SWITCH) {
        if (batch <= SMALL_BATCH) FRCNN_RPN_SAMPLE(SBLOCK_BIG, SMALL_BATCH); else FRCNN_RPN_SAMPLE(SBLOCK_BIG, SMAX_BATCH);
    } else {
        if (batch <= SMALL_BATCH) FRCNN_RPN_SAMPLE(SBLOCK_MID, SMALL_BATCH);
        else FRCNN_RPN_SAMPLE(SBLOCK_MID, SMAX_BATCH);
    }
#undef FRCNN_RPN_SAMPLE
    return FRCNN_OK;
}

extern "C" int frcnn_sample_proposals(const int64_t* p_label, int B, int R, int bg_label,
                                      const uint32_t* keys_pos, const uint32_t* keys_neg,
                                      const uint32_t* keys_all, uint64_t seed, int image_base,
                                      int batch, int max_pos, int64_t* tindex, int32_t* counts,
                                      frcnn_stream_t stream) {
    if (!p_label || !tindex || !counts || B <= 0 || R <= 0 || batch <= 0 || max_pos < 0 ||
        max_pos > batch)
        return FRCNN_E_BADARG;
    const bool any = keys_pos || keys_neg || keys_all;
    const bool all = keys_pos && keys_neg && keys_all;
    if (any && !all) return FRCNN_E_BADARG;
    if (R > (1 << 24) || batch > SMAX_BATCH) return FRCNN_E_UNSUPPORTED;
#define FRCNN_HEAD_SAMPLE(BLK, SMX)                                                                   \
    FRCNN_LAUNCH((sample_proposals_kernel<BLK, SMX>), dim3(B), dim3(BLK), 0, (hipStream_t)stream,       \
                 reinterpret_cast<const long long*>(p_label), R, bg_label, keys_pos, keys_neg,   \
                 keys_all, seed, image_base, batch, max_pos,                                     \
                 reinterpret_cast<long long*>(tindex), counts)
    if (B < SBLOCK_SWITCH) {
        if (batch <= SMALL_BATCH) FRCNN_HEAD_SAMPLE(SBLOCK_BIG, SMALL_BATCH); else FRCNN_HEAD_SAMPLE(SBLOCK_BIG, SMAX_BATCH);
    } else {
        if (batch <= SMALL_BATCH) FRCNN_HEAD_SAMPLE(SBLOCK_MID, SMALL_BATCH);
        else FRCNN_HEAD_SAMPLE(SBLOCK_MID, SMAX_BATCH);
    }
#undef FRCNN_HEAD_SAMPLE
    return FRCNN_OK;
}

extern "C" int frcnn_label_proposals(const float* proposed, const float* gt,
                                     const int64_t* gt_label_in, int B, int R, int G, float fg_thr,
                                     float iou_floor, int bg_label, int num_classes,
                                     int64_t* p_label, int32_t* gt_mask, float* gt_onehot,
                                     float* gt_coord, int32_t* best_gt, float* max_iou,
                                     frcnn_stream_t stream) {
    if (!proposed || !gt || !gt_label_in || !p_label || !gt_mask || !gt_coord || B <= 0 ||
        R <= 0 || G <= 0 || num_classes <= 0)
        return FRCNN_E_BADARG;
    if (G > 2048 || B > 65535) return FRCNN_E_UNSUPPORTED;
    const size_t lds = (size_t)G * (16 + 8 + 4);
    FRCNN_LAUNCH(label_proposals_kernel, dim3((R + FRCNN_TARGET_BLOCK - 1) / FRCNN_TARGET_BLOCK, B),
                 dim3(FRCNN_TARGET_BLOCK), lds,
                       (hipStream_t)stream, reinterpret_cast<const float4*>(proposed),
                       reinterpret_cast<const float4*>(gt),
                       reinterpret_cast<const long long*>(gt_label_in), R, G, fg_thr, iou_floor,
                       bg_label, num_classes, reinterpret_cast<long long*>(p_label), gt_mask,
                       gt_onehot, reinterpret_cast<float4*>(gt_coord), best_gt, max_iou);
    return FRCNN_OK;
}
