// Fused bilinear RoI crop + max-pool, forward, "band-resident" formulation for batches.
//
// A per-RoI kernel fetches a RoI's feature cells again for every RoI and channel slice (several
// times more bytes than the RoI produces, through a vector-memory path of 64 B/clk per CU) and
// pays the box geometry per wave.  Here the feature map enters a CU once: a workgroup owns
// (image, 32-channel group, band of feature rows), copies its band into LDS (119 KB for a
// 37x62 map) and then serves EVERY RoI of the image from there.  The only vector-memory traffic
// left is the pooled output (128 B per bin and group) and a 272-byte descriptor per RoI and
// workgroup; measured HBM traffic is 1.015x the algorithmic bytes.
//
// Bands overlap by the largest number of rows an output row of an in-image box can span, so
// every output row lies wholly inside one band; that band's workgroup produces it.  Output rows
// that fit no band (boxes reaching beyond the image, so that samples are further apart) are
// produced by band 0's workgroups straight from global memory.
//
// The per-RoI geometry (tf.image.crop_and_resize's sample coordinates, module/model.py:54) is
// computed once per RoI by a small kernel into the descriptor: per sample row the two feature
// rows, the weight and what the walk does with its two lerped-row register sets (reuse both /
// shift / read both / row outside the image), per sample column the two cells and the weight,
// per output row its band.
//
// What the main kernel is built around (each item measured on the 600x1000 workload, 64 images):
//  * its 16 waves claim RoIs four at a time from an LDS counter (a fixed share per wave left
//    a tenth of the workgroup's time idle);
//  * the sample-row step is one assembly statement: conditional refills of the row sets become
//    branches the compiler cannot answer with register copies (it added ~20 v_mov_b64 per row to
//    every C++ formulation), the LDS reads of a row are issued a row ahead into pinned registers;
//  * workgroup n runs on XCD n % 8, and the 16 channel groups of one (image, band) go to one XCD;
//  * 96 VGPRs and 137 KB of LDS: one wave per SIMD and 23 KB of LDS stay free for the small
//    kernels of other streams, which otherwise keep a band workgroup waiting for a whole CU.
// It is bound by its stores and by vector-ALU issue (67 % busy), about level, at the 2.0 GHz the
// chip holds while 12.85 GB of stores are in flight: the stores alone take 2.3 ms, the vector
// instructions 1.75 ms at 2.4 GHz, together 3.16 ms.
//
// Arithmetic and its order are those of frcnn_roi.hip (top = TL + (TR-TL)*lx; bottom likewise;
// out = top + (bottom-top)*ly; first maximum wins).  gfx950, wave64, -ffp-contract=off.
#include "frcnn_common.h"
#include <cstddef>

namespace frcnn {

constexpr int BD_MAXCROP = 16;                 // samples per axis
constexpr int BD_CG = 32;                      // channels per workgroup
constexpr int BD_CELL = BD_CG * 4;             // bytes of one cell's channel group
constexpr int BD_WAVES = 16;
constexpr int BD_RING = 4;                     // descriptors a wave keeps in LDS at a time
constexpr int BD_PIECES = BD_RING / 4;         // 16-byte descriptor pieces a lane moves per claim
static_assert(BD_RING % 4 == 0, "a wave moves four descriptors (64 pieces) per instruction");
constexpr int BD_ENTRY = 256;                  // bytes of a descriptor's y and x parts
constexpr int BD_LDS_TOTAL = 160 * 1024;
constexpr int BD_LDS_RING = BD_WAVES * BD_RING * BD_ENTRY;
constexpr int BD_LDS_BUDGET = BD_LDS_TOTAL - BD_LDS_RING - 64;      // bytes left for the band
constexpr uint32_t BD_DIRECT = 255, BD_NONE = 254;

// The walk keeps two lerped feature rows in two register sets A and B.  Per sample row the
// descriptor says what the walk does: REUSE (same floor and ceil rows as the sample row
// before), SHIFT (its floor row is the previous ceil row: one new row) or BOTH (two new rows),
// and into which set the ceil row goes (`par`: the sets swap roles whenever exactly one row
// is added, so nothing is ever moved between registers).
// y[2s] = lo | hi << 12 | ok << 24 | par << 25 | action << 26, y[2s+1] = weight bits
// x[2s] = lo | hi << 12 | ok << 24 | RoI flags << 25, x[2s+1] = weight bits   (sample column s)
// band byte p = band of output row p (BD_DIRECT: no band holds it, BD_NONE: no such row)
constexpr uint32_t BD_REUSE = 0, BD_SHIFT = 1, BD_BOTH = 2;
struct RoiDesc { uint32_t y[2 * BD_MAXCROP]; uint32_t x[2 * BD_MAXCROP]; uint32_t band[4]; };
static_assert(sizeof(RoiDesc) == 272 && sizeof(RoiDesc) % 16 == 0, "descriptor layout");
static_assert(offsetof(RoiDesc, band) == BD_ENTRY, "ring entries are the y and x parts");

// Behind the descriptors the workspace carries a 16-byte header: a magic word and a key made of
// every dimension the descriptors depend on.  The geometry kernel writes it, the band kernel
// compares it with the key of its own launch and returns without reading a descriptor when they
// differ (descriptors of another call, or never written: their row / column words would
// otherwise become wild addresses).
constexpr uint32_t BD_MAGIC = 0x42444831u;
struct BandHeader { uint32_t magic, key, pad0, pad1; };

static uint32_t band_key(int B, int R, int H, int W, int ch, int cw, int pool, int nb, int S, int T) {
    const int v[10] = {B, R, H, W, ch, cw, pool, nb, S, T};
    uint32_t h = 2166136261u;                    // FNV-1a over the ten values
    for (int i = 0; i < 10; ++i)
        for (int k = 0; k < 4; ++k) { h ^= ((uint32_t)v[i] >> (8 * k)) & 0xFFu; h *= 16777619u; }
    return h;
}

__global__ void __launch_bounds__(256)
roi_band_prep_kernel(const float4* __restrict__ rois, int n_roi, int H, int W, int ch, int cw,
                     int pool, int nb, int S, int T, RoiDesc* __restrict__ desc, uint32_t key) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_roi) return;
    if (i == 0) {
        BandHeader* h = reinterpret_cast<BandHeader*>(desc + n_roi);
        h->magic = BD_MAGIC; h->key = key; h->pad0 = 0u; h->pad1 = 0u;
    }
    const float4 box = rois[i];
    RoiDesc* d = desc + i;
    // pass 1: the band of every output row
    int rmin = 1 << 30, rmax = -1;
    uint32_t bandw[4] = {0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu};
    for (int s = 0; s < ch; ++s) {
        int lo = 0, hi = 0, ok = 0;
        float fr = 0.f;
        axis_sample(box.x, box.z, H, ch, s, lo, hi, fr, ok);
        if (ok) { rmin = min(rmin, min(lo, hi)); rmax = max(rmax, max(lo, hi)); }
        if (s % pool == pool - 1) {              // output row p is complete
            const int p = s / pool;
            uint32_t band = 0;                   // rows of zeros only: band 0 writes them
            if (rmax >= 0) {
                const int k = min(rmin / S, nb - 1);
                band = (rmin >= k * S && rmax < min(k * S + T, H)) ? (uint32_t)k : BD_DIRECT;
            }
            bandw[p >> 2] = (bandw[p >> 2] & ~(0xFFu << (8 * (p & 3)))) | (band << (8 * (p & 3)));
            rmin = 1 << 30; rmax = -1;
        }
    }
    // pass 2: what the walk does per sample row.  The walk of a band starts afresh (reads both
    // rows) at the first sample row of an output row whose band differs from the band of the
    // output row before it; a sample row outside the image reads nothing.
    int st_lo = -1, st_hi = -1;                  // rows of the previous sample row inside the image
    uint32_t par = 0;                            // set that holds st_hi
    uint32_t prev_band = BD_NONE;
    for (int s = 0; s < BD_MAXCROP; ++s) {
        int lo = 0, hi = 0, ok = 0;
        float fr = 0.f;
        if (s < ch) axis_sample(box.x, box.z, H, ch, s, lo, hi, fr, ok);
        if (s < ch && s % pool == 0) {
            const int p = s / pool;
            const uint32_t band = (bandw[p >> 2] >> (8 * (p & 3))) & 0xFFu;
            if (band != prev_band) { st_lo = -1; st_hi = -1; }
            prev_band = band;
        }
        uint32_t act = BD_REUSE;
        if (ok) {
            act = BD_BOTH;
            if (lo == st_lo && hi == st_hi) act = BD_REUSE;
            else if (lo == st_hi) { act = BD_SHIFT; par ^= 1u; }
            st_lo = lo; st_hi = hi;
        }
        d->y[2 * s] = (uint32_t)lo | ((uint32_t)hi << 12) | ((uint32_t)ok << 24) | (par << 25) |
                      (act << 26);
        d->y[2 * s + 1] = __float_as_uint(fr);
    }
    uint32_t xflags = 1u << 26;                  // bit 25: a column outside, bit 26: hi = lo + 1 throughout
    for (int s = 0; s < cw; ++s) {
        int lo = 0, hi = 0, ok = 0;
        float fr = 0.f;
        axis_sample(box.y, box.w, W, cw, s, lo, hi, fr, ok);
        if (!ok) xflags |= 1u << 25;
        else if (hi != lo + 1) xflags &= ~(1u << 26);
    }
    for (int s = 0; s < BD_MAXCROP; ++s) {
        int lo = 0, hi = 0, ok = 0;
        float fr = 0.f;
        if (s < cw) axis_sample(box.y, box.w, W, cw, s, lo, hi, fr, ok);
        d->x[2 * s] = (uint32_t)lo | ((uint32_t)hi << 12) | ((uint32_t)ok << 24) | xflags;
        d->x[2 * s + 1] = __float_as_uint(fr);
    }
    for (int j = 0; j < 4; ++j) d->band[j] = bandw[j];
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Quad { f32x2 a, b; };                    // four channels as two packed pairs

// p + (q - p) * w, three packed instructions per pair (each rounds once, in this order)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ Quad q_lerp(Quad p, Quad q, float w) {
    const f32x2 w2 = {w, w};
    Quad r;
    r.a = p.a + pk_sub(q.a, p.a) * w2;
    r.b = p.b + pk_sub(q.b, p.b) * w2;
    return r;
}
__device__ __forceinline__ Quad q_from(f32x4 t) {
    Quad r;
    r.a = __builtin_shufflevector(t, t, 0, 1);
    r.b = __builtin_shufflevector(t, t, 2, 3);
    return r;
}
__device__ __forceinline__ Quad q_zero() {
    Quad r;
    r.a = f32x2{0.f, 0.f}; r.b = r.a;
    return r;
}
__device__ __forceinline__ f32x4 q_pack(Quad v) { return f32x4{v.a.x, v.a.y, v.b.x, v.b.y}; }

__device__ __forceinline__ uint32_t bd_rl(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t bd_first(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// One sample row of the walk, written in assembly: the conditional refills of the two row sets
// are branches inside one statement whose set operands are tied ("+v"), so the compiler sees
// straight-line code and cannot answer the conditional writes with register copies (it did:
// about twenty v_mov_b64 per sample row in every C++ formulation of this step).
// v[64:79] and v[80:95] are the staging registers of the four 16-byte LDS reads of the
// ceil and the floor feature row.
// Dependent packed instructions are kept one instruction apart (gfx940-family forwarding
// hazard of instructions that write through op_sel / packed destinations).
// staging registers R0..R0+15 <- the four 16-byte cells of feature row OFF (LDS byte offset)
#define BD_ISSUE(OFF, R0, R1, R2, R3, R0A, R1A, R2A, R3A)                                    \
    "v_add_u32 v" R0A ", " OFF ", %[a0]\n"                                                   \
    "v_add_u32 v" R1A ", " OFF ", %[a1]\n"                                                   \
    "v_add_u32 v" R2A ", " OFF ", %[a2]\n"                                                   \
    "v_add_u32 v" R3A ", " OFF ", %[a3]\n"                                                   \
    "ds_read_b128 v[" R0 "], v" R0A "\n"                                                     \
    "ds_read_b128 v[" R1 "], v" R1A "\n"                                                     \
    "ds_read_b128 v[" R2 "], v" R2A "\n"                                                     \
    "ds_read_b128 v[" R3 "], v" R3A "\n"
#define BD_ISSUE_LO(OFF) BD_ISSUE(OFF, "80:83", "84:87", "88:91", "92:95", "80", "84", "88", "92")
#define BD_ISSUE_HI(OFF) BD_ISSUE(OFF, "64:67", "68:71", "72:75", "76:79", "64", "68", "72", "76")
// the same for a RoI whose every sample column has its right cell next to its left cell (all
// but boxes with an edge exactly on a feature column): two address adds instead of four, the
// right cells through the 128-byte immediate offset.  The address sits in the SECOND tuple's
// first register, which only that tuple's own read overwrites.
#define BD_ISSUE_REG(OFF, R0, R1, R2, R3, R1A, R3A)                                          \
    "v_add_u32 v" R1A ", " OFF ", %[a0]\n"                                                   \
    "v_add_u32 v" R3A ", " OFF ", %[a2]\n"                                                   \
    "ds_read_b128 v[" R0 "], v" R1A "\n"                                                     \
    "ds_read_b128 v[" R1 "], v" R1A " offset:128\n"                                          \
    "ds_read_b128 v[" R2 "], v" R3A "\n"                                                     \
    "ds_read_b128 v[" R3 "], v" R3A " offset:128\n"
#define BD_ISSUE_LO_REG(OFF) BD_ISSUE_REG(OFF, "80:83", "84:87", "88:91", "92:95", "84", "92")
#define BD_ISSUE_HI_REG(OFF) BD_ISSUE_REG(OFF, "64:67", "68:71", "72:75", "76:79", "68", "76")
// set S <- the x-lerp of a staged row: four independent chains, dependent instructions four apart
#define BD_XLERP(S, C0L, C0H, C1L, C1H, C2L, C2H, C3L, C3H)                                  \
    "v_pk_add_f32 v[" C1L "], v[" C1L "], v[" C0L "] neg_lo:[0,1] neg_hi:[0,1]\n"            \
    "v_pk_add_f32 v[" C1H "], v[" C1H "], v[" C0H "] neg_lo:[0,1] neg_hi:[0,1]\n"            \
    "v_pk_add_f32 v[" C3L "], v[" C3L "], v[" C2L "] neg_lo:[0,1] neg_hi:[0,1]\n"            \
    "v_pk_add_f32 v[" C3H "], v[" C3H "], v[" C2H "] neg_lo:[0,1] neg_hi:[0,1]\n"            \
    "v_pk_mul_f32 v[" C1L "], %[fx0], v[" C1L "]\n"                                          \
    "v_pk_mul_f32 v[" C1H "], %[fx0], v[" C1H "]\n"                                          \
    "v_pk_mul_f32 v[" C3L "], %[fx1], v[" C3L "]\n"                                          \
    "v_pk_mul_f32 v[" C3H "], %[fx1], v[" C3H "]\n"                                          \
    "v_pk_add_f32 %[" S "0], v[" C0L "], v[" C1L "]\n"                                       \
    "v_pk_add_f32 %[" S "1], v[" C0H "], v[" C1H "]\n"                                       \
    "v_pk_add_f32 %[" S "2], v[" C2L "], v[" C3L "]\n"                                       \
    "v_pk_add_f32 %[" S "3], v[" C2H "], v[" C3H "]\n"
#define BD_XLERP_LO(S) BD_XLERP(S, "80:81", "82:83", "84:85", "86:87", "88:89", "90:91", "92:93", "94:95")
#define BD_XLERP_HI(S) BD_XLERP(S, "64:65", "66:67", "68:69", "70:71", "72:73", "74:75", "76:77", "78:79")
// v = L + (H - L) * fy for the four pairs
#define BD_YLERP(L, H)                                                                       \
    "v_pk_add_f32 %[v0a], %[" H "0], %[" L "0] neg_lo:[0,1] neg_hi:[0,1]\n"                        \
    "v_pk_add_f32 %[v0b], %[" H "1], %[" L "1] neg_lo:[0,1] neg_hi:[0,1]\n"                        \
    "v_pk_add_f32 %[v1a], %[" H "2], %[" L "2] neg_lo:[0,1] neg_hi:[0,1]\n"                        \
    "v_pk_add_f32 %[v1b], %[" H "3], %[" L "3] neg_lo:[0,1] neg_hi:[0,1]\n"                        \
    "v_pk_mul_f32 %[v0a], %[fy], %[v0a] op_sel_hi:[0,1]\n"                                   \
    "v_pk_mul_f32 %[v0b], %[fy], %[v0b] op_sel_hi:[0,1]\n"                                   \
    "v_pk_mul_f32 %[v1a], %[fy], %[v1a] op_sel_hi:[0,1]\n"                                   \
    "v_pk_mul_f32 %[v1b], %[fy], %[v1b] op_sel_hi:[0,1]\n"                                   \
    "v_pk_add_f32 %[v0a], %[" L "0], %[v0a]\n"                                                  \
    "v_pk_add_f32 %[v0b], %[" L "1], %[v0b]\n"                                                  \
    "v_pk_add_f32 %[v1a], %[" L "2], %[v1a]\n"                                                  \
    "v_pk_add_f32 %[v1b], %[" L "3], %[v1b]\n"
// bit 27 of the row word: both rows are new; bit 26: the ceil row is new; bit 25: the ceil row
// lives in set A; bit 24 clear: the sample row lies outside the image (zeros, nothing read).
// The reads of a sample row are issued one row ahead (BD_ISSUE_NEXT, from the NEXT row's word
// %[ywn] and row offsets) right after the staged cells of the current row have been consumed,
// so their latency passes behind the current row's y-lerp, its pooling and the next row's
// scalar decode.  The staging registers are operands pinned to v[64:95]: they stay live from
// one statement to the next.
#define BD_ISSUE_NEXT3(TAG, LO, HI)                                                          \
    "s_bitcmp1_b32 %[ywn], 27\n"                                                             \
    "s_cbranch_scc0 .Lbd_n1" TAG "_%=\n"                                                     \
    LO("%[lon]")                                                                             \
    HI("%[hin]")                                                                             \
    "s_branch .Lbd_n0" TAG "_%=\n"                                                           \
    ".Lbd_n1" TAG "_%=:\n"                                                                   \
    "s_bitcmp1_b32 %[ywn], 26\n"                                                             \
    "s_cbranch_scc0 .Lbd_n0" TAG "_%=\n"                                                     \
    HI("%[hin]")                                                                             \
    ".Lbd_n0" TAG "_%=:\n"
#define BD_ISSUE_NEXT(TAG)                                                                   \
    "s_cmp_lg_u32 %[xreg], 0\n"                                                              \
    "s_cbranch_scc1 .Lbd_nr" TAG "_%=\n"                                                     \
    BD_ISSUE_NEXT3(TAG "i", BD_ISSUE_LO, BD_ISSUE_HI)                                        \
    "s_branch .Lbd_ne" TAG "_%=\n"                                                           \
    ".Lbd_nr" TAG "_%=:\n"                                                                   \
    BD_ISSUE_NEXT3(TAG "r", BD_ISSUE_LO_REG, BD_ISSUE_HI_REG)                                \
    ".Lbd_ne" TAG "_%=:\n"
#define BD_ROWSTEP_HALF(TAG, SETLO, SETHI)                                                   \
    "s_bitcmp1_b32 %[yw], 27\n"                                                              \
    "s_cbranch_scc0 .Lbd_one" TAG "_%=\n"                                                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                 \
    BD_XLERP_LO(SETLO)                                                                       \
    BD_XLERP_HI(SETHI)                                                                       \
    "s_branch .Lbd_nx" TAG "_%=\n"                                                           \
    ".Lbd_one" TAG "_%=:\n"                                                                  \
    "s_bitcmp1_b32 %[yw], 26\n"                                                              \
    "s_cbranch_scc0 .Lbd_nx" TAG "_%=\n"                                                     \
    "s_waitcnt lgkmcnt(0)\n"                                                                 \
    BD_XLERP_HI(SETHI)                                                                       \
    ".Lbd_nx" TAG "_%=:\n"                                                                   \
    BD_ISSUE_NEXT(TAG)                                                                       \
    "s_bitcmp1_b32 %[yw], 24\n"                                                              \
    "s_cbranch_scc0 .Lbd_zero_%=\n"                                                          \
    BD_YLERP(SETLO, SETHI)
#define BD_STAGING(c)                                                                        \
    "+{v[64:67]}"(c[0]), "+{v[68:71]}"(c[1]), "+{v[72:75]}"(c[2]), "+{v[76:79]}"(c[3]),  \
    "+{v[80:83]}"(c[4]), "+{v[84:87]}"(c[5]), "+{v[88:91]}"(c[6]), "+{v[92:95]}"(c[7])

// The lerped feature rows a wave holds: two register sets, feature row r lives in set r & 1.
// A sample row's floor and ceil rows are neighbours, so they never compete for a set, and a
// row that the next sample row needs again (as its floor row) is already where it belongs:
// no register moves, whatever the spacing of the sample rows (or their direction).
struct RowSet { Quad h0, h1; };

template <int POOL, bool WIN>
__global__ void __launch_bounds__(64 * BD_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5)))
roi_fwd_band_kernel(const float* __restrict__ fmap, int H, int W, int C, long long sb, long long sh,
                    long long sw, const RoiDesc* __restrict__ desc, int R, int ch, int cw,
                    int ngroup, int nband, int nimg, int S, int T, int rsplit,
                    float* __restrict__ out, uint8_t* __restrict__ winner, uint32_t key) {
    extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];
    {
        // the descriptors must be those of a call with these very dimensions (uniform: a scalar load)
        const BandHeader* h = reinterpret_cast<const BandHeader*>(desc + (size_t)nimg * R);
        if (h->magic != BD_MAGIC || h->key != key) return;
    }
    // Workgroup n runs on XCD n % 8 (each XCD has its own L2).  The ngroup channel groups of one
    // (image, band) are given to one XCD at consecutive times: that L2 then sees all the 128-byte
    // pieces of an output bin (and of a descriptor) instead of an eighth of them.
    // With few images the RoIs of an image are split into ranges of `rsplit` (a multiple of the
    // claim size): workgroup (image, band, group, range), each loading the band for its share.
    int g, k, b, split;
    {
        const int J = nimg * nband, full = J & ~7;
        split = (int)blockIdx.x / (J * ngroup);
        const int n = (int)blockIdx.x - split * (J * ngroup);
        int j;
        if (n < full * ngroup) { j = 8 * ((n >> 3) / ngroup) + (n & 7); g = (n >> 3) % ngroup; }
        else { const int m = n - full * ngroup; j = full + m / ngroup; g = m % ngroup; }
        b = j / nband; k = j - b * nband;
    }
    const int r0 = k * S, r1 = min(r0 + T, H);
    const uint32_t rowpitch = (uint32_t)W * BD_CELL;
    const float* img = fmap + (size_t)b * sb + (size_t)g * BD_CG;
    // ---- the band: [row][x][32 channels]; eight 16-byte parts per thread in flight at a time
    {
        const int nparts = (r1 - r0) * W * (BD_CELL / 16);
        constexpr int U = 8;
#pragma unroll 1
        for (int base = threadIdx.x; base < nparts; base += 64 * BD_WAVES * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 64 * BD_WAVES;
                if (idx < nparts) {
                    const int cell = idx >> 3, part = idx & 7;
                    const int row = cell / W, x = cell - row * W;
                    v[u] = *reinterpret_cast<const f32x4*>(img + (size_t)(r0 + row) * sh + (size_t)x * sw +
                                                           part * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 64 * BD_WAVES;
                if (idx < nparts) *reinterpret_cast<f32x4*>(band_lds + (size_t)idx * 16) = v[u];
            }
        }
    }
    uint32_t* claim_ctr = reinterpret_cast<uint32_t*>(band_lds + (size_t)T * rowpitch + BD_LDS_RING);
    const int r_first = split * rsplit, r_end = min(R, r_first + rsplit);
    if (threadIdx.x == 0) *claim_ctr = (uint32_t)r_first;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 3, quad = lane & 7;
    const int PH = ch / POOL, PW = cw / POOL;
    const bool live0 = 2 * q < cw, live1 = 2 * q + 1 < cw;      // this lane's two sample columns
    const RoiDesc* dimg = desc + (size_t)b * R;
    const size_t bin_stride = (size_t)C;                         // floats between output columns
    // output addressing: a lane's offset inside an output row is fixed for the whole launch, the
    // row offset is uniform -- one 64-bit add per stored row, no per-row multiplies
    const size_t lane_out = (size_t)(POOL == 1 ? 2 * q : q) * bin_stride + (size_t)g * BD_CG + quad * 4;
    const size_t row_out = (size_t)PW * bin_stride;              // floats between output rows
    // byte address of the band inside the LDS (what a ds_read takes)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)band_lds;
    const uint32_t lds_adj = lds_base + (uint32_t)quad * 16u - (uint32_t)r0 * rowpitch;
    // this wave's descriptor ring (private: no barrier, LDS operations of a wave stay in order)
    unsigned char* ring = band_lds + (size_t)T * rowpitch + (size_t)wave * BD_RING * BD_ENTRY;

    // The waves of a workgroup claim the image's RoIs BD_RING at a time from a counter in LDS (a
    // fixed share per wave leaves the faster waves idle for a tenth of the workgroup's time) and
    // copy the claimed descriptors into their ring, whether or not this band has rows of them.
    // Fetching a claim ahead, or reading ring entries ahead of their visit, only cost registers:
    // the other waves of the SIMD cover these latencies.  A small ring also leaves 23 KB of the
    // CU's LDS to the workgroups of other kernels, which otherwise keep a band workgroup waiting
    // for one contiguous 154 KB block.
    const int sel = lane >> 4, part = lane & 15;   // piece t of a lane: part `part` of entry sel + 4 t
    auto claim = [&]() -> int {
        uint32_t n = 0;
        if (lane == 0)
            n = __hip_atomic_fetch_add(claim_ctr, (uint32_t)BD_RING, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)bd_first(n);
    };
    // the claimed descriptors -> this wave's ring; which output rows of each of them lie in this
    // band (lane = (descriptor, output row) compares one band byte, a ballot collects the rows:
    // bits 16 e .. 16 e + 15 of word t belong to descriptor 4 t + e)
    auto fetch = [&](int n, uint64_t (&mine)[BD_PIECES], uint64_t (&mine_direct)[BD_PIECES]) {
        uint4 piece[BD_PIECES];
        uint32_t bandbyte[BD_PIECES];
#pragma unroll
        for (int t = 0; t < BD_PIECES; ++t) {
            const int rr = n + sel + 4 * t;
            piece[t] = make_uint4(0u, 0u, 0u, 0u);
            bandbyte[t] = BD_NONE;
            if (rr < R) {
                piece[t] = *reinterpret_cast<const uint4*>(
                    reinterpret_cast<const unsigned char*>(dimg + rr) + part * 16);
                bandbyte[t] = reinterpret_cast<const unsigned char*>(dimg[rr].band)[part];
            }
        }
#pragma unroll
        for (int t = 0; t < BD_PIECES; ++t) {
            *reinterpret_cast<uint4*>(ring + (sel + 4 * t) * BD_ENTRY + part * 16) = piece[t];
            mine[t] = __ballot(bandbyte[t] == (uint32_t)k);
            mine_direct[t] = k == 0 ? __ballot(bandbyte[t] == BD_DIRECT) : 0ull;
        }
    };
    f32x4 stg[8];                                // cells of a feature row on their way to a row set
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int n_cur = claim();
#pragma unroll 1
    while (n_cur < r_end) {
        uint64_t mine[BD_PIECES], mine_direct[BD_PIECES];
        fetch(n_cur, mine, mine_direct);
        const int n_base = n_cur;
        n_cur = claim();
        {
#pragma unroll 1
            for (int vi = 0; vi < BD_RING; ++vi) {
            uint64_t mw = mine[0], dw = mine_direct[0];
#pragma unroll
            for (int t = 1; t < BD_PIECES; ++t)
                if ((vi >> 2) == t) { mw = mine[t]; dw = mine_direct[t]; }
            const uint32_t rows = (uint32_t)(mw >> (16 * (vi & 3))) & 0xFFFFu;
            const uint32_t rows_direct = (uint32_t)(dw >> (16 * (vi & 3))) & 0xFFFFu;
            const int r = n_base + vi;
            if ((rows | rows_direct) == 0u) continue;                               // uniform
            uint32_t ydv = *reinterpret_cast<const uint32_t*>(ring + vi * BD_ENTRY + (lane & 31) * 4);
            const uint4 xdv = *reinterpret_cast<const uint4*>(ring + vi * BD_ENTRY + 128 + q * 16);
            const uint32_t xlo0 = xdv.x & 0xFFFu, xhi0 = (xdv.x >> 12) & 0xFFFu;
            const uint32_t xlo1 = xdv.z & 0xFFFu, xhi1 = (xdv.z >> 12) & 0xFFFu;
            const float fx0 = __uint_as_float(xdv.y), fx1 = __uint_as_float(xdv.w);
            // two facts about the RoI's sample columns ride in every x word: bit 25 = some column
            // lies outside the image, bit 26 = every column inside has its right cell next to
            // its left cell
            const uint32_t xflags = bd_first(xdv.x) >> 25;
            const uint32_t xregular = (xflags >> 1) & 1u;
            bool ok0 = live0, ok1 = live1;
            uint64_t xout0 = 0ull, xout1 = 0ull;     // lanes whose samples are zeros
            if (xflags & 1u) {                                                       // uniform
                ok0 = live0 && ((xdv.x >> 24) & 1u); ok1 = live1 && ((xdv.z >> 24) & 1u);
                xout0 = __ballot(live0 && !ok0); xout1 = __ballot(live1 && !ok1);
            }
            const uint32_t a0 = xlo0 * BD_CELL + lds_adj, a2 = xlo1 * BD_CELL + lds_adj;
            uint32_t a1 = a0, a3 = a2;               // unused by a regular RoI
            if (!xregular) { a1 = xhi0 * BD_CELL + lds_adj; a3 = xhi1 * BD_CELL + lds_adj; }
            const size_t roi_out = ((size_t)b * R + r) * (size_t)(PH * PW) * C;        // uniform
            float* oroi = out + roi_out + lane_out;
            uint8_t* wroi = WIN ? winner + roi_out + lane_out : nullptr;
            RowSet sA, sB;
            // no initial value: the first sample row of a walk reads both of its rows
            asm volatile("" : "=v"(sA.h0.a), "=v"(sA.h0.b), "=v"(sA.h1.a), "=v"(sA.h1.b),
                              "=v"(sB.h0.a), "=v"(sB.h0.b), "=v"(sB.h1.a), "=v"(sB.h1.b));
            Quad best = q_zero();
            uchar4 arg = make_uchar4(0, 0, 0, 0);

            // one sample row's two samples folded into the running maximum / stored
            auto fold = [&](int oy, int ky, Quad v0, Quad v1) {
                if (POOL == 1) {
                    float* o = oroi + (size_t)oy * row_out;
                    if (live0) *reinterpret_cast<f32x4*>(o) = q_pack(v0);
                    if (live1) *reinterpret_cast<f32x4*>(o + bin_stride) = q_pack(v1);
                    return;
                }
                if (WIN) {
                    const Quad vs[2] = {v0, v1};
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) {
                        const int kk = ky * POOL + kx;
                        if (kk == 0) {
                            best = vs[kx];
                            arg = make_uchar4(0, 0, 0, 0);
                        } else {
                            if (vs[kx].a.x > best.a.x) { best.a.x = vs[kx].a.x; arg.x = kk; }
                            if (vs[kx].a.y > best.a.y) { best.a.y = vs[kx].a.y; arg.y = kk; }
                            if (vs[kx].b.x > best.b.x) { best.b.x = vs[kx].b.x; arg.z = kk; }
                            if (vs[kx].b.y > best.b.y) { best.b.y = vs[kx].b.y; arg.w = kk; }
                        }
                    }
                } else if (ky == 0) {
                    best.a.x = vmax2(v0.a.x, v1.a.x); best.a.y = vmax2(v0.a.y, v1.a.y);
                    best.b.x = vmax2(v0.b.x, v1.b.x); best.b.y = vmax2(v0.b.y, v1.b.y);
                } else {
                    best.a.x = vmax3(best.a.x, v0.a.x, v1.a.x); best.a.y = vmax3(best.a.y, v0.a.y, v1.a.y);
                    best.b.x = vmax3(best.b.x, v0.b.x, v1.b.x); best.b.y = vmax3(best.b.y, v0.b.y, v1.b.y);
                }
                if (ky == POOL - 1 && q < PW) {
                    const size_t o = (size_t)oy * row_out;
                    *reinterpret_cast<f32x4*>(oroi + o) = q_pack(best);
                    if (WIN) *reinterpret_cast<uchar4*>(wroi + o) = arg;
                }
            };
            const f32x2 fx0p = {fx0, fx0}, fx1p = {fx1, fx1};
            const uint32_t la0 = a0, la1 = a1, la2 = a2, la3 = a3;
            if (rows != 0u) {
                // the descriptor words are read by lane index from here on: nothing may wait on
                // the LDS counter inside the walk (the next row's reads are in flight there)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ydv));
                uint32_t rem = rows;
                int oy = __builtin_ctz(rem);
                uint32_t ywn = bd_rl(ydv, 2 * (oy * POOL));
                {
                    const uint32_t lon = (ywn & 0xFFFu) * rowpitch, hin = ((ywn >> 12) & 0xFFFu) * rowpitch;
                    asm volatile(BD_ISSUE_NEXT("p")
                                 : BD_STAGING(stg)
                                 : [a0] "v"(la0), [a1] "v"(la1), [a2] "v"(la2), [a3] "v"(la3),
                                   [ywn] "s"(ywn), [lon] "s"(lon), [hin] "s"(hin), [xreg] "s"(xregular)
                                 : "scc");
                }
#pragma unroll 1
                while (true) {
                    rem &= rem - 1u;
                    const int oyn = rem != 0u ? __builtin_ctz(rem) : 0;
#pragma unroll
                    for (int ky = 0; ky < POOL; ++ky) {
                        const int sy = oy * POOL + ky;
                        const uint32_t yw = ywn;
                        const uint64_t fy = bd_rl(ydv, 2 * sy + 1);
                        if (ky < POOL - 1) ywn = bd_rl(ydv, 2 * (sy + 1));
                        else ywn = rem != 0u ? bd_rl(ydv, 2 * (oyn * POOL)) : 0u;
                        const uint32_t lon = (ywn & 0xFFFu) * rowpitch, hin = ((ywn >> 12) & 0xFFFu) * rowpitch;
                        Quad v0, v1;
                        uint64_t exec_tmp;
                        asm volatile(
                            "s_bitcmp1_b32 %[yw], 25\n"
                            "s_cbranch_scc1 .Lbd_par1_%=\n"
                            BD_ROWSTEP_HALF("0", "A", "B")
                            "s_branch .Lbd_mask_%=\n"
                            ".Lbd_par1_%=:\n"
                            BD_ROWSTEP_HALF("1", "B", "A")
                            ".Lbd_mask_%=:\n"
                            "s_or_b64 %[tmp], %[xout0], %[xout1]\n"
                            "s_cbranch_scc0 .Lbd_end_%=\n"
                            "s_mov_b64 %[tmp], exec\n"
                            "s_and_b64 exec, %[tmp], %[xout0]\n"
                            "v_mov_b64 %[v0a], 0\n"
                            "v_mov_b64 %[v0b], 0\n"
                            "s_and_b64 exec, %[tmp], %[xout1]\n"
                            "v_mov_b64 %[v1a], 0\n"
                            "v_mov_b64 %[v1b], 0\n"
                            "s_mov_b64 exec, %[tmp]\n"
                            "s_branch .Lbd_end_%=\n"
                            ".Lbd_zero_%=:\n"
                            "v_mov_b64 %[v0a], 0\n"
                            "v_mov_b64 %[v0b], 0\n"
                            "v_mov_b64 %[v1a], 0\n"
                            "v_mov_b64 %[v1b], 0\n"
                            ".Lbd_end_%=:\n"
                            : [v0a] "=&v"(v0.a), [v0b] "=&v"(v0.b), [v1a] "=&v"(v1.a), [v1b] "=&v"(v1.b),
                              [A0] "+v"(sA.h0.a), [A1] "+v"(sA.h0.b), [A2] "+v"(sA.h1.a), [A3] "+v"(sA.h1.b),
                              [B0] "+v"(sB.h0.a), [B1] "+v"(sB.h0.b), [B2] "+v"(sB.h1.a), [B3] "+v"(sB.h1.b),
                              BD_STAGING(stg), [tmp] "=&s"(exec_tmp)
                            : [a0] "v"(la0), [a1] "v"(la1), [a2] "v"(la2), [a3] "v"(la3),
                              [fx0] "v"(fx0p), [fx1] "v"(fx1p), [yw] "s"(yw), [fy] "s"(fy),
                              [ywn] "s"(ywn), [lon] "s"(lon), [hin] "s"(hin), [xreg] "s"(xregular),
                              [xout0] "s"(xout0), [xout1] "s"(xout1)
                            : "scc");
                        fold(oy, ky, v0, v1);
                    }
                    if (rem == 0u) break;
                    oy = oyn;
                }
            }

            // output rows no band holds: every corner straight from global memory
            if (rows_direct != 0u) {
                const float* ip = img + quad * 4;
#pragma unroll 1
                for (int sy = 0; sy < ch; ++sy) {
                    if (!((rows_direct >> (sy / POOL)) & 1u)) continue;             // uniform
                    const uint32_t yw = bd_rl(ydv, 2 * sy);
                    const float fy = __uint_as_float(bd_rl(ydv, 2 * sy + 1));
                    Quad v0 = q_zero(), v1 = q_zero();
                    if ((yw >> 24) & 1u) {
                        const float* rt = ip + (size_t)(yw & 0xFFFu) * sh;
                        const float* rb = ip + (size_t)((yw >> 12) & 0xFFFu) * sh;
                        auto ld = [&](const float* rp, uint32_t x) {
                            return q_from(*reinterpret_cast<const f32x4*>(rp + (size_t)x * sw));
                        };
                        const Quad t0 = q_lerp(ld(rt, xlo0), ld(rt, xhi0), fx0);
                        const Quad b0 = q_lerp(ld(rb, xlo0), ld(rb, xhi0), fx0);
                        const Quad t1 = q_lerp(ld(rt, xlo1), ld(rt, xhi1), fx1);
                        const Quad b1 = q_lerp(ld(rb, xlo1), ld(rb, xhi1), fx1);
                        v0 = q_lerp(t0, b0, fy);
                        v1 = q_lerp(t1, b1, fy);
                        if (!ok0) v0 = q_zero();
                        if (!ok1) v1 = q_zero();
                    }
                    fold(sy / POOL, sy % POOL, v0, v1);
                }
            }
            }
        }
    }
}

// ---- [B,C,H,W] contiguous -> [B,H,W,C] contiguous, 64 x 64 tiles through LDS: reads run along
// H*W, writes along C, both 256 bytes per wave row.
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int HW, float* __restrict__ dst) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const float* s = src + (size_t)blockIdx.z * C * HW;
    float* d = dst + (size_t)blockIdx.z * C * HW;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, pp = p0 + tx;
        if (c < C && pp < HW) tile[ty + 4 * i][tx] = s[(size_t)c * HW + pp];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int pp = p0 + ty + 4 * i, c = c0 + tx;
        if (c < C && pp < HW) d[(size_t)pp * C + c] = tile[tx][ty + 4 * i];
    }
}

// rows of the bands for an H x W map and crop height ch: every output row of a box inside the
// image spans at most V + 1 feature rows, bands are T rows every S = T - V rows.
struct BandPlan { int nb, S, T; bool ok; };

static BandPlan band_plan(int H, int W, int ch, int pool) {
    BandPlan p{0, 0, 0, false};
    const long long rowpitch = (long long)W * BD_CELL;
    const int Tmax = (int)(BD_LDS_BUDGET / rowpitch);
    if (Tmax < 1) return p;
    if (Tmax >= H) { p = BandPlan{1, H, H, true}; return p; }
    // samples of one output row are (pool-1)*scale apart, scale <= (H-1)/(ch-1) inside the image
    int V = 1;
    if (pool > 1 && ch > 1) V = (int)(((long long)(pool - 1) * (H - 1) + (ch - 2)) / (ch - 1)) + 1;
    if (Tmax <= V) return p;
    const int nb = (H - V + (Tmax - V) - 1) / (Tmax - V);
    const int T = (H + (nb - 1) * V + nb - 1) / nb;
    p = BandPlan{nb, T - V, T, true};
    return p;
}

}  // namespace frcnn

using namespace frcnn;

static size_t roi_ws_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t frcnn_roi_fwd_workspace_bytes(int B, int R) {
    if (B <= 0 || R <= 0) return 0;
    return roi_ws_align((size_t)B * R * sizeof(RoiDesc) + sizeof(BandHeader));
}

// Which forward kernel a channels-last call is given (alignment and workspace aside).
// The band kernel wins once its workgroups (B * C/32 * bands, one per CU at a time) fill the
// chip about three times over: on the 600x1000 workload (C = 512, R = 2000, 3 bands) it takes
// 0.66 / 1.79 / 3.38 ms at B = 16 / 32 / 64 against 1.02 / 1.94 / 3.92 ms for the per-RoI
// kernel.  Below that size the RoIs of an image are split into ranges (band_split), which
// brings the band kernel in front from four images on (B = 8: 0.43 against 0.50 ms).  A band
// is loaded once per workgroup whatever R is: below about a thousand RoIs per image the per-RoI
// kernel is the faster one.
// FRCNN_ROI_BAND_MIN_WG = n replaces the first threshold and lifts the second (n < 0: never).
// RoI ranges per (image, band, group): with few images the band workgroups alone do not fill the
// chip, so each takes a share of the image's RoIs (at least BD_SPLIT_MIN_ROIS, so that loading the
// band once per workgroup stays small against serving them)
constexpr long long BD_BAND_MIN_WG = 768;
constexpr int BD_SPLIT_MIN_ROIS = 256;
static int band_split(long long base_wg, int R) {
    // as many ranges as bring the launch to the size from which the band kernel wins: measured on
    // the 600x1000 workload (3 bands, C = 512), per-RoI kernel / band kernel at its best split:
    // B = 8: 504 / 425 us (2 ranges), B = 4: 268 / 250 us (5), B = 2: 149 / 145, B = 1: 88 / 86
    // (a workgroup's fixed cost, loading its band, is ~45 us: no gain below four images)
    long long n = base_wg > 0 ? (BD_BAND_MIN_WG + base_wg - 1) / base_wg : 1;
    long long most = R / BD_SPLIT_MIN_ROIS;
    if (const long long t = tuning(TUNE_ROI_BAND_SPLIT); t != TUNE_UNSET && t > 0) {
        n = t;                                              // forced: down to one claim per range
        most = (R + BD_RING - 1) / BD_RING;
    }
    if (n > most) n = most;
    return n < 1 ? 1 : (int)n;
}
// RoIs per range: a multiple of the claim size
static int band_rsplit(int R, int nsplit) {
    const int per = (R + nsplit - 1) / nsplit;
    return (per + BD_RING - 1) / BD_RING * BD_RING;
}

static bool band_wanted(int B, int H, int W, int C, int R, int crop_h, int crop_w, int pool,
                        bool want_winner) {
    if (C % BD_CG || (pool != 1 && pool != 2) || crop_h % pool || crop_w % pool) return false;
    const BandPlan plan = band_plan(H, W, crop_h, pool);
    long long min_wg = BD_BAND_MIN_WG;
    int min_roi = 1024;
    if (const long long t = tuning(TUNE_ROI_BAND_MIN_WG); t != TUNE_UNSET) { min_wg = t; min_roi = 1; }
    const long long base_wg = (long long)B * (C / BD_CG) * plan.nb;
    return plan.ok && plan.nb < (int)BD_NONE && crop_h <= BD_MAXCROP && crop_w <= BD_MAXCROP && H < 4096 && W < 4096 &&
           min_wg >= 0 && base_wg * band_split(base_wg, R) >= min_wg &&
           base_wg * band_split(base_wg, R) <= 0x7FFFFFFFll && R >= min_roi &&
           !(pool == 1 && want_winner);
}

extern "C" int frcnn_roi_fwd_plan(int B, int H, int W, int C, int R, int crop_h, int crop_w, int pool,
                                  int want_winner) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || R <= 0 || crop_h <= 0 || crop_w <= 0) return 0;
    return band_wanted(B, H, W, C, R, crop_h, crop_w, pool, want_winner != 0) ? 1 : 0;
}

extern "C" int frcnn_nchw_to_nhwc(const float* src, int B, int C, int H, int W, float* dst,
                                  frcnn_stream_t stream) {
    if (!src || !dst || src == dst || B <= 0 || C <= 0 || H <= 0 || W <= 0) return FRCNN_E_BADARG;
    const long long HW = (long long)H * W;
    if (HW > 0x7FFFFFFFll || B > 65535 || (C + 63) / 64 > 65535) return FRCNN_E_UNSUPPORTED;
    FRCNN_LAUNCH(nchw_to_nhwc_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), B),
                 dim3(256), 0, (hipStream_t)stream, src, C, (int)HW, dst);
    return FRCNN_OK;
}

static int launch_band_prep(const float* rois, int B, int H, int W, int R, int crop_h, int crop_w,
                            int pool, void* workspace, hipStream_t st) {
    const BandPlan plan = band_plan(H, W, crop_h, pool);
    const int n_roi = B * R;
    FRCNN_LAUNCH(roi_band_prep_kernel, dim3((n_roi + 255) / 256), dim3(256), 0, st,
                 reinterpret_cast<const float4*>(rois), n_roi, H, W, crop_h, crop_w, pool, plan.nb,
                 plan.S, plan.T, reinterpret_cast<RoiDesc*>(workspace),
                 band_key(B, R, H, W, crop_h, crop_w, pool, plan.nb, plan.S, plan.T));
    return FRCNN_OK;
}

static bool band_workspace_ok(const void* workspace, size_t workspace_bytes, int B, int R) {
    return workspace && workspace_bytes >= frcnn_roi_fwd_workspace_bytes(B, R) &&
           (reinterpret_cast<uintptr_t>(workspace) % 16) == 0;
}

extern "C" int frcnn_roi_fwd_prepare(const float* rois, int B, int H, int W, int C, int R, int crop_h,
                                     int crop_w, int pool, int want_winner, void* workspace,
                                     size_t workspace_bytes, frcnn_stream_t stream) {
    if (!rois || B <= 0 || H <= 0 || W <= 0 || C <= 0 || R <= 0 || crop_h <= 0 || crop_w <= 0)
        return FRCNN_E_BADARG;
    if (!band_wanted(B, H, W, C, R, crop_h, crop_w, pool, want_winner != 0) ||
        !band_workspace_ok(workspace, workspace_bytes, B, R))
        return 0;
    const int rc = launch_band_prep(rois, B, H, W, R, crop_h, crop_w, pool, workspace,
                                    (hipStream_t)stream);
    return rc == FRCNN_OK ? 1 : rc;
}

static int roi_fwd_ws_impl(const float* fmap, int B, int H, int W, int C, int64_t sb, int64_t sh,
                           int64_t sw, int64_t sc, const float* rois, int R, int crop_h, int crop_w,
                           int pool, float* out, uint8_t* winner, void* workspace,
                           size_t workspace_bytes, frcnn_stream_t stream, bool prepared) {
    if (!fmap || !rois || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || R <= 0 ||
        crop_h <= 0 || crop_w <= 0)
        return FRCNN_E_BADARG;
    if (pool != 1 && pool != 2) return FRCNN_E_BADARG;
    if (crop_h % pool || crop_w % pool) return FRCNN_E_BADARG;
    // a feature map stored [B,C,H,W] (planar): with room for a channels-last copy behind the
    // descriptors, transpose it once (two passes over 4*H*W*C bytes per image) and run the
    // channels-last kernels on the copy instead of the any-stride kernel on the original
    if (sw == 1 && sh == W && sc == (int64_t)H * W && sb == (int64_t)C * H * W && C > 1 &&
        (C % 4) == 0 && workspace && (reinterpret_cast<uintptr_t>(workspace) % 16) == 0) {
        const size_t desc_bytes = frcnn_roi_fwd_workspace_bytes(B, R);
        const size_t copy_bytes = (size_t)B * H * W * C * sizeof(float);
        if (workspace_bytes >= desc_bytes + copy_bytes) {
            float* copy = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + desc_bytes);
            const int rc = frcnn_nchw_to_nhwc(fmap, B, C, H, W, copy, stream);
            if (rc != FRCNN_OK) return rc;
            return roi_fwd_ws_impl(copy, B, H, W, C, (int64_t)H * W * C, (int64_t)W * C, C, 1, rois, R,
                                   crop_h, crop_w, pool, out, winner, workspace, desc_bytes, stream,
                                   prepared);
        }
    }
    const bool aligned = sc == 1 && (C % BD_CG) == 0 && (sb % 4) == 0 && (sh % 4) == 0 &&
                         (sw % 4) == 0 && sb >= 0 && sh >= 0 && sw >= 0 &&
                         (reinterpret_cast<uintptr_t>(fmap) % 16) == 0 &&
                         (reinterpret_cast<uintptr_t>(out) % 16) == 0 &&
                         (!winner || (reinterpret_cast<uintptr_t>(winner) % 4) == 0);
    const BandPlan plan = band_plan(H, W, crop_h, pool);
    const int ngroup = C / BD_CG;
    const bool band_path = aligned && band_wanted(B, H, W, C, R, crop_h, crop_w, pool, winner != nullptr) &&
                           band_workspace_ok(workspace, workspace_bytes, B, R);
    if (!band_path)
        return frcnn_roi_crop_pool_fwd(fmap, B, H, W, C, sb, sh, sw, sc, rois, R, crop_h, crop_w,
                                       pool, out, winner, stream);
    hipStream_t st = (hipStream_t)stream;
    RoiDesc* desc = reinterpret_cast<RoiDesc*>(workspace);
    if (!prepared) {
        const int rc = launch_band_prep(rois, B, H, W, R, crop_h, crop_w, pool, workspace, st);
        if (rc != FRCNN_OK) return rc;
    }
    const size_t lds = (size_t)plan.T * W * BD_CELL + BD_LDS_RING + 64;    // + the claim counter
    const int nsplit = band_split((long long)ngroup * plan.nb * B, R), rsplit = band_rsplit(R, nsplit);
    const dim3 grid((unsigned)((long long)ngroup * plan.nb * B * nsplit)), block(64 * BD_WAVES);
    const uint32_t key = band_key(B, R, H, W, crop_h, crop_w, pool, plan.nb, plan.S, plan.T);
#define FRCNN_BAND(P, WN)                                                                          \
    do {                                                                                           \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(roi_fwd_band_kernel<P, WN>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return FRCNN_E_LAUNCH;                                                                 \
        FRCNN_LAUNCH((roi_fwd_band_kernel<P, WN>), grid, block, lds, st, fmap, H, W, C,            \
                     (long long)sb, (long long)sh, (long long)sw, desc, R, crop_h, crop_w, ngroup, plan.nb, B, \
                     plan.S, plan.T, rsplit, out, winner, key);                                    \
    } while (0)
    if (pool == 2) { if (winner) FRCNN_BAND(2, true); else FRCNN_BAND(2, false); }
    else           { FRCNN_BAND(1, false); }
#undef FRCNN_BAND
    return FRCNN_OK;
}

extern "C" int frcnn_roi_crop_pool_fwd_ws(const float* fmap, int B, int H, int W, int C, int64_t sb,
                                          int64_t sh, int64_t sw, int64_t sc, const float* rois,
                                          int R, int crop_h, int crop_w, int pool, float* out,
                                          uint8_t* winner, void* workspace, size_t workspace_bytes,
                                          frcnn_stream_t stream) {
    return roi_fwd_ws_impl(fmap, B, H, W, C, sb, sh, sw, sc, rois, R, crop_h, crop_w, pool, out,
                           winner, workspace, workspace_bytes, stream, false);
}

extern "C" int frcnn_roi_crop_pool_fwd_prepared(const float* fmap, int B, int H, int W, int C,
                                                int64_t sb, int64_t sh, int64_t sw, int64_t sc,
                                                const float* rois, int R, int crop_h, int crop_w,
                                                int pool, float* out, uint8_t* winner,
                                                void* workspace, size_t workspace_bytes,
                                                frcnn_stream_t stream) {
    return roi_fwd_ws_impl(fmap, B, H, W, C, sb, sh, sw, sc, rois, R, crop_h, crop_w, pool, out,
                           winner, workspace, workspace_bytes, stream, true);
}
