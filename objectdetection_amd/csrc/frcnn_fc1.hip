// The head's first dense layer on the pooled RoI tensor: out = relu(x W + b), float32 in, float32
// accumulate, on the matrix cores (Faster_RCNN/module/model.py:31-32: TimeDistributed(Flatten)
// followed by TimeDistributed(Dense(4096, relu)); x is the pooled tensor in the (h, w, c) order
// the RoI kernels write, so Flatten is a view and no transpose sits between the two).
//
// gfx950 has float32-input MFMA (v_mfma_f32_32x32x2_f32: 32x32 outputs, two k per instruction,
// 64 cycles per SIMD = 64 FLOP/clk/SIMD, the float32 vector rate).  Its result is a k-ordered
// fmaf chain with one rounding per product, so with one accumulator per output element and k
// walked upwards the GEMM is bit-identical to
//     acc = 0; for k = 0 .. K-1: acc = fmaf(x[m][k], w[k][n], acc); out = max(acc + b[n], 0)
// which is the layer's defined summation order.  There is no split over k for that reason.
//
// Workgroup = 256 x 128 outputs, four waves of 128 x 64 (4 x 2 MFMA blocks, 128 accumulator
// registers); k in steps of 16 through two LDS buffers, the next step's tiles loaded to registers
// while the current one is multiplied.  x rows are read as 64-byte pieces (4 lanes per row) and
// transposed on their way into LDS (As[k][m], row stride BM + 1 floats); W rows are copied as they
// are.  One fragment read per operand block and k pair: 6 LDS reads per 8 MFMAs (512 cycles),
// issued one k pair ahead of the MFMAs that use them (the MFMAs of a pair take a few hundred
// cycles to issue, so a read issued behind them and waited for right away stalled the wave:
// 117 -> 125 TFLOP/s; the 4 x 2 wave tile instead of 2 x 2: -> 127).  Measured the same or slower:
// three or four workgroups per compute unit, loads two k steps ahead, v_mfma_f32_16x16x4_f32
// (bit-identical, 114 TFLOP/s).
//
// Workgroup n runs on XCD n % 8.  Each XCD works through 8 x 8 super-tiles of workgroup tiles
// (the 64 workgroups resident on its 32 CUs at two per CU): inside one, every x tile and every W
// tile is read from HBM once and from that XCD's L2 seven more times.
#include "frcnn_common.h"

namespace frcnn {

typedef float fc_f32x4 __attribute__((ext_vector_type(4)));
typedef float fc_f32x16 __attribute__((ext_vector_type(16)));

#ifndef FRCNN_FC_BK
#define FRCNN_FC_BK 16
#endif
#ifndef FRCNN_FC_WGS
#define FRCNN_FC_WGS 2
#endif
#ifndef FRCNN_FC_MB
#define FRCNN_FC_MB 4
#endif
constexpr int FC_MB = FRCNN_FC_MB, FC_NB = 2;   // 32 x 32 MFMA blocks per wave (rows x columns)
constexpr int FC_BM = 2 * 32 * FC_MB, FC_BN = 2 * 32 * FC_NB, FC_BK = FRCNN_FC_BK;
constexpr int FC_NPA = FC_BM * FC_BK / 4 / 256;   // 16-byte pieces of the x tile per thread
constexpr int FC_NPB = FC_BN * FC_BK / 4 / 256;   // ... of the W tile
constexpr int FC_RP = FC_BK / 4;            // pieces per x row of a k step
constexpr int FC_AS = FC_BM + 1;            // floats between k rows of the transposed x tile
constexpr int FC_BS = FC_BN;                // floats between k rows of the W tile
constexpr int FC_SUPER = 8;                 // workgroup tiles per super-tile edge

__global__ void __launch_bounds__(256, FRCNN_FC_WGS)
fc1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
               const float* __restrict__ bias, long long M, int K, int N, int relu,
               int tiles_m, int tiles_n, float* __restrict__ out) {
    // W tiles first (16-byte stores), then the transposed x tiles
    extern __shared__ __attribute__((aligned(16))) float fc_lds[];
    float (*Bs)[FC_BK * FC_BS] = reinterpret_cast<float (*)[FC_BK * FC_BS]>(fc_lds);
    float (*As)[FC_BK * FC_AS] = reinterpret_cast<float (*)[FC_BK * FC_AS]>(fc_lds + 2 * FC_BK * FC_BS);
    int tm, tn;
    {
        const int SN = (tiles_n + FC_SUPER - 1) / FC_SUPER;
        const int SM = (tiles_m + FC_SUPER - 1) / FC_SUPER;
        const int id = (int)blockIdx.x, xcd = id & 7, local = id >> 3;
        const int super = (local >> 6) * 8 + xcd, within = local & 63;
        if (super >= SM * SN) return;
        tm = (super / SN) * FC_SUPER + (within >> 3);
        tn = (super % SN) * FC_SUPER + (within & 7);
        if (tm >= tiles_m || tn >= tiles_n) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m_base = (long long)tm * FC_BM;
    const int n_base = tn * FC_BN;

    // global -> register staging: four 16-byte pieces of each operand per thread
    // x tile: piece idx = tid + 256 i -> row idx / 8, floats 4 (idx % 8) .. +3 of the k step
    // W tile: piece idx -> k row idx / 32, floats 4 (idx % 32) .. +3
    fc_f32x4 ra[FC_NPA], rb[FC_NPB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < FC_NPA; ++i) {
            const int idx = tid + 256 * i;
            const int am = idx / FC_RP, ak = k0 + 4 * (idx % FC_RP);
            const long long gm = m_base + am;
            ra[i] = fc_f32x4{0.f, 0.f, 0.f, 0.f};
            if (gm < M && ak < K)
                ra[i] = *reinterpret_cast<const fc_f32x4*>(x + (size_t)gm * (size_t)K + ak);
        }
#pragma unroll
        for (int i = 0; i < FC_NPB; ++i) {
            const int idx = tid + 256 * i;
            const int bk = k0 + (idx >> 5), bn = n_base + 4 * (idx & 31);
            rb[i] = fc_f32x4{0.f, 0.f, 0.f, 0.f};
            if (bk < K && bn < N)
                rb[i] = *reinterpret_cast<const fc_f32x4*>(w + (size_t)bk * (size_t)N + bn);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < FC_NPA; ++i) {
            const int idx = tid + 256 * i;
            const int am = idx / FC_RP, ak = 4 * (idx % FC_RP);
            float* a = &As[buf][ak * FC_AS + am];
            a[0] = ra[i].x; a[FC_AS] = ra[i].y; a[2 * FC_AS] = ra[i].z; a[3 * FC_AS] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < FC_NPB; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<fc_f32x4*>(&Bs[buf][(idx >> 5) * FC_BS + 4 * (idx & 31)]) = rb[i];
        }
    };

    fc_f32x16 acc[FC_MB][FC_NB];
#pragma unroll
    for (int i = 0; i < FC_MB; ++i)
#pragma unroll
        for (int j = 0; j < FC_NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (K + FC_BK - 1) / FC_BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int half = lane >> 5, r32 = lane & 31;
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tiles((t + 1) * FC_BK);
        const float* as = &As[buf][half * FC_AS + wm * (32 * FC_MB) + r32];
        const float* bs = &Bs[buf][half * FC_BS + wn * (32 * FC_NB) + r32];
        // fragments one k pair ahead: the reads of pair kk + 2 are issued in front of the MFMAs
        // of pair kk, whose issue alone takes a few hundred cycles, so the wait that follows is short
        float a[FC_MB], b[FC_NB];
#pragma unroll
        for (int i = 0; i < FC_MB; ++i) a[i] = as[32 * i];
#pragma unroll
        for (int j = 0; j < FC_NB; ++j) b[j] = bs[32 * j];
#pragma unroll
        for (int kk = 0; kk < FC_BK; kk += 2) {
            float na[FC_MB], nb[FC_NB];
#pragma unroll
            for (int i = 0; i < FC_MB; ++i) na[i] = kk + 2 < FC_BK ? as[(kk + 2) * FC_AS + 32 * i] : a[i];
#pragma unroll
            for (int j = 0; j < FC_NB; ++j) nb[j] = kk + 2 < FC_BK ? bs[(kk + 2) * FC_BS + 32 * j] : b[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FC_MB; ++i)
#pragma unroll
                for (int j = 0; j < FC_NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FC_MB; ++i) a[i] = na[i];
#pragma unroll
            for (int j = 0; j < FC_NB; ++j) b[j] = nb[j];
        }
        if (t + 1 < nt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // accumulator element r of a lane: column lane & 31, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < FC_NB; ++j) {
        const int col = n_base + wn * (32 * FC_NB) + j * 32 + r32;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < FC_MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = m_base + wm * (32 * FC_MB) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= M) continue;
                float v = acc[i][j][r];
                if (bias) v = v + bv;
                if (relu) v = v > 0.f ? v : 0.f;
                out[(size_t)row * (size_t)N + col] = v;
            }
    }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_fc1_fwd(const float* x, const float* w, const float* bias, int64_t M, int K,
                             int N, int relu, float* out, frcnn_stream_t stream) {
    if (!x || !w || !out || M <= 0 || K <= 0 || N <= 0) return FRCNN_E_BADARG;
    if ((K % 4) != 0 || (N % 4) != 0) return FRCNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) % 16) || (reinterpret_cast<uintptr_t>(w) % 16))
        return FRCNN_E_BADARG;
    const long long tiles_m = (M + FC_BM - 1) / FC_BM, tiles_n = (N + FC_BN - 1) / FC_BN;
    const long long SM = (tiles_m + FC_SUPER - 1) / FC_SUPER, SN = (tiles_n + FC_SUPER - 1) / FC_SUPER;
    const long long groups = (SM * SN + 7) / 8;
    const long long blocks = groups * 8 * FC_SUPER * FC_SUPER;
    if (tiles_m > 0x7FFFFF || tiles_n > 0x7FFFFF || blocks > 0x7FFFFFFFll) return FRCNN_E_UNSUPPORTED;
    constexpr size_t lds = (size_t)2 * FC_BK * (FC_AS + FC_BS) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fc1_fwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return FRCNN_E_LAUNCH;
    FRCNN_LAUNCH(fc1_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, w, bias,
                 (long long)M, K, N, relu, (int)tiles_m, (int)tiles_n, out);
    return FRCNN_OK;
}
