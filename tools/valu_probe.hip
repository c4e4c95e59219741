// Issue rate of f32 vector instructions on gfx950: plain v_mul_f32 / v_add_f32 against the
// packed forms (v_pk_mul_f32 / v_pk_add_f32), at 1, 2 and 4 waves per SIMD.
// Prints cycles per wave-instruction per SIMD (from s_memtime) for each case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f32x2 m2 = {m, m};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {          // 8 independent scalar multiplies
                a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
            } else if (MODE == 1) {   // 4 packed multiplies = the same 8 products
                p0 *= m2; p1 *= m2; p2 *= m2; p3 *= m2;
            } else if (MODE == 2) {   // lerp, scalar: 8 x (sub, mul, add)
                a0 = a1 + (a0 - a1) * m; a2 = a3 + (a2 - a3) * m;
                a4 = a5 + (a4 - a5) * m; a6 = a7 + (a6 - a7) * m;
                a1 = a0 + (a1 - a0) * m; a3 = a2 + (a3 - a2) * m;
                a5 = a4 + (a5 - a4) * m; a7 = a6 + (a7 - a6) * m;
            } else {                  // lerp, packed: 4 x (pk_sub, pk_mul, pk_add)
                p0 = p1 + (p0 - p1) * m2; p2 = p3 + (p2 - p3) * m2;
                p1 = p0 + (p1 - p0) * m2; p3 = p2 + (p3 - p2) * m2;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE & 1) { a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y; a4 = a5 = a6 = a7 = 0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int insts_per_u, int waves_per_simd) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;          // 256-thread blocks: 1 wave per SIMD each
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double n = (double)iters * 16 * insts_per_u;         // instructions per wave
    // one SIMD runs waves_per_simd waves side by side: SIMD cycles per instruction
    printf("%-22s waves/SIMD %d: %.2f cyc per wave-instr seen by one wave, %.2f SIMD-cyc per instr, "
           "kernel %.3f ms, clock %.2f GHz\n", name, waves_per_simd, mean / n,
           mean / n / waves_per_simd, ms, mean / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32 x8", 8, w);
        run<1>("v_pk_mul_f32 x4", 4, w);
        run<2>("lerp scalar (24 ops)", 24, w);
        run<3>("lerp packed (12 ops)", 12, w);
    }
    return 0;
}
