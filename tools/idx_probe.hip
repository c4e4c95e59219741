// VGPR index mode probes for the patch-owner RoI backward:
//   (1) correctness of relative pk_add with s_set_gpr_idx_on / _idx / _off
//   (2) cycles per sample-sized group: 8 v_pk_add_f32 on fixed registers, the same eight inside
//       one on .. off region with three index changes, and as four on .. off regions
// hipcc -O3 --offload-arch=gfx950 tools/idx_probe.hip -o tools/idx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ACCOPS(a) "+{v[96:111]}"(a[0]), "+{v[112:127]}"(a[1]), "+{v[128:143]}"(a[2]), "+{v[144:159]}"(a[3]), "+{v[160:175]}"(a[4]), \
  "+{v[176:191]}"(a[5]), "+{v[192:207]}"(a[6]), "+{v[208:223]}"(a[7]), "+{v[224:239]}"(a[8]), "+{v[240:255]}"(a[9])
__global__ void __launch_bounds__(64) k(const float* in, float* out, const int* idxs, int n) {
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int j = 0; j < n; ++j) {
        int r = __builtin_amdgcn_readfirstlane(idxs[j]) * 4;
        f32x4 g = *(const f32x4*)(in + (size_t)j * 256 + threadIdx.x * 4);
        f32x2 g0 = {g.x, g.y}, g1 = {g.z, g.w};
        asm volatile(
            "s_set_gpr_idx_on %[r], 0x9\n"
            "v_pk_add_f32 v[96:97], v[96:97], %[g0]\n"
            "v_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
            "s_set_gpr_idx_off\n"
            : ACCOPS(acc) : [r] "s"(r), [g0] "v"(g0), [g1] "v"(g1));
    }
    for (int i = 0; i < 10; ++i) for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][4*q], acc[i][4*q+1], acc[i][4*q+2], acc[i][4*q+3]};
        *(f32x4*)(out + (size_t)(i * 4 + q) * 256 + threadIdx.x * 4) = v;
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) timing(float* out, long long* cyc, int n, int i0, int i1, int i2, int i3) {
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f32x2 g0 = {1.f + threadIdx.x, 2.f}, g1 = {3.f, 4.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n; ++j) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == 0)
            asm volatile(
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "v_pk_add_f32 v[100:101], v[100:101], %[g0]\nv_pk_add_f32 v[102:103], v[102:103], %[g1]\n"
                "v_pk_add_f32 v[120:121], v[120:121], %[g0]\nv_pk_add_f32 v[122:123], v[122:123], %[g1]\n"
                "v_pk_add_f32 v[124:125], v[124:125], %[g0]\nv_pk_add_f32 v[126:127], v[126:127], %[g1]\n"
                : ACCOPS(acc) : [g0] "v"(g0), [g1] "v"(g1));
        else if (MODE == 1)
            asm volatile(
                "s_set_gpr_idx_on %[i0], 0x9\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_idx %[i1]\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_idx %[i2]\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_idx %[i3]\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_off\n"
                : ACCOPS(acc) : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [g0] "v"(g0), [g1] "v"(g1));
        else
            asm volatile(
                "s_set_gpr_idx_on %[i0], 0x9\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_off\ns_set_gpr_idx_on %[i1], 0x9\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_off\ns_set_gpr_idx_on %[i2], 0x9\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_off\ns_set_gpr_idx_on %[i3], 0x9\n"
                "v_pk_add_f32 v[96:97], v[96:97], %[g0]\nv_pk_add_f32 v[98:99], v[98:99], %[g1]\n"
                "s_set_gpr_idx_off\n"
                : ACCOPS(acc) : [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [g0] "v"(g0), [g1] "v"(g1));
      }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int n = 1000;
    std::vector<float> in(n * 256), ref(40 * 256, 0.f), outh(40 * 256);
    std::vector<int> idx(n);
    srand(1);
    for (auto& v : in) v = (float)rand() / RAND_MAX - 0.5f;
    for (int j = 0; j < n; ++j) { idx[j] = rand() % 40; for (int c = 0; c < 256; ++c) ref[idx[j] * 256 + c] += in[j * 256 + c]; }
    float *din, *dout; int* didx; long long* dcyc;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 1 << 20); hipMalloc(&didx, n * 4); hipMalloc(&dcyc, 4096);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, didx, n);
    hipMemcpy(outh.data(), dout, outh.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < outh.size(); ++i) if (outh[i] != ref[i]) ++bad;
    printf("idx-mode test: %d mismatches of %zu\n", bad, outh.size());
    const int N = 5000;                                    // x 4 groups per iteration
    for (int waves = 1; waves <= 1; ++waves) {   // the pinned v[96:255] allow one wave per SIMD
        long long c[3];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(timing<0>, dim3(1), dim3(256 * waves), 0, 0, dout, dcyc, N, 0, 4, 24, 28);
            hipMemcpy(&c[0], dcyc, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(timing<1>, dim3(1), dim3(256 * waves), 0, 0, dout, dcyc, N, 0, 4, 24, 28);
            hipMemcpy(&c[1], dcyc, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(timing<2>, dim3(1), dim3(256 * waves), 0, 0, dout, dcyc, N, 0, 4, 24, 28);
            hipMemcpy(&c[2], dcyc, 8, hipMemcpyDeviceToHost);
        }
        // s_memtime ticks per group of 8 v_pk_add_f32, divided by the waves sharing the SIMD
        printf("%d wave(s) per SIMD, ticks per group of 8 pk_add and wave: fixed %.1f, one region + 3 idx %.1f, four regions %.1f\n",
               waves, (double)c[0] / (4.0 * N * waves), (double)c[1] / (4.0 * N * waves), (double)c[2] / (4.0 * N * waves));
    }
    return bad != 0;
}
