#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
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
            : ACCOPS(acc) : [r] "s"(r), [g0] "v"(g0), [g1] "v"(g1) : "m0");
    }
    for (int i = 0; i < 10; ++i) for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][4*q], acc[i][4*q+1], acc[i][4*q+2], acc[i][4*q+3]};
        *(f32x4*)(out + (size_t)(i * 4 + q) * 256 + threadIdx.x * 4) = v;
    }
}
int main() {
    const int n = 1000;
    std::vector<float> in(n * 256), ref(40 * 256, 0.f), outh(40 * 256);
    std::vector<int> idx(n);
    srand(1);
    for (auto& v : in) v = (float)rand() / RAND_MAX - 0.5f;
    for (int j = 0; j < n; ++j) { idx[j] = rand() % 40; for (int c = 0; c < 256; ++c) ref[idx[j] * 256 + c] += in[j * 256 + c]; }
    float *din, *dout; int* didx;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dout, outh.size() * 4); hipMalloc(&didx, n * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, didx, n);
    hipMemcpy(outh.data(), dout, outh.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < outh.size(); ++i) if (outh[i] != ref[i]) ++bad;
    printf("idx-mode test: %d mismatches of %zu\n", bad, outh.size());
    return bad != 0;
}
