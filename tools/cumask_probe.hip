// Which CUs does a stream made with hipExtStreamCreateWithCUMask use?  Prints, for a mask with
// the first N bits set, the (XCC, SE, CU) triples that ran workgroups.
// build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void where(uint32_t* out) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // burn a little time so that the grid spreads over every CU the stream may use
    float v = (float)threadIdx.x;
    for (int i = 0; i < 20000; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xFu) << 16 | (hw & 0xFFFFu) | (v < 0.f ? 1u << 31 : 0u);
}

int main(int argc, char** argv) {
    const int nbits = argc > 1 ? atoi(argv[1]) : 16;
    const int shift = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<uint32_t> mask(8, 0u);
    for (int i = 0; i < nbits; ++i) mask[(i + shift) / 32] |= 1u << ((i + shift) % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        printf("hipExtStreamCreateWithCUMask failed\n");
        return 1;
    }
    const int n = 8192;
    uint32_t* d;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, d);
    if (hipStreamSynchronize(st) != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<uint32_t> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> cus;
    for (uint32_t v : h) {
        const uint32_t xcc = (v >> 16) & 0xF, cu = (v >> 8) & 0xF, sh = (v >> 12) & 1, se = (v >> 13) & 7;
        cus[xcc << 12 | se << 8 | sh << 4 | cu]++;
    }
    printf("mask bits [%d, %d): %zu distinct (xcc, se, sh, cu):", shift, shift + nbits, cus.size());
    for (auto& kv : cus) printf(" (%u,%u,%u,%u)x%d", kv.first >> 12, (kv.first >> 8) & 0xF, (kv.first >> 4) & 0xF, kv.first & 0xF, kv.second);
    printf("\n");
    return 0;
}
