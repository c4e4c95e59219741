// Write-only probe of the RoI kernel's output pattern: [B][R][7][7][512] f32, 12.8 GB at
// B=64, R=2000.  How fast can it be written, as a function of which wave writes what?
// Build: hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 512, PH = 7, PW = 7;

// mode 0: wave = (roi, 64-channel slice), slice = blockIdx.x % 8; per pooled row two store
//         instructions of 4 x 256 B pieces (stride 2 KB)            [the kernel's pattern]
// mode 1: same, slice = blockIdx.x / R (a slice's blocks are dispatched together)
// mode 2: wave = (roi, pooled row): 7 bins x 2 KB contiguous = 14 store instructions of 1 KB
// mode 3: wave = (roi, 64-channel slice) but 256 threads: 4 waves, wave w takes rows w, w+4
// mode 4: workgroup = (image, 16-channel slice), 16 waves, 147 KB of LDS (one per CU); wave w
//         writes RoIs w, w+16, ...: 49 bins x 64 B, 16 bins per store instruction
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* __restrict__ out, int R, int delay) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    if (MODE == 0 || MODE == 1 || MODE == 3) {
        const int slice = MODE == 1 ? blockIdx.x / R : blockIdx.x % 8;
        const int r = MODE == 1 ? blockIdx.x % R : blockIdx.x / 8;
        const int cq = lane & 15, ql = lane >> 4;
        float* base = out + ((size_t)b * R + r) * (PH * PW) * C + slice * 64 + cq * 4;
        for (int p = (MODE == 3 ? wave : 0); p < PH; p += (MODE == 3 ? 4 : 1)) {
            for (int i = 0; i < delay; ++i) asm volatile("s_sleep 8");
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q = t * 4 + ql;
                if (q < PW) *reinterpret_cast<f32x4*>(base + (size_t)(p * PW + q) * C) = v;
            }
        }
    } else {
        const int r = blockIdx.x / PH, p = blockIdx.x % PH;
        float* base = out + (((size_t)b * R + r) * (PH * PW) + p * PW) * C + lane * 4;
#pragma unroll
        for (int i = 0; i < 14; ++i) *reinterpret_cast<f32x4*>(base + i * 256) = v;
    }
}

__global__ void __launch_bounds__(1024) probe_slice16(float* __restrict__ out, int R) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slice = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x == 0) lds[0] = 1.f;
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int r = wave; r < R; r += 16) {
        float* base = out + ((size_t)b * R + r) * (PH * PW) * C + slice * 16 + (lane & 3) * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int bin = t * 16 + (lane >> 2);
            if (bin < PH * PW) *reinterpret_cast<f32x4*>(base + (size_t)bin * C) = v;
        }
    }
}

// workgroup = (image, 32-channel group, band of feature rows), 16 waves, 127 KB of LDS (one per
// CU); wave w writes RoIs w, w+16, ...: the pooled rows that fall to its band (here: p % 3 ==
// band), 7 bins x 128 B per store instruction
__global__ void __launch_bounds__(1024) probe_band32(float* __restrict__ out, int R) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = blockIdx.x % 16, band = blockIdx.x / 16, b = blockIdx.y;
    if (threadIdx.x == 0) lds[0] = 1.f;
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    const int q = lane >> 3, quad = lane & 7;
    for (int r = wave; r < R; r += 16) {
        float* base = out + ((size_t)b * R + r) * (PH * PW) * C + grp * 32 + quad * 4;
        for (int p = band; p < PH; p += 3)
            if (q < PW) *reinterpret_cast<f32x4*>(base + (size_t)(p * PW + q) * C) = v;
    }
}

void run_band32(float* out, int B, int R) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int lds = 127 * 1024;
    CHECK(hipFuncSetAttribute((const void*)probe_band32, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid(48, B), block(1024);
    probe_band32<<<grid, block, lds>>>(out, R);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        probe_band32<<<grid, block, lds>>>(out, R);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = (double)B * R * PH * PW * C * 4;
    printf("%-64s %7.3f ms  %6.2f TB/s\n", "workgroup = image x 32ch group x band, 128 B pieces", best, bytes / best / 1e9);
}

void run_slice16(float* out, int B, int R) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int lds = 147 * 1024;
    CHECK(hipFuncSetAttribute((const void*)probe_slice16, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid(32, B), block(1024);
    probe_slice16<<<grid, block, lds>>>(out, R);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        probe_slice16<<<grid, block, lds>>>(out, R);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = (double)B * R * PH * PW * C * 4;
    printf("%-64s %7.3f ms  %6.2f TB/s\n", "workgroup = image x 16ch slice, 64 B pieces", best, bytes / best / 1e9);
}

template <int MODE>
void run(const char* name, float* out, int B, int R, int delay) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    dim3 grid(MODE == 2 ? R * PH : R * 8, B), block(MODE == 3 ? 256 : 64);
    probe<MODE><<<grid, block>>>(out, R, delay);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        probe<MODE><<<grid, block>>>(out, R, delay);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = (double)B * R * PH * PW * C * 4;
    printf("%-64s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
}

int main() {
    const int B = 64, R = 2000;
    float* out;
    CHECK(hipMalloc(&out, (size_t)B * R * PH * PW * C * 4));
    run<0>("wave = roi x 64ch, slice = block % 8 (kernel's pattern)", out, B, R, 0);
    run<0>("  same, waves paced (s_sleep between rows)", out, B, R, 4);
    run<1>("wave = roi x 64ch, slice-major blocks", out, B, R, 0);
    run<3>("4 waves per (roi, slice), rows interleaved", out, B, R, 0);
    run<2>("wave = roi x pooled row, 14 KB contiguous", out, B, R, 0);
    run_slice16(out, B, R);
    run_band32(out, B, R);
    return 0;
}
