// Measures what a 16-byte-per-lane vector load costs on gfx950 when the data sits
// in L2, as a function of the active-lane pattern and of the address pattern.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ta_probe.hip -o tools/ta_probe
// Prints ns per wave-level load instruction per CU and the implied bytes/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
    printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: every lane loads; 16-lane groups read 256 contiguous bytes at 4 different cells
// mode 1: lanes 0..31 only          mode 2: lanes 0..15 only
// mode 3: all lanes, 1 KB contiguous
// mode 4: all lanes, every 16-lane group reads the SAME 256 bytes
// mode 5: even 16-lane groups only (0..15, 32..47)
// mode 6: LDS ds_read_b128, all lanes, conflict-free
// mode 7: direct-to-LDS 16-byte loads (1 KiB per wave instruction), 4 x 256 B segments
// mode 8: register load + ds_write_b128 of the same data
// mode 9: 4-byte loads, all lanes, one 256 B segment     mode 10: 8-byte loads, 2 x 256 B
// mode 11: direct-to-LDS 4-byte loads (256 B per wave instruction)
template <int MODE>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ src, int cells, int iters,
                                             float* __restrict__ sink) {
    __shared__ f32x4 lds[2048];
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4, cq = lane & 15;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src), 0, (uint32_t)cells * 2048u, 0x00020000);
    if (MODE == 6) {
        for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
        __syncthreads();
    }
    bool active = true;
    if (MODE == 1) active = lane < 32;
    if (MODE == 2) active = lane < 16;
    if (MODE == 5) active = (grp & 1) == 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t cell = (blockIdx.x * 7u + (threadIdx.x >> 6) * 13u) % (uint32_t)cells;
    if (active) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint32_t c = (cell + (uint32_t)u * 5u) % (uint32_t)cells;
                uint32_t voff;
                if (MODE == 3) voff = (uint32_t)lane * 16u;
                else if (MODE == 4) voff = (uint32_t)cq * 16u;
                else voff = (uint32_t)grp * 2048u * 3u + (uint32_t)cq * 16u;
                if (MODE == 6) {
                    v[u] = lds[(lane + u * 64 + it) & 1023];
                } else if (MODE == 7) {
                    uint32_t soff = c * 2048u;
                    if (c + 9 >= (uint32_t)cells) soff = 0;
                    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rs, (__attribute__((address_space(3))) void*)&lds[(w * 8 + u) * 64], 16, voff, soff, 0, 0);
                    v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                } else if (MODE == 9) {
                    uint32_t soff = c * 2048u;
                    v[u] = (f32x4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)lane * 4u, soff, 0)), 0.f, 0.f, 0.f};
                } else if (MODE == 10) {
                    uint32_t soff = c * 2048u;
                    if (c + 9 >= (uint32_t)cells) soff = 0;
                    const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)(lane >> 5) * 2048u * 3u + (uint32_t)(lane & 31) * 8u, soff, 0);
                    v[u] = (f32x4){__builtin_bit_cast(float, t[0]), __builtin_bit_cast(float, t[1]), 0.f, 0.f};
                } else if (MODE == 11) {
                    uint32_t soff = c * 2048u;
                    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rs, (__attribute__((address_space(3))) void*)&lds[(w * 8 + u) * 64], 4, (uint32_t)lane * 4u, soff, 0, 0);
                    v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                } else if (MODE == 8) {
                    uint32_t soff = c * 2048u;
                    if (c + 9 >= (uint32_t)cells) soff = 0;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
                } else {
                    uint32_t soff = c * 2048u;
                    if (MODE != 3 && MODE != 4 && c + 9 >= (uint32_t)cells) soff = 0;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
                }
            }
            if (MODE == 8) {
                const int w = threadIdx.x >> 6;
#pragma unroll
                for (int u = 0; u < 8; ++u) lds[(w * 8 + u) * 64 + lane] = v[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
            cell = (cell + 41u) % (uint32_t)cells;
        }
    }
    if (MODE == 7 || MODE == 8 || MODE == 11) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        acc += lds[threadIdx.x];
    }
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
void run(const char* name, const float* src, int cells, float* sink, int active_lanes) {
    const int iters = 400, blocks = 256 * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    probe<MODE><<<blocks, 256>>>(src, cells, 10, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    probe<MODE><<<blocks, 256>>>(src, cells, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_loads_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
    const double ns = ms * 1e6 / wave_loads_per_cu;
    const double clk = ns * 2.4;
    printf("%-46s %7.3f ms  %6.2f ns/wave-load/CU  %5.1f clk  %6.1f B/clk/CU (active lanes)\n", name,
           ms, ns, clk, active_lanes * 16.0 / clk);
}

int main() {
    const int cells = 37 * 62;           // 2 KB per cell = one image's 512-channel f32 map (4.7 MB)
    float *src, *sink;
    CHECK(hipMalloc(&src, (size_t)cells * 2048));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(src, 0, (size_t)cells * 2048));
    run<0>("all lanes, 4 x 256 B segments", src, cells, sink, 64);
    run<1>("lanes 0-31, 2 x 256 B segments", src, cells, sink, 32);
    run<2>("lanes 0-15, 1 x 256 B segment", src, cells, sink, 16);
    run<5>("lanes 0-15 + 32-47", src, cells, sink, 32);
    run<3>("all lanes, 1 KB contiguous", src, cells, sink, 64);
    run<4>("all lanes, 4 groups read the same 256 B", src, cells, sink, 64);
    run<6>("LDS ds_read_b128 all lanes", src, cells, sink, 64);
    run<7>("direct-to-LDS dwordx4, 4 x 256 B segments", src, cells, sink, 64);
    run<8>("register load + ds_write_b128", src, cells, sink, 64);
    run<9>("4-byte loads, all lanes, 256 B", src, cells, sink, 16);
    run<10>("8-byte loads, all lanes, 2 x 256 B", src, cells, sink, 32);
    run<11>("direct-to-LDS 4-byte loads, 256 B", src, cells, sink, 16);
    return 0;
}
