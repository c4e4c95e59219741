// Round trip of a flag between two workgroups, on one XCD and across two, with the loads and
// stores a spin-wait protocol could use: agent-scope atomics (served past the per-XCD L2s) against
// sc0 loads + plain stores (served by the XCD's own L2; only valid between workgroups that share
// that L2).  Prints the XCC each workgroup ran on (HW_REG_XCC_ID) and microseconds per round trip.
// build: hipcc --offload-arch=gfx950 -O2 tools/xcd_pingpong.hip -o tools/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ROUNDS = 2000;
constexpr int SPINS = 1 << 22;

template <int MODE>
__device__ __forceinline__ int flag_load(int* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int v;
    if (MODE == 1)
        asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else
        asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void flag_store(int* p, int v) {
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void pingpong(int* flags, int wa, int wb, unsigned long long* out) {
    const int me = blockIdx.x;
    if (me != wa && me != wb) return;
    if (threadIdx.x != 0) return;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    int* mine = flags + (me == wa ? 0 : 64);
    int* theirs = flags + (me == wa ? 64 : 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int bad = 0;
    for (int i = 1; i <= ROUNDS && !bad; ++i) {
        if (me == wa) flag_store<MODE>(mine, i);
        int s = 0;
        while (flag_load<MODE>(theirs) < i) if (++s > SPINS) { bad = 1; break; }
        if (me == wb) flag_store<MODE>(mine, i);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = out + (me == wa ? 0 : 4);
    o[0] = t1 - t0; o[1] = xcc & 0xF; o[2] = bad;
}

template <int MODE>
static void run(const char* name, int wa, int wb, int* flags, unsigned long long* out) {
    hipMemset(flags, 0, 1024);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL(pingpong<MODE>, dim3(64), dim3(64), 0, 0, flags, wa, wb, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    unsigned long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%-28s workgroups %2d (xcc %llu) and %2d (xcc %llu): %.3f us per round trip%s\n", name, wa, h[1],
           wb, h[5], (double)h[0] / 100.0 / ROUNDS, h[2] | h[6] ? "  TIMED OUT" : "");
}

int main() {
    int* flags; unsigned long long* out;
    hipMalloc(&flags, 1024); hipMalloc(&out, 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("agent-scope atomics", 0, 1, flags, out);
        run<0>("agent-scope atomics", 0, 8, flags, out);
        run<2>("sc0 sc1 loads and stores", 0, 1, flags, out);
        run<2>("sc0 sc1 loads and stores", 0, 8, flags, out);
        run<1>("sc0 loads, plain stores", 0, 8, flags, out);
        run<1>("sc0 loads, plain stores", 0, 16, flags, out);
    }
    return 0;
}
