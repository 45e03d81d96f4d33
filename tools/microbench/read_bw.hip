// tools/microbench/read_bw.hip -- what a READ-ONLY stream reaches on this GPU (the ceiling of K2's constraint stream; the
// 6.3 TB/s of MI355X_MICROARCH.md is a copy: read + write).  hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4 gv4;

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n16, unsigned* out) {
    // every workgroup streams a contiguous chunk; UNROLL 1 KiB wave-loads in flight per wave
    const size_t per_block = n16 / gridDim.x;
    const uint4* base = p + (size_t)blockIdx.x * per_block;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per_block; i += (size_t)UNROLL * 256) {
        v4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load((gv4*)(base + i + u * 256)) : *(gv4*)(base + i + u * 256);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int UNROLL, bool NT>
int run(const uint4* d, size_t bytes, unsigned* out, int blocks) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 200; ++w) k_read<UNROLL, NT><<<blocks, 256>>>(d, bytes / 16, out);   // clock ramp
    CHECK(hipDeviceSynchronize());
    const int reps = 50;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_read<UNROLL, NT><<<blocks, 256>>>(d, bytes / 16, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("read-only stream %4zu MB, %5d blocks, %d x 1 KiB in flight per wave, %s: %7.1f us  %6.2f TB/s\n", bytes >> 20, blocks, UNROLL,
           NT ? "nt   " : "plain", ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) * 1e-12);
    return 0;
}

int main() {
    const size_t bytes = (size_t)1 << 30;    // 1 GiB: far outside the Infinity Cache
    uint4* d; unsigned* out;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(d, 1, bytes));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        if (run<1, false>(d, bytes, out, blocks)) return 1;
        if (run<4, false>(d, bytes, out, blocks)) return 1;
        if (run<4, true>(d, bytes, out, blocks)) return 1;
        if (run<8, true>(d, bytes, out, blocks)) return 1;
    }
    return 0;
}
