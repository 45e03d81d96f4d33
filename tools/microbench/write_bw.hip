// tools/microbench/write_bw.hip -- what a WRITE-ONLY stream reaches on this GPU (the ceiling of k_col_direct, which only
// stores), by store pattern.  hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw
//   PAT 0: every lane stores the two 16-byte halves of ITS 32-byte element (lane stride 32 B): one store instruction covers half
//          of every 128-byte line it touches, the second instruction the other half (what fe_store does)
//   PAT 1: every lane stores 16 bytes at lane stride 16 B, twice (one store instruction = 1 KiB contiguous)
// A block writes runs of 8 KiB (256 lanes x 32 B) at a stride of 8 KiB * STEPS like k_col_direct's.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4 gv4;

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ p, unsigned steps, unsigned seed) {
    uint4* base = p + (size_t)blockIdx.x * steps * 512;           // 512 uint4 = 8 KiB per step
    v4 a = {seed, threadIdx.x, blockIdx.x, 1u}, b = {seed, threadIdx.x, blockIdx.x, 2u};
    for (unsigned s = 0; s < steps; ++s) {
        a.w += s; b.w += s;
        gv4 *q0, *q1;
        if (PAT == 0) { q0 = (gv4*)(base + (size_t)s * 512 + 2 * threadIdx.x); q1 = q0 + 1; }
        else { const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63; q0 = (gv4*)(base + (size_t)s * 512 + w * 128 + l); q1 = q0 + 64; }
        if (NT) { __builtin_nontemporal_store(a, q0); __builtin_nontemporal_store(b, q1); }
        else { *q0 = a; *q1 = b; }
    }
}

template <int PAT, bool NT>
int run(uint4* d, size_t bytes, unsigned steps) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)(bytes / ((size_t)steps * 8192));
    for (int w = 0; w < 50; ++w) k_write<PAT, NT><<<blocks, 256>>>(d, steps, w);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_write<PAT, NT><<<blocks, 256>>>(d, steps, r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("write-only %4zu MB, %6u blocks x %3u steps, %s, %s: %8.1f us  %6.2f TB/s\n", bytes >> 20, blocks, steps,
           PAT == 0 ? "lane stride 32 B (2 x 16 B of one element)" : "lane stride 16 B (1 KiB per instruction)   ", NT ? "nt   " : "plain",
           ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) * 1e-12);
    return 0;
}

int main() {
    const size_t bytes = (size_t)2 << 30;    // 2 GiB: far outside the Infinity Cache
    uint4* d;
    CHECK(hipMalloc(&d, bytes));
    {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 5; ++w) CHECK(hipMemsetAsync(d, w, bytes, 0));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) CHECK(hipMemsetAsync(d, r, bytes, 0));
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemsetAsync %4zu MB: %8.1f us  %6.2f TB/s\n", bytes >> 20, ms * 1e3 / 10, bytes / (ms * 1e-3 / 10) * 1e-12);
    }
    for (unsigned steps : {1u, 4u, 32u}) {
        if (run<0, false>(d, bytes, steps)) return 1;
        if (run<0, true>(d, bytes, steps)) return 1;
        if (run<1, false>(d, bytes, steps)) return 1;
        if (run<1, true>(d, bytes, steps)) return 1;
    }
    return 0;
}
