// tools/microbench/gather_rates.hip -- cost of divergent 32-byte gathers from an L2-resident table on gfx950:
//   A: every lane loads its own element as two dwordx4 (lo, hi)          -> 2 instr, 64 distinct lines each
//   B: lane pairs share an element: even lane loads lo, odd lane hi; two instr cover 64 elements
//      (each instruction touches 32 distinct elements, the pair's 32 bytes sit in one 64-byte chunk)
//   C: coalesced reference: lane i loads element base+i
// hipcc --offload-arch=gfx950 -O3 -o gather_rates gather_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4 gv4;
constexpr int ITERS = 256;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t mask, uint4* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    v4 acc = {0, 0, 0, 0};
    uint32_t c = idx[tid & 0xfffff];
    // every workgroup gathers from its OWN window (as the rows of different CUs do), not from one hot region
    table += 2 * (uint64_t)((blockIdx.x % 64) * (mask + 1));
#pragma unroll 8
    for (int it = 0; it < ITERS; ++it) {
        c = (c * 1664525u + 1013904223u) & mask;               // independent of the loaded data: throughput, not latency
        if (MODE == 0) {
            const v4 lo = *(gv4*)(table + 2 * (uint64_t)c), hi = *(gv4*)(table + 2 * (uint64_t)c + 1);
            acc += lo ^ hi;
        } else if (MODE == 1) {
            const uint32_t cp = __shfl_xor(c, 1);
            const uint32_t ce = (lane & 1) ? cp : c, co = (lane & 1) ? c : cp, h = lane & 1;
            const v4 i1 = *(gv4*)(table + 2 * (uint64_t)ce + h), i2 = *(gv4*)(table + 2 * (uint64_t)co + h);
            const v4 send = (lane & 1) ? i1 : i2;
            v4 recv;
            recv.x = __shfl_xor(send.x, 1); recv.y = __shfl_xor(send.y, 1); recv.z = __shfl_xor(send.z, 1); recv.w = __shfl_xor(send.w, 1);
            const v4 lo = (lane & 1) ? recv : i1, hi = (lane & 1) ? i2 : recv;
            acc += lo ^ hi;
        } else {
            const uint32_t base = (c & ~63u & mask);
            const v4 lo = *(gv4*)(table + 2 * (uint64_t)(base + lane)), hi = *(gv4*)(table + 2 * (uint64_t)(base + lane) + 1);
            acc += lo ^ hi;
        }
    }
    if (acc.x == 0x12345678u) out[tid] = make_uint4(acc.x, acc.y, acc.z, acc.w);
}

template <int MODE> void run(const char* name, const uint4* table, const uint32_t* idx, uint32_t mask, uint4* out) {
    const int blocks = 256 * 4;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MODE><<<blocks, 256>>>(table, idx, mask, out); CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0)); probe<MODE><<<blocks, 256>>>(table, idx, mask, out); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double gathers = (double)blocks * 256 * ITERS;
    printf("%-34s window %6u KB: %8.3f ms  %.3e element-gathers/s  (%.1f cycles per wave-gather per CU @2.1GHz)\n", name,
           (mask + 1) * 32 / 1024, ms[2], gathers / (ms[2] * 1e-3), 2.1e9 / (gathers / 64 / 256 / (ms[2] * 1e-3)));
}

int main() {
    const uint32_t N = 1u << 22;   // 128 MB table
    uint4* table; uint32_t* idx; uint4* out;
    CHECK(hipMalloc(&table, (size_t)N * 32)); CHECK(hipMalloc(&idx, (1 << 20) * 4)); CHECK(hipMalloc(&out, 256 * 4 * 256 * 16));
    CHECK(hipMemset(table, 1, (size_t)N * 32));
    std::vector<uint32_t> h(1 << 20); for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
    CHECK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (uint32_t mask : {(1u << 12) - 1, (1u << 13) - 1, (1u << 16) - 1}) {
        run<0>("A own element, lo+hi", table, idx, mask, out);
        run<1>("B paired lanes share an element", table, idx, mask, out);
        run<2>("C coalesced", table, idx, mask, out);
    }
    return 0;
}
