// tools/microbench/fe_rates.hip -- cycles per wave-level field operation on gfx950 (cost model for
// kernels.hip.h).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../arithmetic-circuits_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "fr.hip.h"
using namespace acx;
using F = Bn254Fr;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void probe(const uint4* in, uint4* out) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    Fe a = fe_load(in + 2 * (tid & 1023)), b = fe_load(in + 2 * ((tid + 7) & 1023));
    u32 raw[8];
    fe_pack(a, raw);
    Wide w; wide_zero(w);
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (OP == 0) a = fe_mul<F>(a, b);
        if constexpr (OP == 1) a = fe_add<F>(a, b);
        if constexpr (OP == 2) a = fe_sub<F>(a, b);
        if constexpr (OP == 3) { wide_mac(w, a, b); a.l[0] ^= (u32)w.c[3] & 1; }
        if constexpr (OP == 4) { wide_mac(w, a, b); a = wide_reduce<F>(w); wide_zero(w); }
        if constexpr (OP == 5) { raw[it & 7] += a.l[0]; a = fe_unpack(raw); }
        if constexpr (OP == 6) { fe_pack(a, raw); raw[0] ^= it; a = fe_unpack(raw); }
        if constexpr (OP == 7) a = fe_cond_sub<F::P2>(a), a.l[0] += b.l[0] & 1;
        if constexpr (OP == 8) { a.l[1] += fe_is_zero<F>(a) ? 1 : 0; a.l[0] += 3; }
        if constexpr (OP == 9) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a.l[0]) : "v"(b.l[0]), "s"(0x5555555555555555ull)); }
        if constexpr (OP == 10) { asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a.l[0]) : "v"(b.l[0]), "v"(b.l[1])); }
        if constexpr (OP == 11) { asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a.l[0]) : "v"(b.l[0]), "v"(b.l[1]) : "vcc"); }
    }
    for (int k = 0; k < 17; ++k) a.l[k % 9] ^= (u32)w.c[k];
    fe_carry(a);
    if (a.l[0] == 0x12345) fe_store(out + 2 * tid, a);
}

template <int OP> void run(const char* name, int per_iter, const uint4* in, uint4* out) {
    int blocks = 256 * 4;   // 4 waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<OP><<<blocks, 256>>>(in, out); CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0)); probe<OP><<<blocks, 256>>>(in, out); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double t = ms[2] * 1e-3;
    double wave_ops = (double)blocks * 4 * ITERS * per_iter;
    double per_simd = wave_ops / 1024.0 / t;
    printf("%-28s %8.3f ms  %.3e wave-ops/s/SIMD -> %.1f cycles/op @2.0GHz, chip %.3e ops/s\n", name, ms[2], per_simd, 2.0e9 / per_simd, per_simd * 1024 * 64);
}

int main() {
    uint4 *in, *out; CHECK(hipMalloc(&in, 1024 * 32)); CHECK(hipMalloc(&out, 256 * 4 * 256 * 32));
    std::vector<uint32_t> h(1024 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) >> ((i % 8 == 7) ? 3 : 0);
    CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0>("fe_mul", 1, in, out);
    run<1>("fe_add", 1, in, out);
    run<2>("fe_sub", 1, in, out);
    run<3>("wide_mac", 1, in, out);
    run<4>("wide_mac+reduce+zero", 1, in, out);
    run<5>("fe_unpack", 1, in, out);
    run<6>("fe_pack+unpack", 1, in, out);
    run<7>("fe_cond_sub", 1, in, out);
    run<8>("fe_is_zero", 1, in, out);
    run<9>("v_cndmask_e64 (sgpr mask)", 1, in, out);
    run<10>("v_bfi_b32", 1, in, out);
    run<11>("v_cmp+v_cndmask vcc", 2, in, out);
    return 0;
}
