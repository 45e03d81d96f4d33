// tools/microbench/valu_rates.hip -- issue-rate probe for the integer / fp64 VALU ops a
// 256-bit Montgomery multiply can be built from on gfx950.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
// Prints, per instruction, wave-instructions/s for the whole chip and the implied cycles
// per wave-instruction per SIMD at the measured clock of a v_fma_f32 calibration (2 cyc).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 32768;
constexpr int UNROLL = 8;   // independent chains per thread

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed) {
    uint32_t a[UNROLL], b[UNROLL];
    uint64_t acc[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        a[i] = seed * (threadIdx.x + 1) + i; b[i] = seed ^ (0x9e3779b9u * (i + 1));
        acc[i] = ((uint64_t)a[i] << 32) | b[i]; d[i] = 1.0 + 1e-9 * (double)a[i];
    }
    double dm = 1.0000001, da = 1e-12;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(seed));
            if constexpr (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if constexpr (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if constexpr (OP == 5) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            if constexpr (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
            if constexpr (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
            if constexpr (OP == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(seed));
            if constexpr (OP == 11) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 14) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(*(uint64_t*)&d[i]) : "v"(a[i]), "v"(b[i]));
            if constexpr (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 16) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
            if constexpr (OP == 17) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
            if constexpr (OP == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
            if constexpr (OP == 19) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
            if constexpr (OP == 20) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            if constexpr (OP == 22) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(a[i]) : "v"(b[i]), "v"(seed) : "vcc");
            if constexpr (OP == 23) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_fma_f64 %1, %1, %4, %5" : "+v"(acc[i]), "+v"(d[i]) : "v"(b[i]), "v"(seed), "v"(dm), "v"(da) : "vcc");
            if constexpr (OP == 21) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i]));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ (uint32_t)d[i];
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int OP> double run(const char* name, double calib_cycles_per_inst, double* clock_out) {
    uint32_t* out; CHECK(hipMalloc(&out, 4096));
    int blocks = 256 * 8;   // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<OP><<<blocks, 256>>>(out, 3); CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0)); probe<OP><<<blocks, 256>>>(out, 3 + r); CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1)); float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double t = ms[2] * 1e-3;
    double wave_insts = (double)blocks * 4 * ITERS * UNROLL * ((OP == 22 || OP == 23) ? 2 : 1);       // wave-level instructions
    double per_simd_per_s = wave_insts / 1024.0 / t;                // per SIMD
    double cyc = 0;
    if (clock_out && calib_cycles_per_inst > 0) { *clock_out = per_simd_per_s * calib_cycles_per_inst; }
    if (clock_out) cyc = *clock_out / per_simd_per_s;
    printf("%-22s %8.3f ms  %.3e wave-inst/s/SIMD  -> %.2f cycles/wave-inst (clock %.2f GHz)\n", name, ms[2], per_simd_per_s, cyc, clock_out ? *clock_out * 1e-9 : 0.0);
    CHECK(hipFree(out));
    return cyc;
}

int main() {
    double clk = 0;
    run<0>("v_fma_f32 (calib=2)", 2.0, &clk);
    run<12>("v_mov_b32", 0, &clk);
    run<13>("v_add_u32", 0, &clk);
    run<4>("v_add_co_u32", 0, &clk);
    run<5>("v_addc_co_u32", 0, &clk);
    run<10>("v_add3_u32", 0, &clk);
    run<19>("v_cndmask_b32", 0, &clk);
    run<15>("v_alignbit_b32", 0, &clk);
    run<9>("v_lshl_add_u64", 0, &clk);
    run<21>("v_lshrrev_b64", 0, &clk);
    run<1>("v_mad_u64_u32 (vcc)", 0, &clk);
    run<14>("v_mad_u64_u32 (sgpr)", 0, &clk);
    run<20>("v_mad_i64_i32", 0, &clk);
    run<2>("v_mul_lo_u32", 0, &clk);
    run<3>("v_mul_hi_u32", 0, &clk);
    run<7>("v_mul_u32_u24", 0, &clk);
    run<11>("v_mul_hi_u32_u24", 0, &clk);
    run<8>("v_mad_u32_u24", 0, &clk);
    run<22>("mad_u64+addc (2 inst)", 0, &clk);
    run<23>("mad_u64+fma_f64 (2 inst)", 0, &clk);
    run<6>("v_fma_f64", 0, &clk);
    run<16>("v_mul_f64", 0, &clk);
    run<17>("v_add_f64", 0, &clk);
    run<18>("v_pk_fma_f32", 0, &clk);
    return 0;
}
