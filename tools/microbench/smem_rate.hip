// How much can the SCALAR data path (s_load -> scalar cache -> L2 -> HBM) carry, alone and beside a vector stream?  The question
// behind it: could K2 pull part of its constraint stream into the L2 through scalar loads, which hold no entry of the vector
// L1's miss queue (profiles/r03_r1cs.txt item 9)?  hipcc --offload-arch=gfx950 -O3 -o _build/smem_rate smem_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef u32 v16u32 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) const v16u32 c_v16;
typedef u32 v4u32 __attribute__((ext_vector_type(4)));

// every wave walks its own contiguous region with s_load_dwordx16 (64 B), DEPTH loads in flight
template <int DEPTH>
__global__ void k_scalar(const u32* __restrict__ buf, unsigned long long bytes_per_wave, u32* out) {
    const unsigned long long wave = blockIdx.x;                     // 64-thread workgroups: one wave each
    const char* p = (const char*)buf + wave * bytes_per_wave;
    u32 acc = 0;
    for (unsigned long long o = 0; o < bytes_per_wave; o += 128ull * DEPTH) {
        v16u32 r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) r[d] = *(c_v16*)(unsigned long long)(p + o + 128ull * d);   // one 64-byte load per 128-byte line
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += r[d].x ^ r[d].w;
    }
    if (threadIdx.x == 0 && acc == 0x12345678u) out[0] = acc;
}

// the vector stream: 16 bytes per lane, nontemporal, one wave per 64 threads
__global__ void k_vector(const v4u32* __restrict__ buf, unsigned long long vecs_per_wave, u32* out) {
    const unsigned long long wave = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const v4u32* p = buf + wave * vecs_per_wave;
    u32 acc = 0;
    for (unsigned long long o = lane; o < vecs_per_wave; o += 256) {
        const v4u32 a = __builtin_nontemporal_load(p + o), b = __builtin_nontemporal_load(p + o + 64),
                    c = __builtin_nontemporal_load(p + o + 128), d = __builtin_nontemporal_load(p + o + 192);
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const unsigned long long total = 1ull << 30;                     // 1 GiB region each
    u32 *a, *b, *out;
    hipMalloc(&a, total); hipMalloc(&b, total); hipMalloc(&out, 64);
    hipMemset(a, 1, total); hipMemset(b, 2, total);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1, f0, f1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    auto run_scalar = [&](int depth, unsigned waves, hipStream_t s) {
        const unsigned long long per = total / waves;
        if (depth == 1) hipLaunchKernelGGL(k_scalar<1>, dim3(waves), dim3(64), 0, s, a, per, out);
        else if (depth == 2) hipLaunchKernelGGL(k_scalar<2>, dim3(waves), dim3(64), 0, s, a, per, out);
        else hipLaunchKernelGGL(k_scalar<4>, dim3(waves), dim3(64), 0, s, a, per, out);
    };
    auto run_vector = [&](hipStream_t s) {
        const unsigned waves = 256 * 24;
        hipLaunchKernelGGL(k_vector, dim3(waves / 4), dim3(256), 0, s, (const v4u32*)b, total / 16 / waves, out);
    };
    for (int depth : {1, 2, 4})
        for (unsigned waves : {256u * 4, 256u * 8, 256u * 16, 256u * 32}) {
            run_scalar(depth, waves, s1);
            hipStreamSynchronize(s1);
            hipEventRecord(e0, s1); run_scalar(depth, waves, s1); hipEventRecord(e1, s1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("scalar alone  depth %d waves %5u: %.3f ms  %.2f TB/s of 128-byte lines touched (%.2f TB/s of 64-byte loads)\n", depth, waves, ms,
                   total / ms * 1e-9, total / 2 / ms * 1e-9);
        }
    run_vector(s2); hipStreamSynchronize(s2);
    hipEventRecord(f0, s2); run_vector(s2); hipEventRecord(f1, s2); hipEventSynchronize(f1);
    float msv; hipEventElapsedTime(&msv, f0, f1);
    printf("vector alone: %.3f ms  %.2f TB/s\n", msv, total / msv * 1e-9);
    // both at once on two streams: does the scalar traffic come on top of the vector stream's rate?
    for (unsigned waves : {256u * 4, 256u * 8}) {
        hipDeviceSynchronize();
        hipEventRecord(e0, s1); hipEventRecord(f0, s2);
        run_vector(s2); run_scalar(4, waves, s1);
        hipEventRecord(e1, s1); hipEventRecord(f1, s2);
        hipEventSynchronize(e1); hipEventSynchronize(f1);
        float ms1, ms2; hipEventElapsedTime(&ms1, e0, e1); hipEventElapsedTime(&ms2, f0, f1);
        printf("together (scalar depth 4, %u waves): scalar %.3f ms (%.2f TB/s of lines), vector %.3f ms (%.2f TB/s)\n", waves, ms1, total / ms1 * 1e-9, ms2,
               total / ms2 * 1e-9);
    }
    return 0;
}
