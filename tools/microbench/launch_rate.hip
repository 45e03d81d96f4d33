// Back-to-back launches of a near-empty kernel on one stream: the floor under "one launch per dependency level"
// (acx_r1cs_eval).  hipcc --offload-arch=gfx950 -O3 -o _build/launch_rate launch_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_touch(unsigned* p, unsigned n) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = p[(t * 7919u) % n] + 1u;      // one dependent load + store, as a level has at least
}
int main() {
    unsigned* d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int blocks : {1, 25, 256}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, d, 6400u);
            const auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            const auto t2 = std::chrono::steady_clock::now();
            printf("blocks %3d: enqueue %.2f us per launch, complete %.2f us per launch\n", blocks,
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / 2000);
        }
    }
    return 0;
}
