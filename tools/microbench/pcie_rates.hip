// Host <-> device copy rates as the ABI's host-buffer entry points see them: pageable destination (fresh / touched),
// pinned, and the staged variant (DMA into pinned double buffers + worker-thread memcpy into the pageable destination).
// hipcc --offload-arch=gfx950 -O3 -pthread -o _build/pcie_rates pcie_rates.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char* dst, const char* src, size_t n, unsigned T) {
    if (T <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([=] { size_t a = n * t / T, b = n * (t + 1) / T; memcpy(dst + a, src + a, b - a); });
    for (auto& x : th) x.join();
}
int main() {
    const size_t B = 512ull << 20, CH = 16ull << 20;
    char* d; CK(hipMalloc(&d, B)); CK(hipMemset(d, 1, B));
    char* pin; CK(hipHostMalloc(&pin, B));
    char* stage[2]; CK(hipHostMalloc(&stage[0], CH)); CK(hipHostMalloc(&stage[1], CH));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
    for (int rep = 0; rep < 2; ++rep) {
        char* fresh = (char*)malloc(B);
        double t0 = now(); CK(hipMemcpy(fresh, d, B, hipMemcpyDeviceToHost)); double t1 = now();
        printf("D2H pageable fresh   %6.1f GB/s\n", B / (t1 - t0) * 1e-9);
        t0 = now(); CK(hipMemcpy(fresh, d, B, hipMemcpyDeviceToHost)); t1 = now();
        printf("D2H pageable touched %6.1f GB/s\n", B / (t1 - t0) * 1e-9);
        t0 = now(); CK(hipMemcpy(d, fresh, B, hipMemcpyHostToDevice)); t1 = now();
        printf("H2D pageable         %6.1f GB/s\n", B / (t1 - t0) * 1e-9);
        t0 = now(); CK(hipMemcpy(pin, d, B, hipMemcpyDeviceToHost)); t1 = now();
        printf("D2H pinned           %6.1f GB/s\n", B / (t1 - t0) * 1e-9);
        t0 = now(); CK(hipMemcpy(d, pin, B, hipMemcpyHostToDevice)); t1 = now();
        printf("H2D pinned           %6.1f GB/s\n", B / (t1 - t0) * 1e-9);
        free(fresh);
        for (unsigned T : {1u, 4u, 8u, 16u}) {
            char* dst = (char*)malloc(B);            // fresh every time: page faults included, as for a caller's new buffer
            t0 = now();
            const size_t nch = B / CH;
            for (size_t k = 0; k <= nch; ++k) {
                if (k < nch) { CK(hipMemcpyAsync(stage[k & 1], d + k * CH, CH, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[k & 1], s)); }
                if (k > 0) { CK(hipEventSynchronize(ev[(k - 1) & 1])); par_copy(dst + (k - 1) * CH, stage[(k - 1) & 1], CH, T); }
            }
            t1 = now();
            printf("D2H staged, %2u threads, fresh destination   %6.1f GB/s\n", T, B / (t1 - t0) * 1e-9);
            t0 = now();
            for (size_t k = 0; k <= nch; ++k) {
                if (k < nch) { if (k >= 2) CK(hipEventSynchronize(ev[k & 1])); par_copy(stage[k & 1], dst + k * CH, CH, T);
                               CK(hipMemcpyAsync(d + k * CH, stage[k & 1], CH, hipMemcpyHostToDevice, s)); CK(hipEventRecord(ev[k & 1], s)); }
            }
            CK(hipStreamSynchronize(s));
            t1 = now();
            printf("H2D staged, %2u threads                      %6.1f GB/s\n", T, B / (t1 - t0) * 1e-9);
            free(dst);
        }
    }
    return 0;
}
