// Host <-> device copy rates as the ABI's host-buffer entry points see them: pageable destination (fresh / touched),
// pinned, and the staged variant (DMA into pinned double buffers + worker-thread memcpy into the pageable destination).
// hipcc --offload-arch=gfx950 -O3 -pthread -o _build/pcie_rates pcie_rates.hip
#include <hip/hip_runtime.h>
#include <algorithm>
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
__global__ void k_touch(unsigned long long* p) { if (threadIdx.x == 0) p[0] += 1; }

// The round trip of one blocking ABI call around its kernels: a 32-byte slot copied in, a kernel, the slot copied out, one
// stream wait -- with the host side of the slot on the stack (pageable: what libacx does) and in page-locked memory.
static int slot_round_trips() {
    unsigned long long* d; CK(hipMalloc(&d, 32)); CK(hipMemset(d, 0, 32));
    unsigned long long* pin; CK(hipHostMalloc(&pin, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int reps = 3000;
    for (int mode = 0; mode < 4; ++mode) {
        unsigned long long stack_in[4] = {0, ~0ull, 0, 0}, stack_out[4];
        unsigned long long* in = (mode & 1) ? pin : stack_in;
        unsigned long long* out = (mode & 2) ? pin + 4 : stack_out;
        in[0] = 0; in[1] = ~0ull; in[2] = 0; in[3] = 0;
        double best = 1e9;
        for (int round = 0; round < 3; ++round) {
            const double t0 = now();
            for (int i = 0; i < reps; ++i) {
                CK(hipMemcpyAsync(d, in, 32, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d);
                CK(hipMemcpyAsync(out, d, 32, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
            }
            best = std::min(best, (now() - t0) / reps);
        }
        printf("slot round trip (32 B in, kernel, 32 B out, wait): in %-8s out %-8s %6.1f us\n", (mode & 1) ? "pinned" : "pageable",
               (mode & 2) ? "pinned" : "pageable", best * 1e6);
    }
    // a small result download into the caller's pageable buffer: direct, against a copy into page-locked staging + memcpy
    char* dbuf; CK(hipMalloc(&dbuf, 4 << 20)); CK(hipMemset(dbuf, 3, 4 << 20));
    char* stage; CK(hipHostMalloc(&stage, 4 << 20));
    std::vector<char> user(4 << 20, 1);
    for (size_t bytes : {(size_t)32 << 10, (size_t)256 << 10, (size_t)2 << 20}) {
        double best[2] = {1e9, 1e9};
        for (int round = 0; round < 3; ++round)
            for (int staged = 0; staged < 2; ++staged) {
                const int n = 500;
                const double t0 = now();
                for (int i = 0; i < n; ++i) {
                    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d);
                    CK(hipMemcpyAsync(staged ? stage : user.data(), dbuf, bytes, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    if (staged) memcpy(user.data(), stage, bytes);
                }
                best[staged] = std::min(best[staged], (now() - t0) / n);
            }
        printf("download of %4zu KB after a kernel, pageable destination: direct %6.1f us, staged through page-locked memory %6.1f us\n",
               bytes >> 10, best[0] * 1e6, best[1] * 1e6);
    }
    return 0;
}

int main() {
    if (slot_round_trips()) return 1;
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
