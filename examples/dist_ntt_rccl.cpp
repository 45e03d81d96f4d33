// examples/dist_ntt_rccl.cpp -- the multi-GPU path of libacx from a C/C++ host, no Python: one process per GPU,
// RCCL (rccl.h) for the two collectives the path has.  This is what a Haskell host's C shim does as well
// (INTEGRATION.md section 4).
//
//   * distributed four-step NTT (replaces galois-fft at /root/reference/src/QAP.hs:521-524 for a transform that
//     spans GPUs, BASELINE.json configs[3]): acx_ntt_dist_step_dev (step 0) -> ncclAllToAll -> (step 1)
//   * sharded verifyAssignment (src/QAP.hs:276-282): acx_r1cs_verify_dev on this rank's rows -> ONE ncclAllReduce
//     of the violated-row count
//
// build:  hipcc -std=c++17 -I include examples/dist_ntt_rccl.cpp -L arithmetic-circuits_amd -lacx -lrccl -o dist_ntt_rccl
// run:    one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK set (mpirun, torchrun or a shell loop);
//         ACX_NCCL_ID_FILE names a file on a shared path through which rank 0 publishes the ncclUniqueId.
//         WORLD_SIZE=1 runs the same code with a one-rank communicator (the 1-GPU CI box does this).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "acx.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define ACXCHECK(x) do { int r_ = (x); if (r_ != ACX_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, acx_strerror(r_), acx_last_error()); return r_ == ACX_ERR_NO_DEVICE ? 77 : 1; } } while (0)

static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }

int main() {
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", 0);
    const uint32_t log_n = (uint32_t)env_int("ACX_LOG_N", 20), log_r = log_n / 2;
    HIPCHECK(hipSetDevice(local));
    acx_ctx* ctx = nullptr;
    ACXCHECK(acx_ctx_create(ACX_FIELD_BN254_FR, local, &ctx));
    hipStream_t stream = (hipStream_t)acx_ctx_stream(ctx);       // collectives go on libacx's own stream: no fences needed

    // ---- communicator: rank 0 publishes the unique id through a file
    ncclUniqueId id;
    const char* id_file = getenv("ACX_NCCL_ID_FILE");
    if (world > 1 && !id_file) { fprintf(stderr, "set ACX_NCCL_ID_FILE for WORLD_SIZE > 1\n"); return 1; }
    if (rank == 0) {
        NCCLCHECK(ncclGetUniqueId(&id));
        if (id_file) { FILE* f = fopen(id_file, "wb"); if (!f) return 1; fwrite(&id, sizeof id, 1, f); fclose(f); }
    } else {
        for (int tries = 0;; ++tries) {
            FILE* f = fopen(id_file, "rb");
            if (f && fread(&id, sizeof id, 1, f) == 1) { fclose(f); break; }
            if (f) fclose(f);
            if (tries > 600) { fprintf(stderr, "no unique id\n"); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t comm;
    NCCLCHECK(ncclCommInitRank(&comm, world, id, rank));

    // ---- this rank's COLS block of x[i] = i + 1 (include/acx.h: COLS [i2l][i1] = x[i1*C + rank*C/W + i2l])
    const uint64_t N = 1ull << log_n, R = 1ull << log_r, C = N / R, cw = C / world, local_n = N / world;
    std::vector<acx_fr> host(local_n);
    memset(host.data(), 0, local_n * sizeof(acx_fr));
    for (uint64_t i2l = 0; i2l < cw; ++i2l)
        for (uint64_t i1 = 0; i1 < R; ++i1) {
            const uint64_t v = i1 * C + rank * cw + i2l + 1;
            memcpy(host[i2l * R + i1].b, &v, 8);
        }
    void *x, *xchg_s, *xchg_r, *y, *back;
    for (void** p : {&x, &xchg_s, &xchg_r, &y, &back}) HIPCHECK(hipMalloc(p, local_n * 32));
    HIPCHECK(hipMemcpyAsync(x, host.data(), local_n * 32, hipMemcpyHostToDevice, stream));
    ACXCHECK(acx_dev_from_canonical(ctx, local_n, x, x, nullptr));

    auto exchange = [&](void* send, void* recv) -> int {
        if (world == 1) { HIPCHECK(hipMemcpyAsync(recv, send, local_n * 32, hipMemcpyDeviceToDevice, stream)); return 0; }
        NCCLCHECK(ncclAllToAll(send, recv, local_n * 32 / world, ncclUint8, comm, stream));   // (R/W)*(C/W) elements per peer
        return 0;
    };
    const auto t0 = std::chrono::steady_clock::now();
    // forward: COLS -> XCHG, all-to-all, XCHG -> ROWS
    ACXCHECK(acx_ntt_dist_step_dev(ctx, log_n, log_r, world, rank, 0, 0, nullptr, x, xchg_s));
    if (exchange(xchg_s, xchg_r)) return 1;
    ACXCHECK(acx_ntt_dist_step_dev(ctx, log_n, log_r, world, rank, 0, 1, nullptr, xchg_r, y));
    // inverse: ROWS -> XCHG, all-to-all, XCHG -> COLS
    ACXCHECK(acx_ntt_dist_step_dev(ctx, log_n, log_r, world, rank, 1, 0, nullptr, y, xchg_s));
    if (exchange(xchg_s, xchg_r)) return 1;
    ACXCHECK(acx_ntt_dist_step_dev(ctx, log_n, log_r, world, rank, 1, 1, nullptr, xchg_r, back));
    ACXCHECK(acx_dev_to_canonical(ctx, local_n, back, back));
    std::vector<acx_fr> got(local_n);
    HIPCHECK(hipMemcpyAsync(got.data(), back, local_n * 32, hipMemcpyDeviceToHost, stream));
    ACXCHECK(acx_ctx_sync(ctx));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long mism = memcmp(got.data(), host.data(), local_n * 32) != 0;

    // X[0] = sum of all inputs = N(N+1)/2: rank 0 owns k1 = 0, k2 = 0 at ROWS position 0
    acx_fr x0;
    void* tmp;
    HIPCHECK(hipMalloc(&tmp, 32));
    ACXCHECK(acx_dev_to_canonical(ctx, 1, y, tmp));
    HIPCHECK(hipMemcpyAsync(&x0, tmp, 32, hipMemcpyDeviceToHost, stream));
    ACXCHECK(acx_ctx_sync(ctx));
    if (rank == 0) {
        unsigned __int128 s = (unsigned __int128)N * (N + 1) / 2;
        acx_fr want;
        memset(&want, 0, sizeof want);
        memcpy(want.b, &s, 16);
        mism += memcmp(&x0, &want, 32) != 0;
    }

    // ---- the verdict collective: every rank contributes its violated-row count (here: its mismatch count)
    unsigned long long* d_cnt;
    HIPCHECK(hipMalloc((void**)&d_cnt, 8));
    HIPCHECK(hipMemcpyAsync(d_cnt, &mism, 8, hipMemcpyHostToDevice, stream));
    NCCLCHECK(ncclAllReduce(d_cnt, d_cnt, 1, ncclUint64, ncclSum, comm, stream));
    unsigned long long total = 0;
    HIPCHECK(hipMemcpyAsync(&total, d_cnt, 8, hipMemcpyDeviceToHost, stream));
    ACXCHECK(acx_ctx_sync(ctx));
    if (rank == 0)
        printf("distributed NTT 2^%u over %d rank(s): forward + inverse round trip %s, X[0] checked, %.2f ms incl. first-use table setup\n",
               log_n, world, total == 0 ? "exact" : "MISMATCH", ms);
    ncclCommDestroy(comm);
    acx_ctx_destroy(ctx);
    return total == 0 ? 0 : 1;
}
