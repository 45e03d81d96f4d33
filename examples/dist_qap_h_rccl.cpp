// examples/dist_qap_h_rccl.cpp -- `verificationWitness` (h(x) = (L*R - O) / (x^N - 1), /root/reference/src/QAP.hs:292-327)
// over several GPUs from a C/C++ host, no Python: one process per GPU, the local work through include/acx.h, the
// exchanges through RCCL on libacx's own stream.  BASELINE.json configs[3] in small: the recipe of INTEGRATION.md section 4.
//
//   rows          every rank loads ONLY its rows, block-cyclic and in ascending order: local row [k2][kl] = global row
//                 (rank*R/W + kl) + k2*R, so the residual kernel's <A_i,w>, <B_i,w>, <C_i,w> ARE three evaluation vectors in the
//                 transposed ROWS layout, which the first inverse steps read through their strides (ACX_DIST_ROWS_T);
//                 acx_r1cs_dots_h_dev stores them as <A_i,w> / z, <B_i,w>, -<C_i,w> / z (z = g^N - 1 on the coset)
//   3 inverse     acx_ntt_dist_step_dev(step 0) -> ncclAllToAll -> (step 1): coefficients of L / z, R (each times g^i: their
//                 inverse transforms are coset transforms with shift 1/g) and -O / z in COLS ownership
//   2 coset       L / z and R only: O(x) enters the quotient in coefficient form
//   1 inverse coset through acx_ntt_dist_step_fused_dev: its first step transforms the PRODUCT (L / z) * R as it loads the
//                 points, its second step adds -O / z behind the closing multiplication: h in COLS ownership (rank g holds
//                 h[i1*C + g*C/W + i2l]) with no elementwise pass outside the transforms
//   verdict       ONE ncclAllReduce of the violated-row counts
// Six all-to-alls per h(x).  Every rank also computes h(x) of the whole system on its own GPU (acx_qap_h) and compares
// its block: the distributed four-step transforms against the single-GPU pass kernels.
//
// build:  hipcc -std=c++17 -I include examples/dist_qap_h_rccl.cpp -L arithmetic-circuits_amd -lacx -lrccl -o dist_qap_h_rccl
// run:    like dist_ntt_rccl (RANK / WORLD_SIZE / LOCAL_RANK, ACX_NCCL_ID_FILE for WORLD_SIZE > 1); ACX_LOG_N (default 14).
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
static acx_fr fr_u64(uint64_t v) { acx_fr x; memset(&x, 0, sizeof x); memcpy(x.b, &v, 8); return x; }

// The constraint system: n = N rows, row i checks w[a_i] * (w[b_i] + 3) = w[out_i] with a_i, b_i among K small inputs and
// one output wire per row (m = 1 + K + n).  Rows of a subset in the given order, as three CSR matrices.
struct Rows {
    std::vector<uint32_t> pa, ca, pb, cb, pc, cc;
    std::vector<acx_fr> va, vb, vc;
};
static const uint64_t K = 64;
static uint64_t a_of(uint64_t i) { return 1 + (i * 7 + 3) % K; }
static uint64_t b_of(uint64_t i) { return 1 + (i * 13 + 5) % K; }
static Rows rows_of(const std::vector<uint64_t>& which) {
    Rows r;
    r.pa.push_back(0); r.pb.push_back(0); r.pc.push_back(0);
    for (uint64_t i : which) {
        r.ca.push_back((uint32_t)a_of(i)); r.va.push_back(fr_u64(1));
        r.cb.push_back(0); r.vb.push_back(fr_u64(3));                       // the constant wire
        r.cb.push_back((uint32_t)b_of(i)); r.vb.push_back(fr_u64(1));
        r.cc.push_back((uint32_t)(1 + K + i)); r.vc.push_back(fr_u64(1));
        r.pa.push_back((uint32_t)r.ca.size()); r.pb.push_back((uint32_t)r.cb.size()); r.pc.push_back((uint32_t)r.cc.size());
    }
    return r;
}
static int load(acx_ctx* ctx, const Rows& r, uint64_t n, uint64_t m, acx_r1cs** out) {
    const acx_csr A{r.pa.data(), r.ca.data(), r.va.data()}, B{r.pb.data(), r.cb.data(), r.vb.data()}, C{r.pc.data(), r.cc.data(), r.vc.data()};
    return acx_r1cs_load(ctx, n, m, &A, &B, &C, out);
}

int main() {
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", 0);
    const uint32_t log_n = (uint32_t)env_int("ACX_LOG_N", 14), log_r = log_n / 2;
    HIPCHECK(hipSetDevice(local));
    acx_ctx* ctx = nullptr;
    ACXCHECK(acx_ctx_create(ACX_FIELD_BN254_FR, local, &ctx));
    hipStream_t stream = (hipStream_t)acx_ctx_stream(ctx);

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

    const uint64_t N = 1ull << log_n, R = 1ull << log_r, C = N / R, rw = R / world, cw = C / world, L = N / world, m = 1 + K + N;
    // ---- witness (replicated): inputs 2 .. K+1, every output wire = w[a] * (w[b] + 3); one copy corrupted for the negative case
    std::vector<acx_fr> w(m);
    std::vector<uint64_t> wi(m);
    wi[0] = 1;
    for (uint64_t k = 1; k <= K; ++k) wi[k] = k + 1;
    for (uint64_t i = 0; i < N; ++i) wi[1 + K + i] = wi[a_of(i)] * (wi[b_of(i)] + 3);
    for (uint64_t k = 0; k < m; ++k) w[k] = fr_u64(wi[k]);

    // ---- this rank's rows in ASCENDING order (local row [k2][kl]: runs of R/W consecutive rows -- the gathers of the rows in
    // flight stay in a narrow window of the witness), and (for the check) the whole system in natural order
    std::vector<uint64_t> mine(L), all(N);
    for (uint64_t k2 = 0; k2 < C; ++k2) for (uint64_t kl = 0; kl < rw; ++kl) mine[k2 * rw + kl] = (rank * rw + kl) + k2 * R;
    for (uint64_t i = 0; i < N; ++i) all[i] = i;
    acx_r1cs *r_local = nullptr, *r_full = nullptr;
    ACXCHECK(load(ctx, rows_of(mine), L, m, &r_local));
    ACXCHECK(load(ctx, rows_of(all), N, m, &r_full));

    void *d_w, *dots, *coef, *send, *recv, *h;
    HIPCHECK(hipMalloc(&d_w, m * 32));
    for (void** p : {&dots, &coef}) HIPCHECK(hipMalloc(p, 3 * L * 32));
    for (void** p : {&send, &recv, &h}) HIPCHECK(hipMalloc(p, L * 32));
    uint64_t* d_res;
    HIPCHECK(hipMalloc((void**)&d_res, 16));
    acx_fr g = fr_u64(5);                                                    // coset generator: 5^N != 1 in BN254 Fr
    // 1/5 mod r (little endian): an inverse coset transform with shift 1/g multiplies its result by g^i, which is how L and R
    // receive their coset factor -- on the closing multiplication of their inverse transform, not on the load of the forward one
    acx_fr ginv = {{0x67, 0x66, 0x66, 0xc6, 0xd4, 0xfb, 0xf3, 0xe7, 0x06, 0x2d, 0x4a, 0xca, 0xe9, 0x5c, 0xae, 0xa9, 0x8b, 0x56, 0xcd, 0x33, 0x7c, 0xb5, 0xb9, 0x49, 0xaa, 0xd9, 0x13, 0x5a, 0x94, 0x52, 0x5b, 0x13}};
    auto at = [&](void* base, uint64_t k) { return (void*)((char*)base + k * L * 32); };
    auto exchange = [&](void* s, void* r) -> int {
        if (world == 1) { HIPCHECK(hipMemcpyAsync(r, s, L * 32, hipMemcpyDeviceToDevice, stream)); return 0; }
        NCCLCHECK(ncclAllToAll(s, r, L * 32 / world, ncclUint8, comm, stream));
        return 0;
    };
    // rows_t: `in` is the transposed ROWS block [k2][kl] -- what the residual kernel writes for rows loaded in ascending order
    auto transform = [&](int inverse, const acx_fr* shift, void* in, void* out, uint32_t rows_t = 0) -> int {
        ACXCHECK(acx_ntt_dist_step_ex_dev(ctx, log_n, log_r, world, rank, inverse, 0, rows_t, shift, in, send));
        if (exchange(send, recv)) return 1;
        ACXCHECK(acx_ntt_dist_step_dev(ctx, log_n, log_r, world, rank, inverse, 1, shift, recv, out));
        return 0;
    };

    unsigned long long mism = 0;
    for (int pass = 0; pass < 2; ++pass) {                                   // 0: satisfying witness, 1: one wire corrupted
        std::vector<acx_fr> ww = w;
        if (pass == 1) ww[1 + K + N / 3].b[0] ^= 1;
        HIPCHECK(hipMemcpyAsync(d_w, ww.data(), m * 32, hipMemcpyHostToDevice, stream));
        ACXCHECK(acx_dev_from_canonical(ctx, m, d_w, d_w, nullptr));
        const uint64_t init[2] = {0, ~0ull};
        HIPCHECK(hipMemcpyAsync(d_res, init, 16, hipMemcpyHostToDevice, stream));
        ACXCHECK(acx_r1cs_dots_h_dev(r_local, d_w, 0, d_res, dots, log_n, &g));            // dots: transposed ROWS layout, three vectors, 1/z and -1/z riding on them
        for (uint64_t k = 0; k < 3; ++k) if (transform(1, k < 2 ? &ginv : nullptr, at(dots, k), at(coef, k), ACX_DIST_ROWS_T)) return 1;      // -> coefficients (COLS): g^i L_i, g^i R_i, O_i
        for (uint64_t k = 0; k < 2; ++k) if (transform(0, nullptr, at(coef, k), at(dots, k))) return 1;      // L, R on the coset (ROWS): plain transforms of the shifted coefficients
        // the last transform: (L / z) * R on the way in, -O / z on the way out -> h (COLS)
        ACXCHECK(acx_ntt_dist_step_fused_dev(ctx, log_n, log_r, world, rank, 1, 0, 0, &g, at(dots, 0), at(dots, 1), nullptr, send));
        if (exchange(send, recv)) return 1;
        ACXCHECK(acx_ntt_dist_step_fused_dev(ctx, log_n, log_r, world, rank, 1, 1, 0, &g, recv, nullptr, at(coef, 2), h));
        NCCLCHECK(ncclAllReduce(d_res, d_res, 1, ncclUint64, ncclSum, comm, stream));                         // the verdict
        uint64_t res[2];
        HIPCHECK(hipMemcpyAsync(res, d_res, 16, hipMemcpyDeviceToHost, stream));
        ACXCHECK(acx_dev_to_canonical(ctx, L, h, h));
        std::vector<acx_fr> got(L);
        HIPCHECK(hipMemcpyAsync(got.data(), h, L * 32, hipMemcpyDeviceToHost, stream));
        ACXCHECK(acx_ctx_sync(ctx));
        // ---- the same on one GPU
        std::vector<acx_fr> want(N + 1);
        uint64_t h_len = 0;
        int ok = 0;
        ACXCHECK(acx_qap_h(r_full, ww.data(), nullptr, want.data(), &h_len, &ok));
        if ((res[0] == 0) != (ok != 0) || (pass == 0) != (ok != 0)) ++mism;
        for (uint64_t i2l = 0; i2l < cw; ++i2l)
            for (uint64_t i1 = 0; i1 < R; ++i1)
                if (memcmp(&got[i2l * R + i1], &want[i1 * C + rank * cw + i2l], 32) != 0) ++mism;
        if (rank == 0)
            printf("pass %d: %s witness, %llu violated rows over all ranks, h(x) of degree %llu, this rank's block %s\n", pass,
                   ok ? "satisfying" : "corrupted", (unsigned long long)res[0], (unsigned long long)(h_len ? h_len - 1 : 0),
                   mism ? "MISMATCH" : "identical to the single-GPU h(x)");
    }
    unsigned long long* d_cnt;
    HIPCHECK(hipMalloc((void**)&d_cnt, 8));
    HIPCHECK(hipMemcpyAsync(d_cnt, &mism, 8, hipMemcpyHostToDevice, stream));
    NCCLCHECK(ncclAllReduce(d_cnt, d_cnt, 1, ncclUint64, ncclSum, comm, stream));
    unsigned long long total = 0;
    HIPCHECK(hipMemcpyAsync(&total, d_cnt, 8, hipMemcpyDeviceToHost, stream));
    ACXCHECK(acx_ctx_sync(ctx));
    if (rank == 0)
        printf("distributed h(x), 2^%u constraints over %d rank(s), six all-to-alls per h(x): %s\n", log_n, world,
               total == 0 ? "every block exact" : "MISMATCH");
    ncclCommDestroy(comm);
    acx_r1cs_destroy(r_local);
    acx_r1cs_destroy(r_full);
    acx_ctx_destroy(ctx);
    return total == 0 ? 0 : 1;
}
