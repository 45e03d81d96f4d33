/* examples/mgpu_host.c -- `verifyAssignment` and `verificationWitness` (/root/reference/src/QAP.hs:276-327) over SEVERAL GPUs from
 * ONE plain-C process: the shape of the reference's callers (one thread, one pure call; test/Test/Circuit/Arithmetic.hs:200-209).
 * The host binds nothing but include/acx.h -- no RCCL, no HIP, no MPI: libacx shards the rows over the devices, replicates the
 * witness and issues the collectives (one ncclAllReduce per verdict, one ncclAllToAll per transform) itself.
 *
 *   ./mgpu_host [device,device,...] [log2 rows]     default "0" 13
 * A list with repeated ordinals ("0,0,0,0") places several shards on one GPU (exchange by device copies): the multi-shard
 * code path on a one-GPU machine.  The same system is also loaded on a single-GPU context (acx_r1cs_load / acx_qap_h) and every
 * result must agree byte for byte.  exit: 0 ok, 77 no usable GPU (there is no CPU fallback), 1 wrong result.
 *
 * build: gcc -std=c11 -I include examples/mgpu_host.c -L arithmetic-circuits_amd -lacx -Wl,-rpath,$PWD/arithmetic-circuits_amd -o mgpu_host */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acx.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != ACX_OK) {                                                          \
            fprintf(stderr, "%s -> %d (%s: %s)\n", #call, rc_, acx_strerror(rc_), acx_last_error()); \
            return rc_ == ACX_ERR_NO_DEVICE ? 77 : 1;                                 \
        }                                                                             \
    } while (0)

static acx_fr fr_u64(uint64_t v) {
    acx_fr f;
    memset(&f, 0, sizeof f);
    for (int i = 0; i < 8; ++i) f.b[i] = (uint8_t)(v >> (8 * i));
    return f;
}

/* row i checks  w[a_i] * (3 + w[b_i]) = w[out_i]  with a_i, b_i among K inputs and one output wire per row: m = 1 + K + n */
enum { K = 64 };
static uint64_t a_of(uint64_t i) { return 1 + (i * 7 + 3) % K; }
static uint64_t b_of(uint64_t i) { return 1 + (i * 13 + 5) % K; }

int main(int argc, char** argv) {
    int ids[64];
    uint32_t n_dev = 0;
    char list[256];
    snprintf(list, sizeof list, "%s", argc > 1 ? argv[1] : "0");
    for (char* tok = strtok(list, ","); tok && n_dev < 64; tok = strtok(NULL, ",")) ids[n_dev++] = atoi(tok);
    const uint32_t log_n = argc > 2 ? (uint32_t)atoi(argv[2]) : 13;
    const uint64_t n = ((uint64_t)1 << log_n) - 5, m = 1 + K + n, N = (uint64_t)1 << log_n;   /* five rows of padding */

    /* the three matrices as plain CSR */
    uint32_t *pa = malloc((n + 1) * 4), *pb = malloc((n + 1) * 4), *pc = malloc((n + 1) * 4);
    uint32_t *ca = malloc(n * 4), *cb = malloc(2 * n * 4), *cc = malloc(n * 4);
    acx_fr *va = malloc(n * 32), *vb = malloc(2 * n * 32), *vc = malloc(n * 32), *w = malloc(m * 32);
    if (!pa || !pb || !pc || !ca || !cb || !cc || !va || !vb || !vc || !w) return 1;
    pa[0] = pb[0] = pc[0] = 0;
    w[0] = fr_u64(1);
    for (uint64_t k = 1; k <= K; ++k) w[k] = fr_u64(1000 + 17 * k);
    for (uint64_t i = 0; i < n; ++i) {
        ca[i] = (uint32_t)a_of(i); va[i] = fr_u64(1);
        cb[2 * i] = 0; vb[2 * i] = fr_u64(3);
        cb[2 * i + 1] = (uint32_t)b_of(i); vb[2 * i + 1] = fr_u64(1);
        cc[i] = (uint32_t)(1 + K + i); vc[i] = fr_u64(1);
        pa[i + 1] = (uint32_t)(i + 1); pb[i + 1] = (uint32_t)(2 * i + 2); pc[i + 1] = (uint32_t)(i + 1);
        w[1 + K + i] = fr_u64((1000 + 17 * a_of(i)) * (3 + 1000 + 17 * b_of(i)));
    }
    const acx_csr A = {pa, ca, va}, B = {pb, cb, vb}, C = {pc, cc, vc};

    /* several GPUs, one handle */
    acx_mgpu* mg = NULL;
    CHECK(acx_mgpu_create(ACX_FIELD_BN254_FR, ids, n_dev, &mg));
    CHECK(acx_mgpu_set_shard_threshold(mg, 10));
    int transport = -1;
    CHECK(acx_mgpu_info(mg, NULL, &transport, NULL));
    acx_mgpu_r1cs* mr = NULL;
    CHECK(acx_mgpu_r1cs_load(mg, n, m, &A, &B, &C, 0, &mr));
    uint32_t shards = 0;
    CHECK(acx_mgpu_r1cs_dims(mr, NULL, NULL, NULL, &shards));

    int ok = 0;
    uint64_t n_bad = 0, first = 0, h_len = 0;
    CHECK(acx_mgpu_r1cs_verify(mr, w, &ok, &n_bad, &first));                 /* verifyAssignment */
    if (!ok || n_bad) { fprintf(stderr, "valid witness rejected\n"); return 1; }
    acx_fr* h = malloc((N + 1) * 32);
    CHECK(acx_mgpu_qap_h(mr, w, NULL, h, &h_len, &ok));                      /* verificationWitness */
    if (!ok) { fprintf(stderr, "no h(x) for a valid witness\n"); return 1; }

    /* the same on ONE GPU through the single-device entry points */
    acx_ctx* ctx = NULL;
    acx_r1cs* r = NULL;
    CHECK(acx_ctx_create(ACX_FIELD_BN254_FR, ids[0], &ctx));
    CHECK(acx_r1cs_load(ctx, n, m, &A, &B, &C, &r));
    acx_fr* h1 = malloc((N + 1) * 32);
    uint64_t h1_len = 0;
    int ok1 = 0;
    CHECK(acx_qap_h(r, w, NULL, h1, &h1_len, &ok1));
    if (!ok1 || h1_len != h_len || memcmp(h, h1, h_len * 32) != 0) { fprintf(stderr, "h(x) differs from the single-GPU result\n"); return 1; }

    /* createPolynomialsFFT for 24 wires of A: the wires shared out over the devices, no exchange */
    {
        const uint64_t wires = 24, wb = 1 + K / 2;
        acx_fr* cols = malloc(wires * N * 32);
        acx_fr* cols1 = malloc(wires * N * 32);
        uint64_t lens[24], lens1[24];
        CHECK(acx_mgpu_qap_columns(mr, 0, wb, wires, cols, lens));
        CHECK(acx_qap_columns(r, 0, wb, wires, cols1, lens1));
        if (memcmp(cols, cols1, wires * N * 32) != 0 || memcmp(lens, lens1, sizeof lens) != 0) { fprintf(stderr, "QAP columns differ from the single-GPU result\n"); return 1; }
        free(cols); free(cols1);
    }

    /* a corrupted assignment: same count, same first violated row, Nothing */
    const uint64_t victim = 1 + K + n / 3;
    w[victim].b[0] ^= 1;
    uint64_t bad1 = 0, first1 = 0;
    CHECK(acx_mgpu_r1cs_verify(mr, w, &ok, &n_bad, &first));
    CHECK(acx_r1cs_verify(r, w, &ok1, &bad1, &first1));
    if (ok || ok1 || n_bad != bad1 || first != first1 || first != n / 3) { fprintf(stderr, "corrupted witness: verdicts differ\n"); return 1; }
    CHECK(acx_mgpu_qap_h(mr, w, NULL, h, &h_len, &ok));
    if (ok) { fprintf(stderr, "h(x) for an invalid witness\n"); return 1; }

    printf("Valid assignment; h(x) with %llu coefficients identical on %u shard(s) [%s] and on one GPU; corrupted copy: row %llu\n",
           (unsigned long long)h1_len, shards, transport == ACX_MGPU_RCCL ? "RCCL" : "peer copies", (unsigned long long)first);
    acx_r1cs_destroy(r);
    acx_ctx_destroy(ctx);
    acx_mgpu_r1cs_destroy(mr);
    acx_mgpu_destroy(mg);
    return 0;
}
