/* oracle/acx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the evaluation-domain form of the reference's hot path
 * (sdiehl/arithmetic-circuits v0.2.0), used (1) as the large-size parity checker for the
 * HIP kernels and (2) as the timed "cpu_baseline" ("port") of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; libacx never does.
 *
 * What it restates (paths into /root/reference):
 *   - verifyAssignment / verificationWitnessZk  src/QAP.hs:276-327
 *       T | (L*R - O)  <=>  for every constraint row g: <A_g,w> * <B_g,w> - <C_g,w> = 0,
 *       where A_g[k] is the GenQAP value of wire k at root g (src/QAP.hs:366-474) -- the
 *       dot-product form named by BASELINE.json's north_star.
 *       h = (L*R - O) / (x^N - 1) is computed on the coset g*<omega_N>.
 *   - createPolynomialsFFT src/QAP.hs:512-525 -> FFT.interpolate (galois-fft-0.1.0, third
 *       party, not in the tree): column values in ascending root order, zero padded to
 *       N = 2^ceil(log2 n), inverse DFT over <omega_N> with P(omega_N^i) = v_i.
 *   - Prime-field arithmetic of galois-field-1.0.2 (canonical residues mod p).
 *
 * This file is validated against oracle/ref_qap.py (the literal restatement of the
 * reference's polynomial algorithm) in tests/test_oracle_cross.py; ref_qap.py in turn is
 * pinned on the reference's own known-answer tests.  Polynomial coefficients / FFT
 * outputs are never pinned by the reference's tests: "parity unpinned" for those values
 * beyond mathematical uniqueness (see oracle/ref_qap.py header).
 *
 * Elements cross this API as 32-byte little-endian canonical integers (4 x uint64).
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -pthread)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

typedef struct {
    fe p;          /* modulus */
    fe r2;         /* R^2 mod p, R = 2^256 */
    fe one;        /* R mod p */
    uint64_t n0;   /* -p^-1 mod 2^64 */
    int two_adicity;
    fe omega_max;  /* primitive 2^two_adicity-th root, Montgomery form */
    fe gen;        /* multiplicative generator used as coset shift, Montgomery form */
} orc_field;

/* ------------------------------------------------------------------ field */
static int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

static uint64_t fe_add_raw(fe *o, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; o->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t fe_sub_raw(fe *o, const fe *a, const fe *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        o->l[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static void fe_add(const orc_field *F, fe *o, const fe *a, const fe *b) {
    fe t; uint64_t c = fe_add_raw(&t, a, b);
    if (c || fe_geq(&t, &F->p)) fe_sub_raw(&t, &t, &F->p);
    *o = t;
}
static void fe_sub(const orc_field *F, fe *o, const fe *a, const fe *b) {
    fe t; if (fe_sub_raw(&t, a, b)) fe_add_raw(&t, &t, &F->p);
    *o = t;
}
/* Montgomery product a*b*R^-1 mod p: schoolbook 512-bit product, then word-by-word reduction. */
static void fe_mul(const orc_field *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[i + j];
            t[i + j] = (uint64_t)c; c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    uint64_t top = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t m = t[i] * F->n0;
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)m * F->p.l[j] + t[i + j];
            t[i + j] = (uint64_t)c; c >>= 64;
        }
        for (int k = i + 4; k < 8 && c; ++k) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
        top += (uint64_t)c;
    }
    fe r = {{t[4], t[5], t[6], t[7]}};
    if (top || fe_geq(&r, &F->p)) fe_sub_raw(&r, &r, &F->p);
    *o = r;
}
static void fe_to_mont(const orc_field *F, fe *o, const fe *a) { fe_mul(F, o, a, &F->r2); }
static void fe_from_mont(const orc_field *F, fe *o, const fe *a) {
    fe one = {{1, 0, 0, 0}}; fe_mul(F, o, a, &one);
}
static void fe_pow_u64(const orc_field *F, fe *o, const fe *a, uint64_t e) {
    fe acc = F->one, base = *a;
    while (e) { if (e & 1) fe_mul(F, &acc, &acc, &base); fe_mul(F, &base, &base, &base); e >>= 1; }
    *o = acc;
}
/* a^(p-2) */
static void fe_inv(const orc_field *F, fe *o, const fe *a) {
    fe e = F->p; fe two = {{2, 0, 0, 0}}; fe_sub_raw(&e, &e, &two);
    fe acc = F->one, base = *a;
    for (int i = 0; i < 256; ++i) {
        if ((e.l[i / 64] >> (i % 64)) & 1) fe_mul(F, &acc, &acc, &base);
        fe_mul(F, &base, &base, &base);
    }
    *o = acc;
}

/* ------------------------------------------------------------------ API: field */
/* Fill a field descriptor from canonical constants (modulus, generator g, two-adicity s):
 * omega_max = g^((p-1)/2^s).  Returns 0. */
int orc_field_init(orc_field *F, const uint64_t p[4], uint64_t generator, int two_adicity) {
    memset(F, 0, sizeof(*F));
    memcpy(F->p.l, p, 32);
    /* n0 = -p^-1 mod 2^64 by Newton iteration */
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
    F->n0 = (uint64_t)0 - inv;
    /* one = 2^256 mod p by 256 doublings of 1; r2 by 256 more */
    fe x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; ++i) {
        fe t; uint64_t c = fe_add_raw(&t, &x, &x);
        if (c || fe_geq(&t, &F->p)) fe_sub_raw(&t, &t, &F->p);
        x = t;
        if (i == 255) F->one = x;
    }
    F->r2 = x;
    F->two_adicity = two_adicity;
    fe g = {{generator, 0, 0, 0}}; fe_to_mont(F, &F->gen, &g);
    /* exponent (p-1) >> s */
    fe e = F->p; e.l[0] -= 1;
    for (int k = 0; k < two_adicity; ++k) {
        for (int i = 0; i < 4; ++i) e.l[i] = (e.l[i] >> 1) | (i < 3 ? e.l[i + 1] << 63 : 0);
    }
    fe acc = F->one, base = F->gen;
    for (int i = 0; i < 256; ++i) {
        if ((e.l[i / 64] >> (i % 64)) & 1) fe_mul(F, &acc, &acc, &base);
        fe_mul(F, &base, &base, &base);
    }
    F->omega_max = acc;
    return 0;
}
size_t orc_field_sizeof(void) { return sizeof(orc_field); }

/* out = a op b on canonical inputs; op: 0 add, 1 sub, 2 mul, 3 inv(a), 4 is-canonical(a) */
int orc_field_op(const orc_field *F, int op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    fe x, y, z; memcpy(&x, a, 32); if (b) memcpy(&y, b, 32);
    if (op == 4) return !fe_geq(&x, &F->p);
    fe_to_mont(F, &x, &x); if (b) fe_to_mont(F, &y, &y);
    switch (op) {
        case 0: fe_add(F, &z, &x, &y); break;
        case 1: fe_sub(F, &z, &x, &y); break;
        case 2: fe_mul(F, &z, &x, &y); break;
        case 3: fe_inv(F, &z, &x); break;
        default: return -1;
    }
    fe_from_mont(F, &z, &z); memcpy(out, &z, 32);
    return 0;
}
/* canonical primitive 2^k-th root of unity: pairing `getRootOfUnity k` */
int orc_root_of_unity(const orc_field *F, int k, uint64_t out[4]) {
    if (k < 0 || k > F->two_adicity) return -1;
    fe w = F->omega_max;
    for (int i = k; i < F->two_adicity; ++i) fe_mul(F, &w, &w, &w);
    fe_from_mont(F, &w, &w); memcpy(out, &w, 32);
    return 0;
}

/* ------------------------------------------------------------------ threads */
typedef void (*range_fn)(void *ctx, size_t lo, size_t hi, int tid);
typedef struct { range_fn fn; void *ctx; size_t lo, hi; int tid; } job_t;
static void *job_main(void *arg) { job_t *j = arg; j->fn(j->ctx, j->lo, j->hi, j->tid); return 0; }
static void parallel_for(size_t n, int nthreads, range_fn fn, void *ctx) {
    if (nthreads <= 1 || n < 2) { fn(ctx, 0, n, 0); return; }
    if ((size_t)nthreads > n) nthreads = (int)n;
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = malloc(sizeof(job_t) * nthreads);
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){fn, ctx, n * t / nthreads, n * (t + 1) / nthreads, t};
        pthread_create(&th[t], 0, job_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------ R1CS */
typedef struct {
    const uint32_t *rowptr; const uint32_t *col; const uint64_t *val; /* val: nnz x 4, canonical */
    const fe *val_mont;  /* optional: the same values already in Montgomery form */
} orc_csr;

typedef struct {
    const orc_field *F; size_t n, m; orc_csr M[3]; const fe *w_mont;
    int repeat;        /* baseline timing only: evaluate every row this many times */
    fe *dots[3];       /* optional outputs, Montgomery */
    uint64_t *res_out; /* optional residuals canonical, n x 4 */
    uint64_t *bad_count; uint64_t *first_bad; /* per thread */
} r1cs_job;

static void csr_dot(const orc_field *F, const orc_csr *M, const fe *w, size_t row, fe *acc) {
    fe a = {{0, 0, 0, 0}};
    for (uint32_t e = M->rowptr[row]; e < M->rowptr[row + 1]; ++e) {
        fe c, t;
        if (M->val_mont) c = M->val_mont[e];
        else { memcpy(&c, M->val + 4 * (size_t)e, 32); fe_to_mont(F, &c, &c); }
        fe_mul(F, &t, &c, &w[M->col[e]]);
        fe_add(F, &a, &a, &t);
    }
    *acc = a;
}
static void r1cs_range(void *vctx, size_t lo, size_t hi, int tid) {
    r1cs_job *J = vctx; const orc_field *F = J->F;
    uint64_t bad = 0, first = UINT64_MAX;
    for (int rep = 0; rep < (J->repeat > 1 ? J->repeat : 1); ++rep) {
    bad = 0; first = UINT64_MAX;
    for (size_t i = lo; i < hi; ++i) {
        fe a, b, c, r;
        csr_dot(F, &J->M[0], J->w_mont, i, &a);
        csr_dot(F, &J->M[1], J->w_mont, i, &b);
        csr_dot(F, &J->M[2], J->w_mont, i, &c);
        if (J->dots[0]) { J->dots[0][i] = a; J->dots[1][i] = b; J->dots[2][i] = c; }
        fe_mul(F, &r, &a, &b); fe_sub(F, &r, &r, &c);
        if (!fe_is_zero(&r)) { bad++; if (first == UINT64_MAX) first = i; }
        if (J->res_out) { fe_from_mont(F, &r, &r); memcpy(J->res_out + 4 * i, &r, 32); }
    }
    }
    J->bad_count[tid] = bad; J->first_bad[tid] = first;
}

static fe *witness_to_mont(const orc_field *F, const uint64_t *w, size_t m) {
    fe *wm = malloc(sizeof(fe) * (m ? m : 1));
    for (size_t k = 0; k < m; ++k) { fe t; memcpy(&t, w + 4 * k, 32); fe_to_mont(F, &wm[k], &t); }
    return wm;
}

/* Per-row residuals r_i = <A_i,w>*<B_i,w> - <C_i,w>.  residuals (n x 4, canonical) may be NULL.
 * n_bad = number of rows with r_i != 0; first_bad = smallest such row or UINT64_MAX. */
int orc_r1cs_residuals(const orc_field *F, uint64_t n, uint64_t m,
                       const uint32_t *a_rowptr, const uint32_t *a_col, const uint64_t *a_val,
                       const uint32_t *b_rowptr, const uint32_t *b_col, const uint64_t *b_val,
                       const uint32_t *c_rowptr, const uint32_t *c_col, const uint64_t *c_val,
                       const uint64_t *witness, uint64_t *residuals,
                       uint64_t *n_bad, uint64_t *first_bad, int nthreads, int repeat) {
    if (nthreads < 1) nthreads = 1;
    fe *wm = witness_to_mont(F, witness, m);
    r1cs_job J = {F, n, m, {{a_rowptr, a_col, a_val, 0}, {b_rowptr, b_col, b_val, 0}, {c_rowptr, c_col, c_val, 0}},
                  wm, repeat, {0, 0, 0}, residuals, calloc(nthreads, 8), calloc(nthreads, 8)};
    fe *vm[3] = {0, 0, 0};
    if (repeat > 1) {   /* baseline timing: convert the coefficients once, like the GPU engine does at load */
        for (int k = 0; k < 3; ++k) {
            size_t nnz = J.M[k].rowptr[n];
            vm[k] = malloc(sizeof(fe) * (nnz ? nnz : 1));
            for (size_t e = 0; e < nnz; ++e) { fe t; memcpy(&t, J.M[k].val + 4 * e, 32); fe_to_mont(F, &vm[k][e], &t); }
            J.M[k].val_mont = vm[k];
        }
    }
    for (int t = 0; t < nthreads; ++t) J.first_bad[t] = UINT64_MAX;
    parallel_for(n, nthreads, r1cs_range, &J);
    uint64_t bad = 0, first = UINT64_MAX;
    for (int t = 0; t < nthreads; ++t) { bad += J.bad_count[t]; if (J.first_bad[t] < first) first = J.first_bad[t]; }
    if (n_bad) *n_bad = bad;
    if (first_bad) *first_bad = first;
    free(J.bad_count); free(J.first_bad); free(wm);
    for (int k = 0; k < 3; ++k) free(vm[k]);
    return 0;
}

/* ------------------------------------------------------------------ NTT */
static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r;
}
typedef struct { const orc_field *F; fe *x; const fe *tw; size_t n; size_t half; } stage_job;
static void stage_range(void *vctx, size_t lo, size_t hi, int tid) {
    (void)tid; stage_job *S = vctx; const orc_field *F = S->F;
    size_t half = S->half, step = S->n / (2 * half);
    for (size_t b = lo; b < hi; ++b) {           /* butterfly index 0..n/2 */
        size_t grp = b / half, j = b % half, i0 = grp * 2 * half + j, i1 = i0 + half;
        fe t; fe_mul(F, &t, &S->x[i1], &S->tw[j * step]);
        fe u = S->x[i0];
        fe_add(F, &S->x[i0], &u, &t); fe_sub(F, &S->x[i1], &u, &t);
    }
}
/* In-place natural-order transform on Montgomery data:  X[k] = sum_i x[i] * w^(i k), w = root. */
static void ntt_core(const orc_field *F, fe *x, int log_n, const fe *root, int nthreads) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) { size_t j = bitrev((uint32_t)i, log_n); if (j > i) { fe t = x[i]; x[i] = x[j]; x[j] = t; } }
    if (log_n == 0) return;
    fe *tw = malloc(sizeof(fe) * (n / 2));
    tw[0] = F->one; for (size_t i = 1; i < n / 2; ++i) fe_mul(F, &tw[i], &tw[i - 1], root);
    for (size_t half = 1; half < n; half <<= 1) {
        stage_job S = {F, x, tw, n, half};
        parallel_for(n / 2, (n >= 4096) ? nthreads : 1, stage_range, &S);
    }
    free(tw);
}
static void root_for(const orc_field *F, int log_n, int inverse, fe *w) {
    *w = F->omega_max;
    for (int i = log_n; i < F->two_adicity; ++i) fe_mul(F, w, w, w);
    if (inverse) fe_inv(F, w, w);
}
static void ntt_mont(const orc_field *F, fe *x, int log_n, int inverse, const fe *coset, int nthreads) {
    size_t n = (size_t)1 << log_n; fe w; root_for(F, log_n, inverse, &w);
    if (!inverse && coset) { /* evaluate on coset: x[i] *= g^i first */
        fe gi = F->one; for (size_t i = 0; i < n; ++i) { fe_mul(F, &x[i], &x[i], &gi); fe_mul(F, &gi, &gi, coset); }
    }
    ntt_core(F, x, log_n, &w, nthreads);
    if (inverse) {
        fe nn = {{n, 0, 0, 0}}, ninv; fe_to_mont(F, &nn, &nn); fe_inv(F, &ninv, &nn);
        fe ginv = F->one, gi = F->one; if (coset) fe_inv(F, &ginv, coset);
        for (size_t i = 0; i < n; ++i) {
            fe_mul(F, &x[i], &x[i], &ninv);
            if (coset) { fe_mul(F, &x[i], &x[i], &gi); fe_mul(F, &gi, &gi, &ginv); }
        }
    }
}
/* data: batch x 2^log_n canonical elements, transformed in place.
 * inverse=0: X[k] = sum x[i] (g*w)^... i.e. evaluations p(shift * w^k); inverse=1 undoes it
 * (FFT.interpolate when shift == NULL).  shift: canonical coset generator or NULL. */
int orc_ntt(const orc_field *F, int log_n, uint64_t batch, int inverse, const uint64_t *shift,
            uint64_t *data, int nthreads) {
    if (log_n < 0 || log_n > F->two_adicity) return -1;
    size_t n = (size_t)1 << log_n;
    fe sh, *psh = 0; if (shift) { memcpy(&sh, shift, 32); fe_to_mont(F, &sh, &sh); psh = &sh; }
    fe *x = malloc(sizeof(fe) * n);
    for (uint64_t b = 0; b < batch; ++b) {
        uint64_t *d = data + 4 * n * b;
        for (size_t i = 0; i < n; ++i) { fe t; memcpy(&t, d + 4 * i, 32); fe_to_mont(F, &x[i], &t); }
        ntt_mont(F, x, log_n, inverse, psh, nthreads);
        for (size_t i = 0; i < n; ++i) { fe t; fe_from_mont(F, &t, &x[i]); memcpy(d + 4 * i, &t, 32); }
    }
    free(x);
    return 0;
}

/* ------------------------------------------------------------------ QAP columns */
/* createPolynomialsFFT for wires [wire_begin, wire_begin+wire_count) of one matrix:
 * out[w][0..N) = coefficients (canonical, NOT stripped) of the interpolant of column w. */
int orc_qap_columns(const orc_field *F, uint64_t n, int log_n,
                    const uint32_t *rowptr, const uint32_t *col, const uint64_t *val,
                    uint64_t wire_begin, uint64_t wire_count, uint64_t *out, int nthreads) {
    size_t N = (size_t)1 << log_n;
    if (n > N) return -1;
    memset(out, 0, wire_count * N * 32);
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            uint64_t k = col[e];
            if (k >= wire_begin && k < wire_begin + wire_count) {
                /* duplicates within a row accumulate (the loader never emits them) */
                fe cur, add, s; memcpy(&cur, out + 4 * ((k - wire_begin) * N + i), 32); memcpy(&add, val + 4 * (size_t)e, 32);
                uint64_t c = fe_add_raw(&s, &cur, &add); if (c || fe_geq(&s, &F->p)) fe_sub_raw(&s, &s, &F->p);
                memcpy(out + 4 * ((k - wire_begin) * N + i), &s, 32);
            }
        }
    return orc_ntt(F, log_n, wire_count, 1, 0, out, nthreads);
}

/* ------------------------------------------------------------------ h(x) */
/* verificationWitnessZk (src/QAP.hs:300-327) in the evaluation domain, FFT-path target x^N-1.
 * out_h: N+1 canonical coefficients (zero padded, NOT stripped); *ok = 1 iff remainder == 0,
 * i.e. every row residual is zero.  delta: 3 canonical elements or NULL (= 0,0,0). */
int orc_qap_h(const orc_field *F, uint64_t n, uint64_t m, int log_n,
              const uint32_t *a_rowptr, const uint32_t *a_col, const uint64_t *a_val,
              const uint32_t *b_rowptr, const uint32_t *b_col, const uint64_t *b_val,
              const uint32_t *c_rowptr, const uint32_t *c_col, const uint64_t *c_val,
              const uint64_t *witness, const uint64_t *delta, uint64_t *out_h, int *ok, int nthreads) {
    size_t N = (size_t)1 << log_n;
    if (n > N || log_n + 1 > F->two_adicity) return -1;
    if (nthreads < 1) nthreads = 1;
    fe *wm = witness_to_mont(F, witness, m);
    fe *d[3]; for (int k = 0; k < 3; ++k) d[k] = calloc(N + 1, sizeof(fe));
    r1cs_job J = {F, n, m, {{a_rowptr, a_col, a_val, 0}, {b_rowptr, b_col, b_val, 0}, {c_rowptr, c_col, c_val, 0}},
                  wm, 1, {d[0], d[1], d[2]}, 0, calloc(nthreads, 8), calloc(nthreads, 8)};
    parallel_for(n, nthreads, r1cs_range, &J);
    uint64_t bad = 0; for (int t = 0; t < nthreads; ++t) bad += J.bad_count[t];
    *ok = (bad == 0);
    free(J.bad_count); free(J.first_bad); free(wm);
    /* evaluations -> coefficients of L0, R0, O0 (degree < N) */
    for (int k = 0; k < 3; ++k) ntt_mont(F, d[k], log_n, 1, 0, nthreads);
    fe *L0 = malloc(sizeof(fe) * N), *R0 = malloc(sizeof(fe) * N);
    memcpy(L0, d[0], sizeof(fe) * N); memcpy(R0, d[1], sizeof(fe) * N);
    /* coset evaluations, shift g = field generator (g^N != 1) */
    for (int k = 0; k < 3; ++k) ntt_mont(F, d[k], log_n, 0, &F->gen, nthreads);
    fe gN, zinv; fe_pow_u64(F, &gN, &F->gen, (uint64_t)N); fe_sub(F, &gN, &gN, &F->one); fe_inv(F, &zinv, &gN);
    for (size_t i = 0; i < N; ++i) {
        fe t; fe_mul(F, &t, &d[0][i], &d[1][i]); fe_sub(F, &t, &t, &d[2][i]); fe_mul(F, &d[0][i], &t, &zinv);
    }
    ntt_mont(F, d[0], log_n, 1, &F->gen, nthreads);   /* h0 coefficients, degree <= N-2 when ok */
    fe *h = d[0]; memset(&h[N], 0, sizeof(fe));
    if (delta) {
        /* (L0+d1 T)(R0+d2 T) - (O0+d3 T) = h0 T + T (d1 R0 + d2 L0 + d1 d2 T - d3),  T = x^N - 1 */
        fe dl[3]; for (int k = 0; k < 3; ++k) { memcpy(&dl[k], delta + 4 * k, 32); fe_to_mont(F, &dl[k], &dl[k]); }
        fe d12; fe_mul(F, &d12, &dl[0], &dl[1]);
        for (size_t i = 0; i < N; ++i) {
            fe t; fe_mul(F, &t, &dl[0], &R0[i]); fe_add(F, &h[i], &h[i], &t);
            fe_mul(F, &t, &dl[1], &L0[i]); fe_add(F, &h[i], &h[i], &t);
        }
        fe_sub(F, &h[0], &h[0], &d12); fe_sub(F, &h[0], &h[0], &dl[2]);
        fe_add(F, &h[N], &h[N], &d12);
    }
    for (size_t i = 0; i <= N; ++i) { fe t; fe_from_mont(F, &t, &h[i]); memcpy(out_h + 4 * i, &t, 32); }
    free(L0); free(R0); for (int k = 0; k < 3; ++k) free(d[k]);
    return 0;
}

/* ------------------------------------------------------------------ the reference's OWN algorithm (baseline timing)
 * verifyAssignment / verificationWitness as src/QAP.hs:276-327 computes them -- in the POLYNOMIAL domain, on the dense
 * per-wire polynomials createPolynomialsFFT produces (src/QAP.hs:512-525), single threaded like the reference:
 *     L = sum_k w_k * A_k(x),  R, O alike      foldQapSet . combineWithDefaults (src/QAP.hs:163-181,226-230,314-323):
 *                                               one scalar x polynomial product and one polynomial sum per wire
 *     P = L * R - O                             dense polynomial product (poly's VPoly)
 *     (q, r) = P `quotRem` T,  T = x^N - 1      long division over the DENSE coefficient vector of T (src/QAP.hs:324-327)
 *     Just q  iff  r == 0
 * cols: 3 x m x N canonical coefficients (matrix, wire, coefficient: three orc_qap_columns outputs back to back),
 * witness: m canonical.  out_q: N canonical coefficients of the quotient or NULL.  *ok = (r == 0).
 * BASELINE.md section 2 "reference-algorithm mode": what the Haskell ALGORITHM costs on configs[0]; a C restatement, not
 * GHC.  The product of a canonical and a Montgomery operand is canonical, so nothing is converted per coefficient. */
int orc_ref_verify(const orc_field *F, uint64_t m, int log_n, const uint64_t *cols, const uint64_t *witness,
                   uint64_t *out_q, int *ok) {
    size_t N = (size_t)1 << log_n;
    fe *wm = witness_to_mont(F, witness, m);
    fe *S[3];
    for (int k = 0; k < 3; ++k) {
        S[k] = calloc(N, sizeof(fe));
        for (uint64_t j = 0; j < m; ++j) {
            const uint64_t *col = cols + 4 * (((size_t)k * m + j) * N);
            for (size_t i = 0; i < N; ++i) {
                fe c, t; memcpy(&c, col + 4 * i, 32);
                fe_mul(F, &t, &c, &wm[j]);               /* canonical x Montgomery = canonical */
                fe_add(F, &S[k][i], &S[k][i], &t);
            }
        }
    }
    /* P = L * R - O, 2N - 1 coefficients (+ one zero so that the division below can address P[2N - 1]) */
    fe *P = calloc(2 * N, sizeof(fe)), *Lm = malloc(sizeof(fe) * N);
    for (size_t i = 0; i < N; ++i) fe_to_mont(F, &Lm[i], &S[0][i]);
    for (size_t i = 0; i < N; ++i)
        for (size_t j = 0; j < N; ++j) {
            fe t; fe_mul(F, &t, &Lm[i], &S[1][j]);
            fe_add(F, &P[i + j], &P[i + j], &t);
        }
    for (size_t i = 0; i < N; ++i) fe_sub(F, &P[i], &P[i], &S[2][i]);
    /* long division by the dense T = (-1, 0, ..., 0, 1): quotient coefficient i - N = leading coefficient, then
     * P -= q * x^(i-N) * T over all N + 1 coefficients of T, zeros included (a dense-vector quotRem does not skip them) */
    fe *T = calloc(N + 1, sizeof(fe)), *Q = calloc(N, sizeof(fe));
    fe one_c = {{1, 0, 0, 0}}, zero = {{0, 0, 0, 0}};
    T[N] = one_c; fe_sub(F, &T[0], &zero, &one_c);
    for (size_t i = 2 * N - 1; i >= N; --i) {
        fe q = P[i], qm; Q[i - N] = q;
        fe_to_mont(F, &qm, &q);
        for (size_t j = 0; j <= N; ++j) {
            fe t; fe_mul(F, &t, &qm, &T[j]);
            fe_sub(F, &P[i - N + j], &P[i - N + j], &t);
        }
    }
    int zero_rem = 1;
    for (size_t i = 0; i < N; ++i) if (!fe_is_zero(&P[i])) { zero_rem = 0; break; }
    *ok = zero_rem;
    if (out_q) memcpy(out_q, Q, N * 32);
    free(wm); free(P); free(Lm); free(T); free(Q);
    for (int k = 0; k < 3; ++k) free(S[k]);
    return 0;
}
