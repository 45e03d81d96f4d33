/* include/acx.h -- C ABI of libacx.so, the MI355X-native R1CS / QAP evaluation engine.
 *
 * The reference (sdiehl/arithmetic-circuits v0.2.0, pure Haskell) has NO FFI; this header IS
 * the drop-in boundary a Haskell host would bind with `foreign import ccall` to replace the
 * bodies of the `QAP` module's hot-path functions (export list src/QAP.hs:11-39) while keeping
 * their signatures.  Each entry point cites the reference interface it replaces
 * (paths into /root/reference).  INTEGRATION.md shows the Haskell-side binding.
 *
 * Conventions
 *   - Field element ("Fr"): 32 bytes, little-endian, canonical integer in [0, p)
 *     (Haskell side: `fromP` -> bytes).  Non-canonical input => ACX_ERR_NONCANONICAL.
 *   - Wire = {kind, index} as in `data Wire` (src/Circuit/Arithmetic.hs:32-36).
 *   - Flat witness vector w[m] follows `qapSetToMap` (src/QAP.hs:605-620): index 0 = constant
 *     wire, then inputs, intermediates, outputs; holes (unassigned wires) are 0.
 *   - Caller owns every host buffer; the library owns device memory behind opaque handles.
 *   - Every function returns 0 (ACX_OK) or a negative acx_status; nothing aborts or throws
 *     across the ABI (the reference's `panic` sites become error codes).
 *   - Entry points are thread-safe; each sets its device and uses its context's streams.  The host-buffer entry
 *     points that block on the GPU (acx_r1cs_verify, acx_r1cs_residuals, acx_qap_h, acx_qap_columns, acx_ntt) run
 *     concurrently for up to four callers per context (one HIP stream + scratch arena each): `safe` foreign
 *     calls from several Haskell capabilities overlap (test/Test/Circuit/Arithmetic.hs:209 maps verifyAssignment
 *     over many inputs).  Four callers with page-locked witness buffers (hipHostRegister / pinned allocations) reach
 *     1.7 - 1.9 times one caller's rate on every host measured; with pageable buffers the HIP runtime stages the copies
 *     itself and the same four callers gave between 0.8 and 1.9 times, depending on the host (profiles/r04_bench_line*.json,
 *     `e2e`).  The device-pointer entry points share the single stream acx_ctx_stream() returns.
 *   - There is NO CPU fallback: without a usable gfx950 device acx_ctx_create fails with
 *     ACX_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef ACX_H
#define ACX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACX_VERSION 0x000100

typedef enum acx_status {
    ACX_OK = 0,
    ACX_ERR_INVALID_ARG = -1,
    ACX_ERR_NONCANONICAL = -2,   /* element >= p */
    ACX_ERR_NO_DEVICE = -3,      /* no HIP device / wrong architecture */
    ACX_ERR_HIP = -4,            /* HIP runtime error (see acx_last_error) */
    ACX_ERR_ROOT_COUNT = -5,     /* src/QAP.hs:445,474 "wrong number of roots supplied" */
    ACX_ERR_UNDEFINED_WIRE = -6, /* src/Circuit/Arithmetic.hs:128,137 "the impossible happened" */
    ACX_ERR_DUPLICATE_ROOT = -7, /* roots must be distinct */
    ACX_ERR_TOO_LARGE = -8,      /* n > 2^two_adicity, or index overflow */
    ACX_ERR_OOM = -9,
    ACX_ERR_BAD_CIRCUIT = -10,   /* malformed marshalled gate list */
    ACX_ERR_UNSUPPORTED = -11
} acx_status;

typedef enum acx_field {
    ACX_FIELD_BN254_FR = 0,     /* pairing-1.0.0 Data.Pairing.BN254.Fr (bench/Circuit.hs:10) */
    ACX_FIELD_BLS12_381_FR = 1  /* field swap of BASELINE.json configs[4] */
} acx_field;

typedef struct acx_ctx acx_ctx;         /* device, streams, twiddle tables */
typedef struct acx_r1cs acx_r1cs;       /* device-resident GenQAP: CSR A/B/C (+CSC), row order */
typedef struct acx_circuit acx_circuit; /* host-side marshalled ArithCircuit */

typedef struct acx_fr { uint8_t b[32]; } acx_fr;

enum { ACX_WIRE_INPUT = 0, ACX_WIRE_INTERMEDIATE = 1, ACX_WIRE_OUTPUT = 2 };
typedef struct acx_wire { uint32_t kind; uint32_t index; } acx_wire;

enum { ACX_GATE_MUL = 0, ACX_GATE_EQUAL = 1, ACX_GATE_SPLIT = 2 };
enum { ACX_AFF_ADD = 0, ACX_AFF_SCALARMUL = 1, ACX_AFF_CONST = 2, ACX_AFF_VAR = 3 };

/* Marshalled `ArithCircuit f` = [Gate Wire f] (src/Circuit/Arithmetic.hs:44-59,149-150).
 * Affine circuits (src/Circuit/Affine.hs:26-31) are serialised in PRE-ORDER as token streams:
 *   ADD            -> followed by its two sub-trees
 *   SCALARMUL(arg) -> scalars[arg], followed by its sub-tree
 *   CONST(arg)     -> scalars[arg]
 *   VAR(arg)       -> aff_wires[arg]
 * Gate g owns tokens [tok_ofs[2g], tok_ofs[2g+1]) = mulLeft and [tok_ofs[2g+1], tok_ofs[2g+2]) =
 * mulRight (both empty for Equal/Split) and wires [wire_ofs[g], wire_ofs[g+1]):
 *   Mul:   {mulOutput}      Equal: {eqInput, eqMagic, eqOutput}     Split: {splitInput, splitOutputs...} */
typedef struct acx_gate_list {
    uint64_t n_gates;
    const uint8_t* kind;       /* [n_gates] ACX_GATE_* */
    const uint64_t* tok_ofs;   /* [2*n_gates + 1] */
    const uint8_t* tok_op;     /* [n_tokens] ACX_AFF_* */
    const uint32_t* tok_arg;   /* [n_tokens] */
    const acx_fr* scalars;     /* [n_scalars] canonical */
    uint64_t n_scalars;
    const acx_wire* aff_wires; /* [n_aff_wires] */
    uint64_t n_aff_wires;
    const uint64_t* wire_ofs;  /* [n_gates + 1] */
    const acx_wire* wires;     /* [wire_ofs[n_gates]] */
} acx_gate_list;

/* Host CSR view of one constraint matrix: row i holds the GenQAP values of constraint i. */
typedef struct acx_csr {
    const uint32_t* rowptr;    /* [n + 1] */
    const uint32_t* col;       /* [nnz] flat wire index */
    const acx_fr* val;         /* [nnz] canonical */
} acx_csr;

enum { ACX_MATRIX_A = 0, ACX_MATRIX_B = 1, ACX_MATRIX_C = 2 };

/* ---------------------------------------------------------------- library / context */
const char* acx_strerror(int status);
/* Thread-local detail of the last failure on the calling thread ("" if none). */
const char* acx_last_error(void);
uint32_t acx_version(void);

/* Replaces the type-class dictionary choice `GaloisField k` / `Fr` (src/QAP.hs:513,531) and the
 * `(Int -> k)` root-of-unity argument (src/QAP.hs:514; `getRootOfUnity`, bench/Circuit.hs:33):
 * the context owns the field, the device and the omega table omega_k = omega_max^(2^(s-k)). */
int acx_ctx_create(int field, int device_id, acx_ctx** out);
void acx_ctx_destroy(acx_ctx* ctx);
/* Optional override of the 2^k-th roots: omega must be a primitive 2^two_adicity-th root. */
int acx_ctx_set_root(acx_ctx* ctx, uint32_t two_adicity, const acx_fr* omega);
int acx_ctx_root_of_unity(acx_ctx* ctx, uint32_t k, acx_fr* out); /* `getRootOfUnity k` */
int acx_ctx_sync(acx_ctx* ctx);
/* HIP stream (hipStream_t) the context launches on; for event timing by a harness. */
void* acx_ctx_stream(acx_ctx* ctx);

/* ---------------------------------------------------------------- circuit (host marshalling) */
/* Pure host code: needs no device.  Copies and validates a marshalled gate list over the given
 * acx_field.  Wire numbering is fixed here:
 * num_inputs / num_intermediates / num_outputs = max index + 1 over every wire the circuit
 * mentions (src/QAP.hs:605-620 applies the same rule to the assignment's key sets).  Validation, the Montgomery
 * conversion of the scalars and the rows of every gate (gateToGenQAP) are computed here, over gate ranges on the host's
 * cores (ACX_HOST_THREADS overrides the thread count).
 * acx_circuit_destroy may be called at any time: systems built from the circuit stay valid (they keep what their
 * evaluation plan needs referenced until it has been derived). */
int acx_circuit_create(int field, const acx_gate_list* gates, acx_circuit** out);
void acx_circuit_destroy(acx_circuit* c);
int acx_circuit_dims(const acx_circuit* c, uint64_t* n_rows, uint64_t* m_wires,
                     uint64_t* n_inputs, uint64_t* n_intermediates, uint64_t* n_outputs);
/* `generateRoots` row count per gate (src/Circuit/Arithmetic.hs:194-216): Mul 1, Equal 2,
 * Split 1+#outputs.  out[n_gates]. */
int acx_circuit_rows_per_gate(const acx_circuit* c, uint32_t* out);
/* `validArithCircuit` (src/Circuit/Arithmetic.hs:158-185): *valid = 0/1. */
int acx_circuit_valid(const acx_circuit* c, int* valid);

/* `generateAssignment` = evalArithCircuit over initialQapSet (src/QAP.hs:591-603,
 * src/Circuit/Arithmetic.hs:106-145,221-235).  inputs[n_inputs] (present[i]==0 marks an absent
 * Map key; present may be NULL = all present).  Writes the flat witness w[m] (w[0] = 1) and, if
 * assigned != NULL, assigned[m] = 1 for wires the QapSet would hold.  Sequential over gates by
 * definition (host code). */
int acx_circuit_eval(const acx_circuit* c, const acx_fr* inputs, const uint8_t* present,
                     uint64_t n_inputs, acx_fr* witness, uint8_t* assigned);

/* `arithCircuitToGenQAP roots circuit` (src/QAP.hs:530-539): gateToGenQAP per gate
 * (src/QAP.hs:366-474, affineCircuitToAffineMap src/Circuit/Affine.hs:90-105), rows placed in
 * ascending-root order (`Map.elems`, src/QAP.hs:521-523).  roots: one per row in gate order
 * (n_roots must equal the row count, else ACX_ERR_ROOT_COUNT) or NULL for the `fresh` numbering
 * 0,1,2.. (src/Fresh.hs:16-20).  The result is device resident and never densified
 * (`addMissingZeroes` src/QAP.hs:566-576 is implicit). */
int acx_circuit_to_r1cs(acx_ctx* ctx, const acx_circuit* c, const acx_fr* roots, uint64_t n_roots,
                        acx_r1cs** out);
/* The reference takes roots as one list PER GATE (`[[k]]`, src/QAP.hs:530-539) and panics when a gate's list has
 * the wrong length (src/QAP.hs:444-445,474).  A host that flattens the lists itself calls this first: counts[g] =
 * length of gate g's list, n_lists = number of lists; ACX_ERR_ROOT_COUNT unless n_lists == #gates and every
 * count equals the gate's row count.  This flat form is STRICT: duplicate roots are an error (ACX_ERR_DUPLICATE_ROOT) and so
 * are surplus or missing lists.  A drop-in host uses acx_circuit_to_r1cs_lists with ACX_ROOTS_REFERENCE_SEMANTICS instead,
 * which returns what the reference returns on such lists. */
int acx_circuit_check_root_counts(const acx_circuit* c, const uint32_t* counts, uint64_t n_lists);
/* `arithCircuitToGenQAP rootsPerGate circuit` (src/QAP.hs:530-539) taking the roots as the reference does -- one list PER
 * GATE: roots = the lists concatenated, counts[g] = length of list g, n_lists = number of lists -- so that a host need not
 * check or flatten anything itself.  flags = 0: the strict contract above (ACX_ERR_ROOT_COUNT unless there is one list per gate
 * of the gate's row count, ACX_ERR_DUPLICATE_ROOT on a repeated root).  flags = ACX_ROOTS_REFERENCE_SEMANTICS: the
 * reference's own result on degenerate lists, value for value --
 *   - `zipWith` (src/QAP.hs:539): lists beyond the last gate make no rows; gates beyond the last list are dropped;
 *   - `Map.fromList` per wire (src/QAP.hs:233-239): of two rows with the SAME root the later one wins on every wire it
 *     mentions (the constant and the explicit zeros of src/QAP.hs:396-473 included), other wires keep the earlier value;
 *   - `addMissingZeroes (concat rootsPerGate)` (src/QAP.hs:566-576): every root of every list owns a row, zero if no gate
 *     wrote it.
 * The system then has one row per DISTINCT root, in ascending order; a list whose length does not fit its gate stays the
 * reference's panic (src/QAP.hs:444-445,474): ACX_ERR_ROOT_COUNT.  Regular lists (what `generateRoots` produces) take the same
 * path as acx_circuit_to_r1cs; a degenerate system has no GPU evaluation plan (acx_r1cs_eval: ACX_ERR_UNSUPPORTED). */
enum { ACX_ROOTS_REFERENCE_SEMANTICS = 1 };
int acx_circuit_to_r1cs_lists(acx_ctx* ctx, const acx_circuit* c, const acx_fr* roots, const uint32_t* counts, uint64_t n_lists,
                              uint32_t flags, acx_r1cs** out);
/* The same rows on the host (pure host code, no device).  Every output may be NULL: call once for *n_rows (the number of
 * distinct roots) and *nnz, then again with rowptr[*n_rows + 1], col / val[*nnz] and sorted_roots[*n_rows] (the distinct roots
 * ascending = the abscissae of the naive path, `createPolynomials` src/QAP.hs:486-508). */
int acx_circuit_rows_lists(const acx_circuit* c, const acx_fr* roots, const uint32_t* counts, uint64_t n_lists, uint32_t flags,
                           int matrix, uint64_t* n_rows, uint64_t* nnz, uint32_t* rowptr, uint32_t* col, acx_fr* val,
                           acx_fr* sorted_roots);
/* The same rows on the host (pure host code), e.g. for a multi-GPU host that shards rows before
 * acx_r1cs_load.  Call acx_circuit_nnz first to size the buffers: rowptr[n_rows+1], col/val[nnz]. */
int acx_circuit_nnz(const acx_circuit* c, uint64_t nnz[3]);
int acx_circuit_rows(const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, int matrix,
                     uint32_t* rowptr, uint32_t* col, acx_fr* val);

/* ---------------------------------------------------------------- R1CS / GenQAP (device) */
/* Pre-flattened constraint system (a GenQAP in row form): n rows, m wires.  Rows need not be sorted by
 * column; duplicate columns in a row are summed.  Limits: 1 <= m < 2^32 - 1, n < 2^32 - 1, n <= 2^two-adicity
 * of the field (2^28 for BN254 Fr, 2^32 for BLS12-381 Fr), fewer than 2^32 entries per matrix
 * (ACX_ERR_TOO_LARGE otherwise); columns must be < m and values canonical (ACX_ERR_INVALID_ARG /
 * ACX_ERR_NONCANONICAL).  The structure of all three matrices is checked on the host before any value is
 * uploaded; canonicity is checked on the device, one flag for the whole load: of a structural defect and a
 * non-canonical value in the same call the structural defect is the one reported. */
int acx_r1cs_load(acx_ctx* ctx, uint64_t n, uint64_t m, const acx_csr* A, const acx_csr* B,
                  const acx_csr* C, acx_r1cs** out);
void acx_r1cs_destroy(acx_r1cs* r);
int acx_r1cs_dims(const acx_r1cs* r, uint64_t* n, uint64_t* m, uint32_t* log_n, uint64_t nnz[3]);
/* How the residual kernel stores the system (introspection; results never depend on it).  small_mask bit k (A, B, C):
 * every coefficient of matrix k's rows of <= 6 entries is c or p - c with c <= 2^27 -- the shape programs compile to
 * (src/Circuit/Expr.hs:256-305: +-1, +-2, small constants) -- and the matrix is held as {coefficient, column} pairs of
 * 8 bytes with no 32-byte value stream; a term is then nine multiply-adds instead of a 81-multiply product.
 * unit_c: every C coefficient is 1 (src/QAP.hs:406-409: a Mul gate's output row).  n_long: rows with more than 6 entries
 * in some matrix (Split gates), which take the CSR kernel.  Any pointer may be NULL. */
int acx_r1cs_format(const acx_r1cs* r, uint32_t* small_mask, uint32_t* unit_c, uint64_t* n_long);
/* Copy one matrix back as canonical CSR (caller sizes buffers from acx_r1cs_dims). */
int acx_r1cs_export(const acx_r1cs* r, int matrix, uint32_t* rowptr, uint32_t* col, acx_fr* val);

/* `verifyAssignment qap assignment` (src/QAP.hs:276-282) in the evaluation domain:
 * *ok = 1 iff every row satisfies <A_i,w>*<B_i,w> - <C_i,w> = 0  (<=> T | L*R-O, roots distinct).
 * n_bad = number of violated rows, first_bad = smallest violated row (UINT64_MAX if none);
 * either may be NULL.  witness[m] canonical (absent wires = 0, `combineWithDefaults`
 * src/QAP.hs:163-181,314). */
int acx_r1cs_verify(acx_r1cs* r, const acx_fr* witness, int* ok, uint64_t* n_bad, uint64_t* first_bad);
/* `all (verifyAssignment qap . generateAssignment program) inputs` (test/Test/Circuit/Arithmetic.hs:200-209: build the
 * QAP once, verify many assignments) in ONE call: witnesses = count x m canonical elements, witness k at
 * witnesses[k*m].  ok[k] (and n_bad[k], first_bad[k] when given) as acx_r1cs_verify would report them.  The witnesses
 * cross PCIe in one copy per chunk (256 MiB of device memory per chunk) and a chunk is verified by one batched
 * launch; any non-canonical element fails the whole call with ACX_ERR_NONCANONICAL. */
int acx_r1cs_verify_many(acx_r1cs* r, uint64_t count, const acx_fr* witnesses, uint8_t* ok, uint64_t* n_bad,
                         uint64_t* first_bad);
/* `generateAssignment` on the GPU: the gates are evaluated level by level (a
 * level = gates whose inputs are all produced by earlier levels), one launch per level (one launch
 * per RUN of levels of at most 128 gates: small circuits are a single launch); Mul gates
 * reuse their own constraint rows; the magic wires of Equal gates (inverses, read by no gate of a
 * valid circuit) are filled by one launch after the last level -- inside the levels when some gate does read one.  Available for systems built by acx_circuit_to_r1cs from a
 * circuit in single-assignment form (else ACX_ERR_UNSUPPORTED: use acx_circuit_eval).  Same
 * arguments and results as acx_circuit_eval; witness/assigned may be NULL.  The witness also stays
 * resident on the device for acx_r1cs_verify_resident.  The plan (levels, per-gate records) is derived on the first
 * call, not at load: a caller that only verifies never pays for it. */
int acx_r1cs_eval(acx_r1cs* r, const acx_fr* inputs, const uint8_t* present, uint64_t n_inputs,
                  acx_fr* witness, uint8_t* assigned);
/* verifyAssignment of the witness left on the device by acx_r1cs_eval. */
int acx_r1cs_verify_resident(acx_r1cs* r, int* ok, uint64_t* n_bad, uint64_t* first_bad);
/* Residual vector r_i (canonical), out[n]. */
int acx_r1cs_residuals(acx_r1cs* r, const acx_fr* witness, acx_fr* out);

/* `verificationWitnessZk d1 d2 d3 qap assignment` (src/QAP.hs:300-327) for the FFT-path target
 * x^N - 1: *ok = 0 => Nothing; else out_h[0..*h_len) are the coefficients of the quotient,
 * low to high, trailing zeros stripped (poly `toPoly`).  out_h must hold N+1 elements.
 * delta = 3 elements or NULL (= verificationWitness, src/QAP.hs:292-298). */
int acx_qap_h(acx_r1cs* r, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h,
              uint64_t* h_len, int* ok);

/* `createPolynomialsFFT primRoots genQap` (src/QAP.hs:512-525) for wires
 * [wire_begin, wire_begin + wire_count) of one matrix: FFT.interpolate of each column =
 * out[w*N .. w*N+N) coefficients (canonical, zero padded to N; degree+1 in out_len[w] if
 * out_len != NULL, i.e. `toPoly` stripping).  Target polynomial is x^N - 1 (implicit). */
int acx_qap_columns(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count,
                    acx_fr* out, uint64_t* out_len);

/* ---------------------------------------------------------------- naive-roots path
 * `createPolynomials` / `arithCircuitToQAP` (src/QAP.hs:486-508,542-549): Lagrange interpolation
 * on ARBITRARY distinct roots, target T(x) = prod (x - r_i) -- what the reference's unit tests use
 * (roots 7,8,9: test/Test/QAP.hs:73-74).  roots[i] belongs to row i of the system, which
 * acx_circuit_to_r1cs stores in ascending-root order, so roots must be strictly ascending.
 * O(n^2) like the reference's ("terrible complexity", src/QAP.hs:483-485); n <= 4096.
 * verifyAssignment itself needs no roots: use acx_r1cs_verify. */
typedef struct acx_naive acx_naive;
int acx_naive_create(acx_r1cs* r, const acx_fr* roots, uint64_t n_roots, acx_naive** out);
void acx_naive_destroy(acx_naive* nv);
/* qapTarget: n + 1 coefficients, low to high (monic). */
int acx_naive_target(acx_naive* nv, acx_fr* out);
/* Per-wire polynomials of one matrix: out[w*n .. w*n+n), out_len[w] = stripped length. */
int acx_naive_columns(acx_naive* nv, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out,
                      uint64_t* out_len);
/* verificationWitnessZk on the naive QAP: quotient of (L*R - O) by T; out_h holds n + 1 elements. */
int acx_naive_h(acx_naive* nv, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h, uint64_t* h_len,
                int* ok);

/* ---------------------------------------------------------------- NTT (replaces galois-fft) */
/* batch independent length-2^log_n transforms on host data, in place semantic (in may == out).
 * inverse = 0: out[k] = sum_i in[i] * (shift*omega^k)^i  (`FFT.fft`; shift NULL = 1)
 * inverse = 1: the inverse map (`FFT.interpolate` without stripping when shift == NULL). */
int acx_ntt(acx_ctx* ctx, uint32_t log_n, uint64_t batch, int inverse, const acx_fr* shift,
            const acx_fr* in, acx_fr* out);

/* ---------------------------------------------------------------- device-pointer variants
 * Same computations on caller-provided DEVICE buffers, asynchronous on acx_ctx_stream(ctx).
 * Device element format is opaque ("dev" = 32-byte internal Montgomery form); convert with
 * acx_dev_from_canonical / acx_dev_to_canonical.  Used by multi-GPU hosts (one process per
 * GPU) that keep data resident between collectives, and by bench.py. */
int acx_dev_from_canonical(acx_ctx* ctx, uint64_t count, const void* d_in, void* d_out, uint32_t* d_err);
int acx_dev_to_canonical(acx_ctx* ctx, uint64_t count, const void* d_in, void* d_out);
/* d_witness: m dev elements.  d_result: 2 x uint64 {n_bad, first_bad}, accumulated with
 * atomicAdd / atomicMin so that several shards can target one buffer; caller initialises it to
 * {0, UINT64_MAX}.  row_offset is added to the local row index for first_bad.
 * d_residuals (n dev elements) and d_dots (3*N dev elements: <A,w>,<B,w>,<C,w>) may be NULL. */
int acx_r1cs_verify_dev(acx_r1cs* r, const void* d_witness, uint64_t row_offset, uint64_t* d_result,
                        void* d_residuals, void* d_dots);
int acx_ntt_dev(acx_ctx* ctx, uint32_t log_n, uint64_t batch, int inverse, const acx_fr* shift,
                void* d_data);
/* `verificationWitnessZk` (src/QAP.hs:300-327) on device-resident data: d_witness m dev elements; d_h receives
 * N+1 dev elements = the quotient's coefficients low to high, zero beyond its degree (not stripped);
 * d_result {n_bad, first_bad} accumulates like acx_r1cs_verify_dev (n_bad != 0 <=> `Nothing`).  delta may be NULL. */
int acx_qap_h_dev(acx_r1cs* r, const void* d_witness, const acx_fr* delta, void* d_h, uint64_t* d_result);
/* `createPolynomialsFFT` (src/QAP.hs:512-525) with the results left on the device: d_out receives wire_count * N
 * dev elements (column w at d_out + w*N), d_len (may be NULL) the stripped lengths (`toPoly`), computed by a
 * reduction on the device.  The column view of the matrix is built on the device on first use. */
int acx_qap_columns_dev(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count, void* d_out,
                        uint64_t* d_len);

/* Local step of the DISTRIBUTED four-step NTT (replaces galois-fft at src/QAP.hs:521-524 when one
 * transform spans several GPUs, BASELINE.json configs[3]).  N = 2^log_n = R*C with R = 2^log_r; index split
 * i = i1*C + i2 (input side), k = k1 + k2*R (output side); `world` ranks (power of two dividing R and C), one
 * process per GPU; rank g owns the i2 block g of the input side and the k1 block g of the output side.
 * Local layouts, N/world dev elements each:
 *     COLS  [i2l][i1]         x[i1*C + g*C/W + i2l]
 *     ROWS  [kl][k2]          X[(g*R/W + kl) + k2*R]
 *     XCHG  [peer][kl][i2l]   W contiguous chunks of (R/W)*(C/W) elements = the send / receive buffer of ONE
 *                             all-to-all (RCCL ncclAllToAll / torch all_to_all_single), issued by the host
 *   forward (inverse = 0):  step 0: COLS -> XCHG,  all-to-all,  step 1: XCHG -> ROWS
 *   inverse (inverse = 1):  step 0: ROWS -> XCHG,  all-to-all,  step 1: XCHG -> COLS
 * Each step is one kernel launch: the transposes are strides of the pass, the w_N^(i2*k1) twiddle (and the 1/N of
 * an inverse transform) is the kernel's closing multiplication from tables.  shift != NULL: coset transform
 * (forward evaluates on shift*<omega>, inverse undoes it).  5 <= log_r, log_n - log_r <= 12; d_in != d_out.
 * world = 1 is the degenerate case (no exchange needed: XCHG of step 0 is the input of step 1). */
int acx_ntt_dist_step_dev(acx_ctx* ctx, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse,
                          int step, const acx_fr* shift, const void* d_in, void* d_out);
/* The same with flags.  ACX_DIST_ROWS_T (inverse step 0 only): d_in is the rank's ROWS block stored TRANSPOSED, [k2][kl] --
 * i.e. the rank's rows in ASCENDING order (runs of R/W consecutive rows, one run out of every R).  A host that loads its
 * block-cyclic rows in that order gets their dot products from acx_r1cs_verify_dev in this layout, and the residual kernel's
 * gathers then stay inside a window 8x narrower than in ROWS order (residual without outputs: 216 -> 165 us per 2^21 rows). */
enum { ACX_DIST_ROWS_T = 1 };
int acx_ntt_dist_step_ex_dev(acx_ctx* ctx, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse,
                             int step, uint32_t flags, const acx_fr* shift, const void* d_in, void* d_out);
/* The two fused forms the h(x) pipeline uses (src/QAP.hs:292-327; DESIGN.md section 4 "h(x) in round 3"):
 *   d_mul != NULL: the step transforms the POINTWISE PRODUCT d_in[i] * d_mul[i] (same layout) -- L * R on the coset is formed as
 *                  the points are loaded, no product vector exists (not on step 0 of a forward coset transform);
 *   d_add != NULL: d_out[k] = X[k] + d_add[k] (the layout of d_out) -- the coefficient-domain -O/z joins behind the closing
 *                  multiplication of the last step.
 * Both NULL: acx_ntt_dist_step_ex_dev. */
int acx_ntt_dist_step_fused_dev(acx_ctx* ctx, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse,
                                int step, uint32_t flags, const acx_fr* shift, const void* d_in, const void* d_mul,
                                const void* d_add, void* d_out);
/* acx_r1cs_verify_dev storing the dot products FOR h(x) over a transform of 2^h_log_n points (the GLOBAL size: a rank's own
 * system has N / world rows): <A_i,w> is stored times 1/z and <C_i,w> times -1/z, z = shift^N - 1 (shift = the coset the
 * host's transforms use; NULL = the field's generator, which acx_qap_h uses), so that (L/z) R + (-O/z) needs no pointwise
 * pass and no scaled subtraction later; the verdict comes from the plain values.  d_dots: 3 * 2^(log_n of r) dev elements. */
int acx_r1cs_dots_h_dev(acx_r1cs* r, const void* d_witness, uint64_t row_offset, uint64_t* d_result, void* d_dots,
                        uint32_t h_log_n, const acx_fr* shift);

/* The pointwise step of h(x) on a coset (src/QAP.hs:325-327 in evaluation form): out[i] = (a[i]*b[i] - c[i]) /
 * (shift^N - 1), N = 2^log_n, on `count` dev elements (any slice of the evaluation vectors: the operation is
 * layout agnostic, which is what lets the distributed pipeline keep its block layouts).
 * d_c = NULL: out[i] = a[i]*b[i] / (shift^N - 1).  The pipelines use this form: the transforms are linear and a coset
 * transform followed by its inverse is the identity, so O(x) never needs its coset evaluations --
 * h = icoset(L*R / z) - O / z with O in COEFFICIENT form (acx_qap_sub_o_dev), six transforms per h(x) instead of seven. */
int acx_qap_pointwise_dev(acx_ctx* ctx, uint32_t log_n, uint64_t count, const acx_fr* shift, const void* d_a,
                          const void* d_b, const void* d_c, void* d_out);

/* d_h[i] -= d_o[i] / (shift^N - 1) on `count` dev elements: the coefficient-domain half of the quotient above
 * (h and O's coefficients in the same layout; in the distributed pipeline both are in COLS ownership). */
int acx_qap_sub_o_dev(acx_ctx* ctx, uint32_t log_n, uint64_t count, const acx_fr* shift, void* d_h, const void* d_o);

/* Batched verification: `count` independent (constraint system, witness) pairs checked by ONE
 * kernel launch -- the shape of the reference's property tests, `all (verifyAssignment qap .
 * generateAssignment program) inputs` (test/Test/Circuit/Arithmetic.hs:200-209), and of a
 * constraint system stored as independent blocks.  d_witnesses[i]: m_i dev elements.
 * result_stride = 2: pair i accumulates into d_results[2i..2i+1] = {n_bad, first_bad};
 * result_stride = 0: every pair accumulates into d_results[0..1], first_bad counted over the
 * concatenation of the systems' rows.  The caller initialises d_results ({0, UINT64_MAX}). */
typedef struct acx_batch acx_batch;
int acx_batch_create(acx_ctx* ctx, uint64_t count, acx_r1cs* const* systems, const void* const* d_witnesses,
                     uint64_t* d_results, uint64_t result_stride, acx_batch** out);
int acx_batch_verify_dev(acx_batch* batch);
void acx_batch_destroy(acx_batch* batch);

/* ---------------------------------------------------------------- one process, several GPUs
 * The reference's callers make ONE pure call from one thread -- `verifyAssignment qap assignment` (src/QAP.hs:276-282),
 * `all (verifyAssignment qap . generateAssignment program) inputs` (test/Test/Circuit/Arithmetic.hs:200-209),
 * `verificationWitness` (src/QAP.hs:292-327) -- so a drop-in host reaches the GPUs of a node through THIS handle, with the
 * same call shapes as the single-GPU entry points and nothing else to bind: the library shards the constraint rows over the
 * devices (block-cyclic, SURVEY.md 8e: shard g owns the rows k with (k mod R) in block g of R / n_devices, N = R * C), replicates
 * the witness (ONE host-to-device copy and conversion, then ncclBroadcast over the fabric; ACX_MGPU_WITNESS=copies|pinned: one
 * host-to-device copy per GPU instead, from pageable or from page-locked memory), and issues the collectives itself over RCCL:
 * ONE ncclAllReduce for the verdict of a check, ONE ncclAllToAll per transform (six per h(x)), overlapped with the local steps
 * on a second stream per GPU.  Every GPU has its own issuing host thread for the life of the handle (its own communicator:
 * nothing crosses threads on the host), so the API calls of the N per-GPU pipelines are made side by side.
 * BASELINE.json configs[3] (2^24 constraints over 8 GPUs) is `acx_mgpu_create(field, {0..7}, 8, &mg)` + the calls below.
 *
 * device_ids: n_devices (a power of two, <= 64) HIP device ordinals.  Distinct ids: RCCL (bound with dlopen here, so a
 * single-GPU user of libacx never maps it); ACX_MGPU_TRANSPORT=peer selects hipMemcpyPeerAsync copies and a host-side sum
 * instead (DMA engines over xGMI, no collective kernels).  A list with REPEATED ids places several shards on one GPU --
 * which RCCL cannot do, so the peer-copy transport is used: the n_devices = 2 / 4 / 8 paths on a one-GPU machine.
 * Results never depend on n_devices or the transport: every output below is bit-identical to the single-GPU call.
 *
 * Systems below the shard threshold (N < 2^14 by default, acx_mgpu_set_shard_threshold; never below 2^10 nor above 2^24
 * for h(x), 2 n_devices <= sqrt(N)) are held whole on the first device and every call on them is the single-GPU one.
 * acx_mgpu_* calls on one handle are serialised (the collectives are ordered); different handles are independent. */
typedef struct acx_mgpu acx_mgpu;
typedef struct acx_mgpu_r1cs acx_mgpu_r1cs;
enum { ACX_MGPU_RCCL = 0, ACX_MGPU_PEER_COPY = 1 };
enum { ACX_MGPU_VERIFY_ONLY = 1 };      /* load flag: no block-cyclic copy (acx_mgpu_qap_h then reports ACX_ERR_UNSUPPORTED) */

/* replaces acx_ctx_create for a multi-GPU host (SURVEY.md 8b proposed `device_ids[], n_devices`) */
int acx_mgpu_create(int field, const int* device_ids, uint32_t n_devices, acx_mgpu** out);
void acx_mgpu_destroy(acx_mgpu* mg);
int acx_mgpu_info(const acx_mgpu* mg, uint32_t* n_devices, int* transport, uint32_t* shard_threshold_log_n);
/* the context of one shard (its device, stream, root table): acx_ctx_root_of_unity, acx_ctx_stream for event timing */
acx_ctx* acx_mgpu_ctx(acx_mgpu* mg, uint32_t shard);
int acx_mgpu_set_shard_threshold(acx_mgpu* mg, uint32_t log_n);
int acx_mgpu_set_root(acx_mgpu* mg, uint32_t two_adicity, const acx_fr* omega);      /* acx_ctx_set_root on every shard */
int acx_mgpu_sync(acx_mgpu* mg);

/* acx_r1cs_load / acx_circuit_to_r1cs (`arithCircuitToGenQAP`, src/QAP.hs:530-539) with the rows sharded over the devices:
 * each shard's rows are gathered and uploaded by its own host thread; no device ever holds the whole system.  Two row
 * ownerships are kept (memory is plentiful: 288 GB per GPU, a 2^24-constraint system is ~1 GB per GPU per copy): contiguous
 * slabs balanced by entry count for verifyAssignment (the rows in flight on a GPU then gather from one narrow window of the
 * witness: 1.5-2x on the residual kernel), and the block-cyclic rows whose dot products ARE the transforms' input layout for
 * h(x).  flags: 0, or ACX_MGPU_VERIFY_ONLY to skip the second copy. */
int acx_mgpu_r1cs_load(acx_mgpu* mg, uint64_t n, uint64_t m, const acx_csr* A, const acx_csr* B, const acx_csr* C,
                       uint32_t flags, acx_mgpu_r1cs** out);
int acx_mgpu_circuit_to_r1cs(acx_mgpu* mg, const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, uint32_t flags,
                             acx_mgpu_r1cs** out);
void acx_mgpu_r1cs_destroy(acx_mgpu_r1cs* r);
int acx_mgpu_r1cs_dims(const acx_mgpu_r1cs* r, uint64_t* n, uint64_t* m, uint32_t* log_n, uint32_t* n_shards);

/* `verifyAssignment` (src/QAP.hs:276-282) over all devices: arguments and results of acx_r1cs_verify (first_bad = the smallest
 * violated GLOBAL row; passing NULL saves the second all-reduce of a failing check). */
int acx_mgpu_r1cs_verify(acx_mgpu_r1cs* r, const acx_fr* witness, int* ok, uint64_t* n_bad, uint64_t* first_bad);
/* `all (verifyAssignment qap) assignments` (test/Test/Circuit/Arithmetic.hs:200-209) over all devices in ONE call: witnesses =
 * count x m canonical elements; ok[k] (and n_bad[k] when given) as acx_mgpu_r1cs_verify reports them.  Witness k+1 crosses PCIe
 * while witness k is checked; the verdicts of up to 16 witnesses share one all-reduce.  Any non-canonical element fails the
 * whole call with ACX_ERR_NONCANONICAL. */
int acx_mgpu_r1cs_verify_many(acx_mgpu_r1cs* r, uint64_t count, const acx_fr* witnesses, uint8_t* ok, uint64_t* n_bad);
/* `verificationWitnessZk` (src/QAP.hs:300-327) over all devices: arguments and results of acx_qap_h (out_h holds N + 1 elements).
 * Total over everything acx_qap_h accepts: a transform size the distributed four-step form does not cover (N above 2^24, or
 * fewer than 2 * n_devices points per digit) is answered from ONE device, on a copy of the whole system it receives on the
 * first such call.  Only a system loaded with ACX_MGPU_VERIFY_ONLY refuses (ACX_ERR_UNSUPPORTED). */
int acx_mgpu_qap_h(acx_mgpu_r1cs* r, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h, uint64_t* h_len, int* ok);
/* `createPolynomialsFFT primRoots genQap` (src/QAP.hs:512-525) for a wire range of one matrix, the wires shared out over
 * the devices (columns are independent: no exchange at all): arguments and results of acx_qap_columns, every
 * device writing its wires' coefficients straight into `out`.  Wires are owned block-cyclically (64 consecutive wires per
 * block, block j on device j mod n_devices), so any request of a few hundred wires spreads over all devices.  A column's
 * interpolation needs its own column of every row and nothing else: the FIRST call builds, on every device, the column view
 * of THAT device's wires (from the row slabs read back; every entry of the system is then held once more, 40 bytes, by
 * exactly one device; kept until acx_mgpu_r1cs_destroy) -- callers that only verify or compute h(x) never pay for it. */
int acx_mgpu_qap_columns(acx_mgpu_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out, uint64_t* out_len);
/* `FFT.fft` / `FFT.interpolate` (galois-fft; src/QAP.hs:521-524) of ONE 2^log_n-point vector spread over the devices: host data
 * in natural order in and out, arguments of acx_ntt with batch = 1. */
int acx_mgpu_ntt(acx_mgpu* mg, uint32_t log_n, int inverse, const acx_fr* shift, const acx_fr* in, acx_fr* out);

/* The same with the witness already resident (sharded systems only): upload once, verify / compute h(x) many times; h(x)
 * stays on the devices (COLS ownership) until it is fetched.  What bench.py times: inputs in HBM when the clock starts. */
int acx_mgpu_witness_upload(acx_mgpu_r1cs* r, const acx_fr* witness);
int acx_mgpu_r1cs_verify_resident(acx_mgpu_r1cs* r, int* ok, uint64_t* n_bad, uint64_t* first_bad);
int acx_mgpu_qap_h_resident(acx_mgpu_r1cs* r, const acx_fr* delta, int* ok);
int acx_mgpu_qap_h_fetch(acx_mgpu_r1cs* r, acx_fr* out_h, uint64_t* h_len);
/* Throughput form (`all (verifyAssignment qap) inputs` with the inputs resident): enqueue adds the violated-row count of one
 * check of the resident witness to result slot `slot` (< 16) on every device and returns without waiting; verdicts combines
 * the slots [slot0, slot0 + count) with ONE all-reduce, waits, returns the counts and clears the slots. */
int acx_mgpu_r1cs_verify_enqueue(acx_mgpu_r1cs* r, uint32_t slot);
int acx_mgpu_r1cs_verdicts(acx_mgpu_r1cs* r, uint32_t slot0, uint32_t count, uint64_t* n_bad);

#ifdef __cplusplus
}
#endif
#endif /* ACX_H */
