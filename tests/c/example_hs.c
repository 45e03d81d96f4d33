/* tests/c/example_hs.c -- the reference's Example.hs (/root/reference/Example.hs:10-38, README.tex.md:220-262)
 * driven through include/acx.h from plain C: no Python, no torch, nothing but libacx.so.
 *
 *   program = (i0 * i1) * (i0 + i2)        built by execCircuitBuilder: the shared counter gives the wires
 *             InputWire 0,1,2, IntermediateWire 3,4 and no OutputWire (src/Circuit/Expr.hs:201-217)
 *   roots   = evalFresh (generateRoots ((+1) <$> fresh) program) = [[1],[2]]
 *   qap     = arithCircuitToQAPFFT getRootOfUnity roots program
 *   inputs  = {0:7, 1:5, 2:4};  assignment = generateAssignment program inputs   (wires 3,4 = 35, 385)
 *   verifyAssignment qap assignment  ->  "Valid assignment"
 * plus what the reference's tests never pin but the maths fixes: h = [42], the
 * interpolated column of wire i0 in A is [1/2, 1/2], and a corrupted assignment is "Invalid".
 * exit: 0 ok, 77 no usable GPU (there is no CPU fallback), 1 wrong result. */
#include <stdio.h>
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
static int fr_is_u64(const acx_fr* f, uint64_t v) {
    acx_fr g = fr_u64(v);
    return memcmp(f, &g, sizeof g) == 0;
}

int main(void) {
    /* ---- marshal [Mul (Var I0) (Var I1) M3, Mul (Var M3) (Add (Var I0) (Var I2)) M4] (src/Circuit/Arithmetic.hs:44-59) */
    const uint8_t kind[2] = {ACX_GATE_MUL, ACX_GATE_MUL};
    /* gate 0: left = Var aff[0], right = Var aff[1]; gate 1: left = Var aff[2], right = Add (Var aff[3]) (Var aff[4]) */
    const uint8_t ops[6] = {ACX_AFF_VAR, ACX_AFF_VAR, ACX_AFF_VAR, ACX_AFF_ADD, ACX_AFF_VAR, ACX_AFF_VAR};
    const uint32_t args[6] = {0, 1, 2, 0, 3, 4};
    const uint64_t tok_ofs[5] = {0, 1, 2, 3, 6};
    const acx_wire aff[5] = {{ACX_WIRE_INPUT, 0}, {ACX_WIRE_INPUT, 1}, {ACX_WIRE_INTERMEDIATE, 3},
                             {ACX_WIRE_INPUT, 0}, {ACX_WIRE_INPUT, 2}};
    const uint64_t wire_ofs[3] = {0, 1, 2};
    const acx_wire outs[2] = {{ACX_WIRE_INTERMEDIATE, 3}, {ACX_WIRE_INTERMEDIATE, 4}};
    acx_gate_list gl;
    memset(&gl, 0, sizeof gl);
    gl.n_gates = 2; gl.kind = kind; gl.tok_ofs = tok_ofs; gl.tok_op = ops; gl.tok_arg = args;
    gl.scalars = NULL; gl.n_scalars = 0; gl.aff_wires = aff; gl.n_aff_wires = 5; gl.wire_ofs = wire_ofs; gl.wires = outs;

    acx_circuit* circ = NULL;
    CHECK(acx_circuit_create(ACX_FIELD_BN254_FR, &gl, &circ));
    uint64_t n_rows, m, n_in, n_mid, n_out;
    CHECK(acx_circuit_dims(circ, &n_rows, &m, &n_in, &n_mid, &n_out));
    if (n_rows != 2 || n_in != 3 || n_mid != 5 || n_out != 0 || m != 9) { fprintf(stderr, "unexpected dims\n"); return 1; }
    int valid = 0;
    CHECK(acx_circuit_valid(circ, &valid));                      /* validArithCircuit */
    const uint32_t root_counts[2] = {1, 1};
    CHECK(acx_circuit_check_root_counts(circ, root_counts, 2));  /* roots :: [[Fr]] = [[1],[2]] */
    const acx_fr roots[2] = {fr_u64(1), fr_u64(2)};

    /* ---- generateAssignment program inputs (host code, no GPU) */
    const acx_fr inputs[3] = {fr_u64(7), fr_u64(5), fr_u64(4)};
    acx_fr w[9];
    uint8_t assigned[9];
    CHECK(acx_circuit_eval(circ, inputs, NULL, 3, w, assigned));
    /* flat numbering (qapSetToMap): 0 const, 1..3 inputs, 4..8 intermediates 0..4 */
    if (!fr_is_u64(&w[0], 1) || !fr_is_u64(&w[7], 35) || !fr_is_u64(&w[8], 385) || assigned[4] || !assigned[8]) {
        fprintf(stderr, "generateAssignment mismatch\n"); return 1;
    }

    /* ---- the GPU part: context, GenQAP, verifyAssignment, verificationWitness, one interpolated column */
    acx_ctx* ctx = NULL;
    CHECK(acx_ctx_create(ACX_FIELD_BN254_FR, 0, &ctx));
    acx_r1cs* r = NULL;
    CHECK(acx_circuit_to_r1cs(ctx, circ, roots, 2, &r));          /* arithCircuitToGenQAP roots program */
    int ok = 0;
    uint64_t n_bad = 0, first_bad = 0;
    CHECK(acx_r1cs_verify(r, w, &ok, &n_bad, &first_bad));        /* verifyAssignment qap assignment */
    puts(ok ? "Valid assignment" : "Invalid assignment");
    if (!ok) return 1;

    acx_fr h[3];
    uint64_t h_len = 0;
    int h_ok = 0;
    CHECK(acx_qap_h(r, w, NULL, h, &h_len, &h_ok));               /* verificationWitness: Just [42] */
    if (!h_ok || h_len != 1 || !fr_is_u64(&h[0], 42)) { fprintf(stderr, "h(x) mismatch\n"); return 1; }

    acx_fr col[2];
    uint64_t col_len = 0;
    CHECK(acx_qap_columns(r, ACX_MATRIX_A, 1, 1, col, &col_len)); /* createPolynomialsFFT: wire i0 of qapInputsLeft */
    /* evaluations [1,0] on {1,-1} interpolate to 1/2 + x/2, 1/2 = (r+1)/2 */
    static const uint8_t half[32] = {0x01,0x00,0x00,0xf8,0xc9,0xfa,0xf0,0xa1,0x48,0xb8,0xdc,0x3c,0x24,0xf4,0x19,0x94,
                                     0x2e,0xac,0xc0,0x40,0xdb,0x22,0x28,0xdc,0x14,0xd0,0x98,0x70,0x39,0x27,0x32,0x18};
    if (col_len != 2 || memcmp(col[0].b, half, 32) != 0 || memcmp(col[1].b, half, 32) != 0) { fprintf(stderr, "column mismatch\n"); return 1; }

    w[8] = fr_u64(386);                                            /* corrupt the product wire */
    CHECK(acx_r1cs_verify(r, w, &ok, &n_bad, &first_bad));
    if (ok || n_bad != 1 || first_bad != 1) { fprintf(stderr, "corrupted assignment accepted\n"); return 1; }
    puts("Invalid assignment (corrupted copy): row 1 violated");

    acx_r1cs_destroy(r);
    acx_ctx_destroy(ctx);
    acx_circuit_destroy(circ);
    return 0;
}
