"""T2: oracle (b) [oracle/acx_oracle.c, evaluation-domain algorithm = what the GPU runs] against
oracle (a) [oracle/ref_qap.py, the reference's literal polynomial algorithm].  Pure CPU.
Pins: "polynomial-division Bool == all-residuals-zero Bool" incl. corrupted witnesses; the
quotient h(x) and the per-wire interpolants are bit-equal between the two formulations."""
import random

import numpy as np
import pytest

from oracle import ref_qap as R
from oracle.c_oracle import COracle, ints_to_limbs, limbs_to_ints
from tests import helpers as H


def _case(rnd, field, size, split_bits=8):
    p = field.p
    num_vars = rnd.randrange(1, 6)
    gates = H.arb_arith_circuit(rnd, p, num_vars, size, split_bits=split_bits)
    roots = R.fresh_roots(gates, 1)
    gen = R.arith_circuit_to_gen_qap(roots, gates, p)
    dims = H.circuit_dims(gates)
    n, m, mats = H.gen_qap_to_csr(gen, dims, p)
    return gates, num_vars, gen, dims, n, m, mats


@pytest.mark.parametrize("fname", ["bn254", "bls12_381"])
@pytest.mark.parametrize("seed", range(4))
def test_residual_bool_equals_division_bool(fname, seed):
    field = R.BN254 if fname == "bn254" else R.BLS12_381
    p = field.p
    orc = COracle(fname)
    rnd = random.Random(4000 + seed)
    gates, num_vars, gen, dims, n, m, mats = _case(rnd, field, rnd.randrange(2, 10))
    qap = R.create_polynomials_fft(field.root_of_unity, gen, p)
    log_n = max(0, (n - 1).bit_length())
    for trial in range(4):
        a = R.generate_assignment(gates, H.arb_input_vector(rnd, p, num_vars), p)
        w = H.qapset_to_flat(a, dims, p)
        if trial >= 2:  # corrupt one wire (also in the literal QapSet)
            k = rnd.randrange(1, m)
            w[k] = (w[k] + 1 + rnd.randrange(p - 1)) % p
            base = [1, 1 + dims[0], 1 + dims[0] + dims[1]]
            kind = 2 if k >= base[2] else (1 if k >= base[1] else 0)
            (a.inputs, a.intermediates, a.outputs)[kind][k - base[kind]] = w[k]
        want_h = R.verification_witness(qap, a, p)
        res, nbad, first = orc.r1cs_residuals(n, m, *mats, ints_to_limbs(w))
        assert (nbad == 0) == (want_h is not None)
        res_int = limbs_to_ints(res)
        assert nbad == sum(1 for r in res_int if r) and (first == min(i for i, r in enumerate(res_int) if r) if nbad else first == 2**64 - 1)
        h, ok = orc.qap_h(n, m, log_n, *mats, ints_to_limbs(w))
        assert ok == (want_h is not None)
        if ok:
            assert R.to_poly(limbs_to_ints(h), p) == want_h
        # zero-knowledge variant
        d = [rnd.randrange(p) for _ in range(3)]
        want_zk = R.verification_witness_zk(d[0], d[1], d[2], qap, a, p)
        hz, okz = orc.qap_h(n, m, log_n, *mats, ints_to_limbs(w), delta=d)
        assert okz == (want_zk is not None)
        if okz:
            assert R.to_poly(limbs_to_ints(hz), p) == want_zk


@pytest.mark.parametrize("seed", range(3))
def test_columns_equal_fft_interpolate(seed):
    field, p = R.BN254, R.BN254.p
    orc = COracle("bn254")
    rnd = random.Random(5000 + seed)
    gates, num_vars, gen, dims, n, m, mats = _case(rnd, field, rnd.randrange(2, 8))
    qap = R.create_polynomials_fft(field.root_of_unity, gen, p)
    log_n = max(0, (n - 1).bit_length())
    for k, qs in enumerate((qap.left, qap.right, qap.out)):
        cols = orc.qap_columns(n, log_n, mats[k], 0, m)
        got = [R.to_poly(limbs_to_ints(cols[w]), p) for w in range(m)]
        assert got[0] == qs.constant
        for kind, part in enumerate((qs.inputs, qs.intermediates, qs.outputs)):
            for idx, poly in part.items():
                assert got[H.flat_index(dims, R.Wire(kind, idx))] == poly
        # wires the circuit never mentions in this matrix interpolate to the zero polynomial
        mentioned = {0} | {H.flat_index(dims, R.Wire(kd, i)) for kd, part in enumerate((qs.inputs, qs.intermediates, qs.outputs)) for i in part}
        for w in range(m):
            if w not in mentioned:
                assert got[w] == []


def test_split_256_bits_evaluation_domain():
    """The reference generator's 256-bit Split (test/Test/Circuit/Arithmetic.hs:123) through the
    evaluation-domain checker (the literal O(n^2) algorithm is too slow at n = 259 rows x m)."""
    p = R.BN254.p
    orc = COracle("bn254")
    rnd = random.Random(77)
    outs = [R.IntermediateWire(1 + j) for j in range(256)]
    gates = [R.Mul(R.Var(R.InputWire(0)), R.Var(R.InputWire(1)), R.IntermediateWire(0)),
             R.Split(R.IntermediateWire(0), outs),
             R.Mul(R.ConstGate(1), R.unsplit(outs), R.OutputWire(0))]
    roots = R.fresh_roots(gates, 1)
    rows = []
    for rs, g in zip(roots, gates):
        rows += R.gate_to_gen_qap(rs, g, p)
    gen = R.create_map_gen_qap(rows)   # sparse form is enough for CSR conversion
    dims = H.circuit_dims(gates)
    n, m, mats = H.gen_qap_to_csr(gen, dims, p)
    assert n == 1 + 257 + 1
    a = R.generate_assignment(gates, {0: rnd.randrange(p), 1: rnd.randrange(p)}, p)
    assert a.outputs[0] == a.intermediates[0]
    w = H.qapset_to_flat(a, dims, p)
    _, nbad, _ = orc.r1cs_residuals(n, m, *mats, ints_to_limbs(w))
    assert nbad == 0
    w[5] = (w[5] + 1) % p   # break one bit wire: bit*(1-bit) or the recomposition must fail
    _, nbad, first = orc.r1cs_residuals(n, m, *mats, ints_to_limbs(w))
    assert nbad >= 1


def test_ntt_threads_equal_single():
    orc = COracle("bn254")
    rnd = random.Random(9)
    xs = ints_to_limbs([rnd.randrange(orc.p) for _ in range(1 << 13)])
    a = orc.ntt(xs, 13, nthreads=1)
    b = orc.ntt(xs, 13, nthreads=4)
    assert np.array_equal(a, b)
    assert np.array_equal(orc.ntt(a, 13, inverse=True, nthreads=3), xs)


@pytest.mark.parametrize("fname", ["bn254", "bls12_381"])
@pytest.mark.parametrize("seed", range(3))
def test_reference_algorithm_mode_equals_literal_oracle(fname, seed):
    """orc_ref_verify -- the C restatement of the reference's OWN polynomial-domain algorithm (dense per-wire polynomials,
    scalar x polynomial sums, dense product, long division by x^N - 1: /root/reference/src/QAP.hs:276-327), which bench.py
    times on configs[0] as `cpu_baseline.reference_algorithm` -- against the literal big-int oracle: same Bool, same
    quotient, valid and corrupted assignments; and against the evaluation-domain h(x) of the same file."""
    field = R.BN254 if fname == "bn254" else R.BLS12_381
    p = field.p
    orc = COracle(fname)
    rnd = random.Random(7100 + seed)
    gates, num_vars, gen, dims, n, m, mats = _case(rnd, field, rnd.randrange(3, 10))
    qap = R.create_polynomials_fft(field.root_of_unity, gen, p)
    log_n = max(0, (n - 1).bit_length())
    cols = np.stack([orc.qap_columns(n, log_n, mats[k], 0, m) for k in range(3)])
    for trial in range(4):
        a = R.generate_assignment(gates, H.arb_input_vector(rnd, p, num_vars), p)
        w = H.qapset_to_flat(a, dims, p)
        if trial >= 2:
            k = rnd.randrange(1, m)
            w[k] = (w[k] + 1 + rnd.randrange(p - 1)) % p
            base = [1, 1 + dims[0], 1 + dims[0] + dims[1]]
            kind = 2 if k >= base[2] else (1 if k >= base[1] else 0)
            (a.inputs, a.intermediates, a.outputs)[kind][k - base[kind]] = w[k]
        want_h = R.verification_witness(qap, a, p)
        q, ok = orc.ref_verify(m, log_n, cols, ints_to_limbs(w))
        assert ok == (want_h is not None)
        if ok:
            assert R.to_poly(limbs_to_ints(q), p) == want_h
            h, ok2 = orc.qap_h(n, m, log_n, *mats, ints_to_limbs(w))
            assert ok2 and np.array_equal(h[: q.shape[0]], q) and not h[q.shape[0]:].any()
