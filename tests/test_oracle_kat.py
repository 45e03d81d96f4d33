"""T1: the oracle (literal restatement, oracle/ref_qap.py) against every known-answer test the
reference's own test-suite holds for the hot path (SURVEY.md 8c list; reference file:line in
each test).  Pure CPU."""
import random

import pytest

from oracle import ref_qap as R
from tests import helpers as H

P = R.BN254.p
OMEGA = R.BN254.root_of_unity


def kat_circuit():
    """testArithCircuit, test/Test/QAP.hs:48-54."""
    return [R.Mul(R.Var(R.InputWire(0)), R.Var(R.InputWire(1)), R.IntermediateWire(0)),
            R.Mul(R.Var(R.InputWire(2)), R.Var(R.InputWire(3)), R.IntermediateWire(1)),
            R.Mul(R.Add(R.ConstGate(10), R.Var(R.IntermediateWire(0))), R.Var(R.IntermediateWire(1)), R.OutputWire(0))]


KAT_INPUTS = {0: 2, 1: 3, 2: 4, 3: 5}  # test/Test/QAP.hs:56-62


def test_unit_arithCircuitToQapCorrect():
    """test/Test/QAP.hs:68-75: naive roots 7,8,9 => True."""
    qap = R.arith_circuit_to_qap([[7], [8], [9]], kat_circuit(), P)
    assignment = R.generate_assignment(kat_circuit(), KAT_INPUTS, P)
    assert assignment.intermediates == {0: 6, 1: 20} and assignment.outputs == {0: 320}
    assert R.verify_assignment(qap, assignment, P)


def test_unit_arithCircuitToQapNoFalsePositive():
    """test/Test/QAP.hs:77-90: hand-written bad assignment => False."""
    qap = R.arith_circuit_to_qap([[7], [8], [9]], kat_circuit(), P)
    bad = R.QapSet(1, {0: 2, 1: 3, 2: 4, 3: 5}, {0: 7, 1: 20}, {0: 320})
    assert not R.verify_assignment(qap, bad, P)


def test_example_hs_valid_assignment():
    """Example.hs:10-38 / README.tex.md:208-262: (i0*i1)*(i0+i2), inputs 7,5,4 => "Valid assignment".
    The builder's shared counter makes the wires InputWire 0..2, IntermediateWire 3,4."""
    b = R.CircuitBuilder()
    i0, i1, i2 = ("var", b.input()), ("var", b.input()), ("var", b.input())
    out = b.ret(("mul", ("mul", i0, i1), ("add", i0, i2)))
    assert out == R.IntermediateWire(4) and len(b.gates) == 2
    roots = R.fresh_roots(b.gates, 1)
    assert roots == [[1], [2]]
    qap = R.arith_circuit_to_qap_fft(OMEGA, roots, b.gates, P)
    assignment = R.generate_assignment(b.gates, {0: 7, 1: 5, 2: 4}, P)
    assert assignment.intermediates == {3: 35, 4: 385}
    assert R.verify_assignment(qap, assignment, P)
    assert R.verification_witness(qap, assignment, P) == [42]


def test_bench_circuit_all_paths():
    """bench/Circuit.hs:17-36: the benched program through the three translations."""
    program = [R.Mul(R.Var(R.InputWire(0)), R.Var(R.InputWire(1)), R.IntermediateWire(0)),
               R.Mul(R.Var(R.IntermediateWire(0)), R.Add(R.Var(R.InputWire(0)), R.Var(R.InputWire(2))), R.OutputWire(0))]
    roots = R.fresh_roots(program, 0)
    assignment = R.generate_assignment(program, {0: 7, 1: 5, 2: 4}, P)
    assert assignment.outputs == {0: 385}
    assert R.verify_assignment(R.arith_circuit_to_qap_fft(OMEGA, roots, program, P), assignment, P)
    assert R.verify_assignment(R.arith_circuit_to_qap(roots, program, P), assignment, P)
    gen = R.arith_circuit_to_gen_qap(roots, program, P)
    assert sorted(gen.target) == [0, 1] and gen.left.inputs[0] == {0: 1, 1: 0}  # addMissingZeroes densifies


def test_unit_eqGate():
    """test/Test/Circuit/Arithmetic.hs:154-169: 0 -> 0; 1,2,3 -> 1."""
    circ = [R.Equal(R.InputWire(0), R.IntermediateWire(0), R.OutputWire(0))]
    for n, want in ((0, 0), (1, 1), (2, 1), (3, 1)):
        env = R.eval_arith_circuit(R.lookup_at_wire, R.update_at_wire, circ, R.initial_qap_set({0: n}), P)
        assert R.lookup_at_wire(R.OutputWire(0), env) == want


def test_unit_splitUnsplit():
    """test/Test/Circuit/Arithmetic.hs:171-182: 16-bit split/unsplit identity on ALL 0..65535."""
    nbits = 16
    mids = [R.IntermediateWire(i) for i in range(nbits)]
    circ = [R.Split(R.InputWire(0), mids), R.Mul(R.ConstGate(1), R.unsplit(mids), R.OutputWire(0))]
    for n in range(2 ** nbits):
        env = R.eval_arith_circuit(R.lookup_at_wire, R.update_at_wire, circ, R.initial_qap_set({0: n}), P)
        assert env.outputs[0] == n


@pytest.mark.parametrize("seed", range(6))
def test_prop_gateToQapCorrect(seed):
    """test/Test/QAP.hs:92-103: single random Mul or Equal gate, 10 inputs, FFT path."""
    rnd = random.Random(1000 + seed)
    num_vars = rnd.randrange(1, 8)
    if rnd.random() < 0.5:
        gate = R.Mul(H.arb_affine(rnd, P, num_vars, rnd.randrange(0, 4)),
                     H.arb_affine(rnd, P, num_vars, rnd.randrange(0, 4)), R.OutputWire(0))
        roots = [1]
    else:
        gate = R.Equal(R.InputWire(rnd.randrange(num_vars)), R.IntermediateWire(0), R.OutputWire(0))
        roots = [1, 2]
    qap = R.gate_to_qap(OMEGA, roots, gate, P)
    for _ in range(10):
        inp = H.arb_input_vector(rnd, P, num_vars)
        if rnd.random() < 0.2:
            inp[rnd.randrange(num_vars)] = 0      # exercise the Equal gate's zero branch
        assert R.verify_assignment(qap, R.generate_assignment_gate(gate, inp, P), P)


@pytest.mark.parametrize("seed", range(4))
def test_prop_arithCircuitToQAP_slow_and_fft(seed):
    """test/Test/Circuit/Arithmetic.hs:184-209: random circuits (gate mix 50:10:1), roots 1..n;
    validity, naive-Lagrange and FFT translations all verify every generated assignment.
    (Split width reduced from 256 to 8 bits to keep the O(n^2) literal algorithm in seconds; the
    256-bit case is covered by the evaluation-domain suites.)"""
    rnd = random.Random(2000 + seed)
    num_vars = rnd.randrange(1, 6)
    gates = H.arb_arith_circuit(rnd, P, num_vars, rnd.randrange(1, 9), split_bits=8)
    assert R.valid_arith_circuit(gates)
    roots = R.fresh_roots(gates, 1)
    qap_slow = R.arith_circuit_to_qap(roots, gates, P)
    qap_fft = R.create_polynomials_fft(OMEGA, R.arith_circuit_to_gen_qap(roots, gates, P), P)
    for _ in range(5):
        a = R.generate_assignment(gates, H.arb_input_vector(rnd, P, num_vars), P)
        assert R.verify_assignment(qap_slow, a, P)
        assert R.verify_assignment(qap_fft, a, P)


@pytest.mark.parametrize("seed", range(5))
def test_prop_affineCircuitToAffineMap(seed):
    """test/Test/Circuit/Affine.hs:55-62: flatten-then-dot == direct tree evaluation."""
    rnd = random.Random(3000 + seed)
    nv = rnd.randrange(1, 6)
    circ = H.arb_affine(rnd, P, nv, rnd.randrange(0, 6))
    inp = {R.InputWire(i): rnd.randrange(P) for i in range(nv)}
    direct = R.eval_affine_circuit(lambda w, vs: vs.get(w), inp, circ, P)
    assert R.eval_affine_map(R.affine_circuit_to_affine_map(circ, P), inp, P) == direct


def test_derived_coefficient_kats():
    """SURVEY.md Appendix A.6 (derived in the survey session; unpinned by the reference)."""
    half = (P + 1) // 2
    sgn = lambda xs: [x if x < P // 2 else x - P for x in xs]
    qap = R.arith_circuit_to_qap([[7], [8], [9]], kat_circuit(), P)
    assert sgn(qap.target) == [-504, 191, -24, 1]
    assert sgn(qap.left.constant) == [280, -75, 5]
    a = R.generate_assignment(kat_circuit(), KAT_INPUTS, P)
    assert len(R.verification_witness(qap, a, P)) == 2
    qf = R.arith_circuit_to_qap_fft(OMEGA, [[1], [2], [3]], kat_circuit(), P)
    h = R.verification_witness(qf, a, P)
    assert len(h) == 3 and h[0] == 48 and qf.target == [P - 1, 0, 0, 0, 1]
    assert R.fft_interpolate(OMEGA, [1, 0], P) == [half, half]
    assert R.fft_interpolate(OMEGA, [0, 1], P) == [half, half - 1]


def test_roots_of_unity_table():
    """pairing-1.0.0 getRootOfUnity (recollection): omega_k chain is self-consistent."""
    assert R.BN254.root_of_unity(1) == P - 1
    assert R.BN254.root_of_unity(2) == 21888242871839275217838484774961031246007050428528088939761107053157389710902
    assert R.BN254.root_of_unity(28) == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    for f in (R.BN254, R.BLS12_381):
        for k in range(1, 12):
            w = f.root_of_unity(k)
            assert pow(w, 1 << k, f.p) == 1 and pow(w, 1 << (k - 1), f.p) == f.p - 1
    with pytest.raises(ValueError):
        R.BN254.root_of_unity(29)
