"""acx -- MI355X-native R1CS / QAP evaluation engine (drop-in for the hot path of
sdiehl/arithmetic-circuits' `QAP` module).  The directory is named `arithmetic-circuits_amd`;
import it with importlib (`importlib.import_module("arithmetic-circuits_amd")`) or via the
`acx` alias that tests/conftest.py, bench.py and __graft_entry__.py register."""
from . import _lib
from ._lib import AcxError
from .engine import Batch, Circuit, Context, MgR1CS, MultiGpu, Naive, R1CS, fr_to_ints, ints_to_fr
from .circuit import (Add, ArithCircuit, ConstGate, Equal, InputWire, IntermediateWire, Mul, OutputWire,
                      ScalarMul, Split, Var, Wire, freshRoots, generateRoots, unsplit)
from .qap import (GenQAP, NaiveQAP, QAP, QapSet, arithCircuitToGenQAP, arithCircuitToQAP, arithCircuitToQAPFFT,
                  createPolynomials, createPolynomialsFFT,
                  gateToQAP, generateAssignment, generateAssignmentGate, initialQapSet, lookupAtWire,
                  qapSetToMap, verificationWitness, verificationWitnessZk, verifyAssignment, verifyAssignments)
from . import expr, json_io, parallel, synth  # noqa: E402  (host-side mirrors and utilities)
