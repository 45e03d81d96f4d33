"""acx_mgpu_*: the N-GPU path behind the C ABI (include/acx.h; one process, the library shards the rows and issues the
collectives).  Every result must be bit-identical to the single-GPU entry point and to the C oracle:

  * devices = [0]                     one shard through the REAL RCCL calls (ncclCommInitAll, ncclAllToAll, ncclAllReduce)
  * devices = [0, 0] / [0] * 4 / * 8  several shards on the one GPU of the box, exchanged by device copies: the W > 1
                                      ownership, exchange and gather logic
`verifyAssignment` (/root/reference/src/QAP.hs:276-282), `verificationWitness[Zk]` (src/QAP.hs:292-327) and the
transform behind `createPolynomialsFFT` (src/QAP.hs:512-525) each in its one-call shape."""
import numpy as np
import pytest

from oracle import ref_qap as R

pytestmark = pytest.mark.gpu
U64_MAX = 2**64 - 1


def _mg(acx, request, field, devices):
    from tests.conftest import gpu_required
    import torch
    if not torch.cuda.is_available():
        if gpu_required():
            pytest.fail("no GPU visible and the run requires one")
        pytest.skip("no GPU visible")
    mg = acx.MultiGpu(field, devices)
    request.addfinalizer(mg.close)
    return mg


def _orc(request, field):
    return request.getfixturevalue("c_oracle_bn254" if field == "bn254" else "c_oracle_bls")


DEVICE_LISTS = [[0], [0, 0], [0, 0, 0, 0], [0] * 8]


@pytest.mark.parametrize("devices", DEVICE_LISTS, ids=lambda d: f"W{len(d)}")
@pytest.mark.parametrize("field", ["bn254", "bls12_381"])
def test_mgpu_verify_and_h_equal_single_gpu_and_oracle(acx, request, field, devices):
    synth = acx.synth
    mg = _mg(acx, request, field, devices)
    assert mg.transport == ("rccl" if len(devices) == 1 else "peer-copy")
    mg.set_shard_threshold(10)
    orc = _orc(request, field)
    n = (1 << 13) - 37                                        # padding rows in every shard
    s = synth.mulgraph(n, n_in=64, window=512, seed=0xA11CE + len(devices), field=field)
    mats, w = s.rows(), s.witness()
    mr = mg.from_circuit(s.circuit)
    assert (mr.n, mr.m, mr.log_n, mr.n_shards) == (n, s.circuit.m, 13, len(devices))
    assert mr.verify(w) == (True, 0, U64_MAX)
    h, ok = mr.qap_h(w)
    want_h, want_ok = orc.qap_h(n, mr.m, 13, *mats, w)
    assert ok and want_ok
    assert np.array_equal(h, want_h[: h.shape[0]]) and not want_h[h.shape[0]:].any()
    # a corrupted witness: count and smallest violated GLOBAL row as the oracle reports them, h = Nothing
    bad = w.copy()
    bad[s.circuit.m // 2, 0] ^= np.uint64(1)
    _, nbad, first = orc.r1cs_residuals(n, mr.m, *mats, bad)
    assert nbad > 0
    assert mr.verify(bad) == (False, nbad, first)
    assert mr.verify(bad, want_first=False) == (False, nbad, U64_MAX)
    assert mr.qap_h(bad) == (None, False)
    # the zero-knowledge quotient (src/QAP.hs:300-327)
    p = R.BN254.p if field == "bn254" else R.BLS12_381.p
    delta = [3, p - 5, 1234567]
    hz, okz = mr.qap_h(w, delta)
    ctx1 = request.getfixturevalue("ctx_bn254" if field == "bn254" else "ctx_bls")
    r1 = s.circuit.to_r1cs(ctx1)
    hz1, okz1 = r1.qap_h(w, delta)
    assert okz and okz1 and np.array_equal(hz, hz1)
    # resident form: upload once, verify / h many times
    mr.upload_witness(w)
    assert mr.verify_resident() == (True, 0, U64_MAX)
    assert mr.qap_h_resident()
    assert np.array_equal(mr.qap_h_fetch(), h)
    # throughput form: checks enqueued into result slots, ONE collective for a range of slots
    for k in range(4):
        mr.verify_enqueue(k)
    assert mr.verdicts(0, 4).tolist() == [0, 0, 0, 0]
    mr.upload_witness(bad)
    mr.verify_enqueue(2)
    mr.verify_enqueue(2)
    mr.verify_enqueue(5)
    assert mr.verdicts(2, 4).tolist() == [2 * nbad, 0, 0, nbad]
    assert mr.verdicts(0, 16).tolist() == [0] * 16              # read slots are cleared
    r1.close()
    mr.close()
    # many assignments in one call (more than one ring of 16, valid and corrupted interleaved)
    many = np.stack([bad if k % 3 == 1 else w for k in range(37)])
    # verification-only load: no block-cyclic copy, h(x) is refused, the verdict is the same
    mv = mg.from_circuit(s.circuit, verify_only=True)
    assert mv.verify(bad) == (False, nbad, first) and mv.verify(w) == (True, 0, U64_MAX)
    oks, nbads = mv.verify_many(many)
    assert oks.tolist() == [k % 3 != 1 for k in range(37)] and nbads.tolist() == [nbad if k % 3 == 1 else 0 for k in range(37)]
    assert mv.verify_many(many[:0])[0].shape == (0,)
    many[20, 7] = np.array([U64_MAX] * 4, dtype=np.uint64)
    with pytest.raises(acx.AcxError) as e:
        mv.verify_many(many)
    assert e.value.status == acx._lib.STATUS["NONCANONICAL"]
    assert mv.verify(w) == (True, 0, U64_MAX)                   # the handle is usable after the failed call
    with pytest.raises(acx.AcxError) as e:
        mv.qap_h(w)
    assert e.value.status == acx._lib.STATUS["UNSUPPORTED"]
    mv.close()


@pytest.mark.parametrize("devices", DEVICE_LISTS, ids=lambda d: f"W{len(d)}")
def test_mgpu_ntt_matches_single_gpu(acx, request, devices):
    """One vector spread over the shards (natural order in and out): forward, inverse, coset, odd and even sizes."""
    synth = acx.synth
    mg = _mg(acx, request, "bn254", devices)
    ctx1 = request.getfixturevalue("ctx_bn254")
    for log_n in (10, 13, 16):
        if 2 * len(devices) > (1 << (log_n // 2)):
            continue
        x = synth.random_fr(1 << log_n, 77 + log_n, 1)
        for inverse in (False, True):
            for shift in (None, 5):
                got = mg.ntt(x, log_n, inverse=inverse, shift=shift)
                want = ctx1.ntt(x, log_n, inverse=inverse, shift=shift)
                assert np.array_equal(got, want), (log_n, inverse, shift)
    # a non-canonical element is an error, as at every host edge
    x = synth.random_fr(1 << 10, 3, 1)
    x[5] = np.array([U64_MAX] * 4, dtype=np.uint64)
    with pytest.raises(acx.AcxError) as e:
        mg.ntt(x, 10)
    assert e.value.status == acx._lib.STATUS["NONCANONICAL"]


def test_mgpu_small_system_stays_whole(acx, request):
    """Below the shard threshold the system lives on the first device and every call is the single-GPU one (the
    reference's own test sizes)."""
    import random
    from tests import helpers as H
    mg = _mg(acx, request, "bn254", [0, 0])
    p = R.BN254.p
    rnd = random.Random(99)
    gates = H.arb_arith_circuit(rnd, p, 3, 20)
    host = H.to_acx_circuit(acx, gates).marshal("bn254")
    mr = mg.from_circuit(host)
    assert mr.n_shards == 1
    inputs = acx.ints_to_fr([rnd.randrange(p) for _ in range(3)])
    w, _ = host.eval(inputs)
    assert mr.verify(w)[0]
    mr.close()


def test_mgpu_sharded_gate_mix_long_rows(acx, request):
    """A circuit in the reference's gate mix (Mul : Equal : Split, test/Test/Circuit/Arithmetic.hs:136; the 257-entry rows of
    its Split gates take the CSR kernel) sharded over four shards: verdict, first violated row and h(x) against the oracle."""
    import random
    from tests import helpers as H
    mg = _mg(acx, request, "bn254", [0, 0, 0, 0])
    mg.set_shard_threshold(10)
    orc = _orc(request, "bn254")
    p = R.BN254.p
    rnd = random.Random(4242)
    gates = H.arb_arith_circuit(rnd, p, 4, 260, dist=(50, 10, 2))
    host = H.to_acx_circuit(acx, gates).marshal("bn254")
    mats = host.rows()
    inputs = acx.ints_to_fr([rnd.randrange(p) for _ in range(4)])
    w, _ = host.eval(inputs)
    mr = mg.from_circuit(host)
    assert mr.n_shards == 4
    assert mr.verify(w) == (True, 0, U64_MAX)
    h, ok = mr.qap_h(w)
    want_h, _ = orc.qap_h(mr.n, mr.m, mr.log_n, *mats, w)
    assert ok and np.array_equal(h, want_h[: h.shape[0]]) and not want_h[h.shape[0]:].any()
    bad = w.copy()
    bad[1 + 4 + 7, 0] ^= np.uint64(2)
    _, nbad, first = orc.r1cs_residuals(mr.n, mr.m, *mats, bad)
    assert mr.verify(bad) == (False, nbad, first)
    mr.close()


def test_mgpu_argument_errors(acx, request):
    import ctypes as C
    lib = acx._lib.load()
    h = C.c_void_p()
    ids3 = (C.c_int * 3)(0, 0, 0)
    assert lib.acx_mgpu_create(0, ids3, 3, C.byref(h)) == acx._lib.STATUS["INVALID_ARG"]        # not a power of two
    assert lib.acx_mgpu_create(0, None, 1, C.byref(h)) == acx._lib.STATUS["INVALID_ARG"]
    bad = (C.c_int * 1)(4096)
    assert lib.acx_mgpu_create(0, bad, 1, C.byref(h)) == acx._lib.STATUS["NO_DEVICE"]
    mg = _mg(acx, request, "bn254", [0])
    s = acx.synth.mulgraph(1 << 10, n_in=16, window=64)
    mr = mg.from_circuit(s.circuit)                       # 2^10 < threshold 2^14: whole on shard 0
    assert mr.n_shards == 1
    with pytest.raises(acx.AcxError) as e:
        mr.upload_witness(s.witness())
    assert e.value.status == acx._lib.STATUS["UNSUPPORTED"]
    w = s.witness()
    assert mr.verify(w) == (True, 0, U64_MAX)
    h1, ok = mr.qap_h(w)
    assert ok and h1 is not None
    w[3] = np.array([U64_MAX] * 4, dtype=np.uint64)
    mg.set_shard_threshold(10)
    mr2 = mg.from_circuit(s.circuit)
    assert mr2.n_shards == 1 and mr2.log_n == 10
    with pytest.raises(acx.AcxError) as e:
        mr2.verify(w)
    assert e.value.status == acx._lib.STATUS["NONCANONICAL"]
    mr.close()
    mr2.close()


@pytest.mark.parametrize("devices", ["0", "0,0,0,0"])
def test_plain_c_host_drives_several_shards_without_rccl_or_hip_in_the_host(acx, request, tmp_path, devices):
    """examples/mgpu_host.c: a plain-C program (gcc, include/acx.h only: no RCCL, HIP or Python in the host) runs
    verifyAssignment and verificationWitness over the shards of an acx_mgpu handle and compares with the single-GPU
    entry points; with "0" the exchange and the verdict go through real RCCL calls inside libacx."""
    import os, subprocess
    _mg(acx, request, "bn254", [0])
    root = os.path.join(os.path.dirname(__file__), "..")
    libdir = os.path.abspath(os.path.join(root, "arithmetic-circuits_amd"))
    exe = str(tmp_path / "mgpu_host")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "mgpu_host.c"),
                    "-L", libdir, "-lacx", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe],
                   check=True, capture_output=True, text=True)
    out = subprocess.run([exe, devices, "13"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Valid assignment" in out.stdout, (out.stdout, out.stderr[-2000:])
    assert ("RCCL" if devices == "0" else "peer copies") in out.stdout


def test_mgpu_configs3_2_24_constraints_through_rccl(acx, request):
    """BASELINE.json configs[3]'s constraint system -- 2^24 constraints = 256 block-diagonal 2^16-constraint mulgraph
    systems, 4096 x 4096 four-step transforms -- through the N-GPU entry points on the one GPU of the box, twice: ONE shard
    (the six all-to-alls and the verdict all-reduce issued as real RCCL calls inside libacx), and EIGHT shards on the one
    device -- the 8-way split of the config itself: eight issuing threads, 2^21 block-cyclic rows per shard in runs of 512,
    8 x 8 exchange blocks of 8 MiB per transform over peer copies, the witness replicated from shard 0.  h(x) -- all 2^24
    coefficients -- and the verdicts against the C oracle, both times."""
    import os
    synth = acx.synth
    orc = _orc(request, "bn254")
    bs = synth.BlockSystem(synth.mulgraph(1 << 16), 256)
    N = 1 << 24
    assert bs.n == N
    mats, w = bs.full_rows(), bs.witness()
    threads = os.cpu_count() or 1
    want_h, want_ok = orc.qap_h(bs.n, bs.m, 24, *mats, w, nthreads=threads)
    assert want_ok
    bad = w.copy()
    bad[bs.wire(200, 77), 0] ^= np.uint64(1)
    _, nbad, first = orc.r1cs_residuals(bs.n, bs.m, *mats, bad, want_residuals=False, nthreads=threads)
    assert nbad > 0
    for devices, transport in (([0], "rccl"), ([0] * 8, "peer-copy")):
        mg = _mg(acx, request, "bn254", devices)
        mr = mg.load(bs.n, bs.m, *mats)
        assert (mr.log_n, mr.n_shards) == (24, len(devices)) and mg.transport == transport
        assert mr.verify(w) == (True, 0, U64_MAX)
        h, ok = mr.qap_h(w)
        assert ok
        assert np.array_equal(h, want_h[: h.shape[0]]) and not want_h[h.shape[0]:].any()
        del h
        assert mr.verify(bad) == (False, nbad, first)
        assert mr.qap_h(bad) == (None, False)
        mr.close()
        mg.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0, 0]], ids=lambda d: f"W{len(d)}")
def test_mgpu_calls_from_several_threads(acx, request, devices):
    """Haskell `safe` foreign calls arrive from arbitrary OS threads: one handle, four threads, each
    alternating verifyAssignment on a good and a bad witness and the quotient; every answer equals the serial one (the
    handle serialises the calls: its collectives must be issued in one order on every device)."""
    import threading
    synth = acx.synth
    mg = _mg(acx, request, "bn254", devices)
    mg.set_shard_threshold(10)
    n = 1 << 12
    s = synth.mulgraph(n, n_in=32, window=256, seed=0x7EAD, field="bn254")
    w = s.witness()
    mr = mg.from_circuit(s.circuit)
    bads = []
    for k in range(4):
        b = w.copy()
        b[1 + 32 + 97 * (k + 1), 0] ^= np.uint64(1 << k)
        bads.append(b)
    want_bad = [mr.verify(b) for b in bads]
    want_h, ok = mr.qap_h(w)
    assert ok and all(not v[0] and v[1] > 0 for v in want_bad)
    errors = []

    def worker(k):
        try:
            for it in range(6):
                assert mr.verify(w) == (True, 0, U64_MAX)
                assert mr.verify(bads[k]) == want_bad[k]
                if it % 2 == 0:
                    h, okk = mr.qap_h(w)
                    assert okk and np.array_equal(h, want_h)
                else:
                    assert mr.qap_h(bads[k]) == (None, False)
        except Exception as e:                       # assertions inside threads are otherwise lost
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("devices", [[0], [0, 0, 0, 0], [0] * 8], ids=lambda d: f"W{len(d)}")
@pytest.mark.parametrize("field", ["bn254", "bls12_381"])
def test_mgpu_qap_columns_shared_out_by_wire(acx, request, field, devices):
    """createPolynomialsFFT (/root/reference/src/QAP.hs:512-525) behind the multi-GPU handle: wires shared out over the
    shards (block-cyclic by wire), every shard interpolating its wires on the column view of those wires alone (built on
    the first call from the row slabs) -- bit-equal to the C oracle and to acx_qap_columns of one GPU, coefficients and stripped lengths, for
    ranges that divide evenly, ragged ones, fewer wires than shards, and the gate mix whose Split rows are long."""
    synth = acx.synth
    mg = _mg(acx, request, field, devices)
    mg.set_shard_threshold(10)
    orc = _orc(request, field)
    ctx1 = request.getfixturevalue("ctx_bn254" if field == "bn254" else "ctx_bls")
    n = (1 << 12) - 19
    s = synth.mulgraph(n, n_in=48, window=300, seed=0xC01 + len(devices), field=field)
    mats = s.rows()
    mr = mg.from_circuit(s.circuit)
    r1 = s.circuit.to_r1cs(ctx1)
    assert mr.n_shards == len(devices)
    # (wires are owned block-cyclically, 64 per block: ranges inside one block, across several, ending on and off a block edge)
    for k, w0, cnt in ((0, 0, 64), (1, 17, 37), (2, mr.m - 5, 5), (0, 3, 3), (1, 200, 1), (0, 50, 300), (2, 63, 2), (1, 128, 512), (0, 0, min(mr.m, 1500))):
        cols, lens = mr.qap_columns(k, w0, cnt)
        want = orc.qap_columns(n, mr.log_n, mats[k], w0, cnt, nthreads=8)
        one, one_lens = r1.qap_columns(k, w0, cnt)
        assert np.array_equal(cols, want) and np.array_equal(cols, one)
        assert np.array_equal(lens, one_lens)
    # verification and the quotient still work after the copies exist (they are separate handles on the same contexts)
    w = s.witness()
    assert mr.verify(w) == (True, 0, U64_MAX)
    h, ok = mr.qap_h(w)
    h1, ok1 = r1.qap_h(w)
    assert ok and ok1 and np.array_equal(h, h1)
    with pytest.raises(acx.AcxError):
        mr.qap_columns(0, mr.m - 2, 3)
    with pytest.raises(acx.AcxError):
        mr.qap_columns(3, 0, 1)


@pytest.mark.parametrize("mode", ["copies", "pinned"])
def test_mgpu_witness_replication_modes(mode):
    """ACX_MGPU_WITNESS selects how the witness reaches the shards: the default is one host-to-device copy + a device-side
    broadcast; `copies` (one pageable copy per shard) and `pinned` (the same from memory the library page-locks for the call)
    must give the same results -- the verify / h(x) equality test of this file, run in a child process under each mode."""
    import os, subprocess, sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_mgpu.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
           "-k", "verify_and_h_equal_single_gpu_and_oracle and bn254"]
    out = subprocess.run(cmd, cwd=root, env=dict(os.environ, ACX_MGPU_WITNESS=mode), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " passed" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def test_mgpu_qap_columns_device_memory_is_one_system_not_W(acx, request):
    """Eight shards on one device: what acx_mgpu_qap_columns leaves on the device for its column views must stay below 1.5x
    ONE system (every entry is held once, 40 bytes, by the shard that owns its wire) -- the first version gave each of the
    eight shards a copy of the whole system.  hipMemGetInfo around a single-GPU load of the same system is the yardstick."""
    import gc
    import torch
    synth = acx.synth
    ctx1 = request.getfixturevalue("ctx_bn254")
    s = synth.mulgraph(1 << 18, seed=0xC015)
    mats = s.rows()

    def free_bytes():
        gc.collect()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    f0 = free_bytes()
    r1 = s.circuit.to_r1cs(ctx1)
    ctx1.sync()
    one_system = f0 - free_bytes()
    r1.close()
    assert one_system > 50e6
    mg = acx.MultiGpu("bn254", [0] * 8)
    mr = mg.from_circuit(s.circuit)
    assert mr.n_shards == 8
    mr.verify(s.witness())                       # everything verify needs exists before the measurement
    f1 = free_bytes()
    cols, lens = mr.qap_columns(0, 1000, 24)
    grown = f1 - free_bytes()
    orc = _orc(request, "bn254")
    assert np.array_equal(cols, orc.qap_columns(mr.n, mr.log_n, mats[0], 1000, 24, nthreads=8))
    assert grown <= 1.5 * one_system, (grown, one_system)
    f2 = free_bytes()
    mr.qap_columns(2, 5000, 200)                 # later calls add nothing that stays
    assert f2 - free_bytes() <= 0.05 * one_system
    mr.close()
    mg.close()


def test_mgpu_qap_columns_gate_mix_and_small_system(acx, request):
    """The reference's gate mix (long Split rows) through the per-wire path on two shards, and a system below the shard
    threshold (held whole on the first device: same call, same answer)."""
    import random
    from tests import helpers as H
    synth = acx.synth
    mg = _mg(acx, request, "bn254", [0, 0])
    mg.set_shard_threshold(10)
    orc = _orc(request, "bn254")
    rnd = random.Random(777)
    gates = H.arb_arith_circuit(rnd, R.BN254.p, 4, 200, dist=(50, 10, 2))
    host = H.to_acx_circuit(acx, gates).marshal("bn254")
    mats = host.rows()
    mr = mg.from_circuit(host)
    assert mr.n_shards == 2
    for k in range(3):
        cols, lens = mr.qap_columns(k, 1, 40)
        assert np.array_equal(cols, orc.qap_columns(mr.n, mr.log_n, mats[k], 1, 40, nthreads=8))
    small = synth.mulgraph(300, n_in=8, window=64, seed=5, field="bn254")
    ms = mg.from_circuit(small.circuit)
    assert ms.n_shards == 1
    cols, _ = ms.qap_columns(0, 0, 16)
    assert np.array_equal(cols, orc.qap_columns(300, ms.log_n, small.rows()[0], 0, 16, nthreads=4))


def test_mgpu_qap_h_outside_the_distributed_range_answers_from_one_device(acx, request):
    """A sharded system whose transform size the four-step form does not cover (here N = 2^11 on 32 shards: fewer than 2 W
    points per digit; in production N above 2^24): verifyAssignment runs on the slabs as always, verificationWitness still answers -- from one device, on its
    copy of the whole system -- bit-equal to the oracle; only a VERIFY_ONLY load refuses."""
    synth = acx.synth
    mg = _mg(acx, request, "bn254", [0] * 32)
    mg.set_shard_threshold(10)
    orc = _orc(request, "bn254")
    n = 1500
    s = synth.mulgraph(n, n_in=16, window=64, seed=0xF411, field="bn254")
    mats, w = s.rows(), s.witness()
    mr = mg.from_circuit(s.circuit)
    assert (mr.n_shards, mr.log_n) == (32, 11)
    assert mr.verify(w) == (True, 0, U64_MAX)
    delta = [11, 22, R.BN254.p - 33]
    for d in (None, delta):
        h, ok = mr.qap_h(w, d)
        want, want_ok = orc.qap_h(n, mr.m, 11, *mats, w, delta=d)
        assert ok and want_ok and np.array_equal(h, want[: h.shape[0]]) and not want[h.shape[0]:].any()
    bad = w.copy()
    bad[40, 0] ^= np.uint64(1)
    assert mr.qap_h(bad) == (None, False)
    cols, _ = mr.qap_columns(0, 0, 9)
    assert np.array_equal(cols, orc.qap_columns(n, 11, mats[0], 0, 9))
    with pytest.raises(acx.AcxError) as e:                       # the resident pipeline itself stays refused
        mr.upload_witness(w)
        mr.qap_h_resident()
    assert e.value.status == acx._lib.STATUS["UNSUPPORTED"]
    mv = mg.from_circuit(s.circuit, verify_only=True)
    with pytest.raises(acx.AcxError) as e:
        mv.qap_h(w)
    assert e.value.status == acx._lib.STATUS["UNSUPPORTED"]
