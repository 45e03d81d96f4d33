"""Worker for test_gpu_parity.py::test_sharded_layer_two_ranks_on_one_device: the multi-GPU host layer
with the PRODUCT local kernels (libacx through HipOps / ShardedR1CS), two ranks sharing cuda:0 over gloo
(RCCL needs one GPU per rank; the sharding logic and the stream fencing do not).  gloo has no CUDA
all-to-all, so the collectives are tests/helpers_dist.py's host-staged ones; everything else is the code path of a
multi-GPU run."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.c_oracle import COracle                  # noqa: E402
from tests.helpers_dist import HostStagedCollectives  # noqa: E402

acx = importlib.import_module("arithmetic-circuits_amd")
par = importlib.import_module("arithmetic-circuits_amd.parallel")
synth = importlib.import_module("arithmetic-circuits_amd.synth")


def dev(ctx, arr):
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.dev_from_canonical(t.shape[0], t.data_ptr(), t.data_ptr())
    ctx.sync()
    return t


def canon(ctx, t):
    c = torch.empty_like(t)
    torch.cuda.synchronize()
    ctx.dev_to_canonical(t.shape[0], t.data_ptr(), c.data_ptr())
    ctx.sync()
    return c.cpu().numpy().view(np.uint64).reshape(-1, 4)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = acx.Context("bn254", 0)
    orc = COracle("bn254")

    # ---- rows sharded over the ranks, verdict through the collectives, HIP kernels underneath
    s = synth.mulgraph(1 << 14, n_in=64, window=512, seed=7)
    mats, w = s.rows(), s.witness()
    n, m = s.circuit.n_rows, s.circuit.m
    coll = HostStagedCollectives()
    sh = par.ShardedR1CS.from_slabs(mats, m, ctx=ctx, collectives=coll)
    assert sh.r1cs is not None and sh.rows.shape[0] < n
    assert sh.verify(w) == (True, 0, par.U64_MAX)
    bad = w.copy()
    for k in (77, 5000, 16000):
        bad[k, 0] ^= np.uint64(1)
    _, want_bad, want_first = orc.r1cs_residuals(n, m, *mats, bad, want_residuals=False, nthreads=4)
    assert sh.verify(bad) == (False, want_bad, par.U64_MAX)
    assert sh.verify(bad, want_first=True) == (False, want_bad, want_first), (sh.verify(bad, want_first=True), want_bad, want_first)

    # ---- four-step NTT with the HIP local steps: forward / inverse / coset, even and odd digits
    ops = par.HipOps(ctx)
    for log_n, log_r in ((12, 6), (15, 7), (16, 8)):
        N = 1 << log_n
        x = synth.random_fr(N, 11, log_n)
        d = par.DistributedNTT(log_n, ops, log_r=log_r, collectives=coll)
        mine = dev(ctx, x[d.cols_indices()])
        for shift in (None, orc.generator):
            want = orc.ntt(x, log_n, shift=shift, nthreads=4)
            out = d.forward(mine, shift=shift)
            assert np.array_equal(canon(ctx, out), want[d.rows_indices()]), f"forward mismatch log_n={log_n} rank={rank}"
            back = d.inverse(out, shift=shift)
            assert np.array_equal(canon(ctx, back), x[d.cols_indices()]), f"inverse mismatch log_n={log_n} rank={rank}"

    # ---- the whole C4 pipeline: block-cyclic rows marshalled per rank, distributed h(x) == oracle
    log_n, log_r = 14, 7
    source = lambda rows: tuple(par.gather_rows(mt, rows) for mt in mats)
    shc = par.ShardedR1CS.from_cyclic(source, n, m, log_n, log_r, ctx=ctx, collectives=coll)
    assert shc.rows.shape[0] == n // world
    assert shc.verify(bad, want_first=True) == (False, want_bad, want_first)
    dn = par.DistributedNTT(log_n, ops, log_r=log_r, collectives=coll)
    qh = par.DistributedQapH(shc, dn, orc.generator)
    h, ok = qh.run(dev(ctx, w))
    want_h, want_ok = orc.qap_h(n, m, log_n, *mats, w, nthreads=4)
    assert ok and want_ok and np.array_equal(canon(ctx, h), want_h[:1 << log_n][dn.cols_indices()]), "distributed h(x) mismatch"
    assert not qh.run(dev(ctx, bad))[1]
    dist.barrier()
    if rank == 0:
        print("dist gpu worker ok", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
