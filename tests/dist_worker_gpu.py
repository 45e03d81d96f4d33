"""Worker for test_gpu_parity.py::test_sharded_layer_two_ranks_on_one_device: the multi-GPU host layer
with the PRODUCT local kernels (libacx through HipOps / ShardedR1CS), two ranks sharing cuda:0 over gloo
(RCCL needs one GPU per rank; the sharding logic and the stream fencing do not).  gloo has no CUDA
all-to-all, so the exchange of the distributed NTT is staged through host memory here; everything else is
the code path of a multi-GPU run."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.c_oracle import COracle                  # noqa: E402

acx = importlib.import_module("arithmetic-circuits_amd")
par = importlib.import_module("arithmetic-circuits_amd.parallel")
synth = importlib.import_module("arithmetic-circuits_amd.synth")


class StagedNTT(par.DistributedNTT):
    def _all_to_all(self, send):
        if self.world == 1:
            return send
        torch.cuda.synchronize()
        s = send.cpu()
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        return r.to(send.device)


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
    sh = par.ShardedR1CS(mats, m, ctx=ctx)
    assert sh.r1cs is not None and sh.hi - sh.lo < n
    assert sh.verify(w) == (True, 0, par.U64_MAX)
    bad = w.copy()
    for k in (77, 5000, 16000):
        bad[k, 0] ^= np.uint64(1)
    _, want_bad, want_first = orc.r1cs_residuals(n, m, *mats, bad, want_residuals=False, nthreads=4)
    assert sh.verify(bad) == (False, want_bad, want_first), (sh.verify(bad), want_bad, want_first)

    # ---- four-step NTT with the HIP local transforms
    ops = par.HipOps(ctx)
    for log_n in (12, 16):
        N = 1 << log_n
        x = synth.random_fr(N, 11, log_n)
        want = orc.ntt(x, log_n, nthreads=4)
        xd = torch.from_numpy(x.view(np.int64).copy()).cuda()
        torch.cuda.synchronize()
        ctx.dev_from_canonical(N, xd.data_ptr(), xd.data_ptr())
        ctx.sync()
        d = StagedNTT(log_n, ops)
        mine = d.scatter_input(xd)
        out = d.forward(mine)
        torch.cuda.synchronize(); ctx.sync()
        flat = out.reshape(-1, 4).clone()             # the conversion below is in place: keep `out` in dev format
        torch.cuda.synchronize()                      # the clone ran on torch's stream, libacx launches on its own
        ctx.dev_to_canonical(flat.shape[0], flat.data_ptr(), flat.data_ptr())
        ctx.sync()
        got = flat.cpu().numpy().view(np.uint64)
        idx = d.output_indices().reshape(-1).numpy()
        assert np.array_equal(got, want[idx]), f"forward mismatch log_n={log_n} rank={rank}"
        back = d.inverse(out)
        torch.cuda.synchronize(); ctx.sync()

        def canonical(t):           # dev format is lazy (a value and value + p are the same element): compare canonically
            c = t.reshape(-1, 4).clone()
            torch.cuda.synchronize()
            ctx.dev_to_canonical(c.shape[0], c.data_ptr(), c.data_ptr())
            ctx.sync()
            return c
        cb, cm = canonical(back), canonical(mine)
        if not torch.equal(cb, cm):
            bad = (cb != cm).any(dim=1).reshape(back.shape[0], back.shape[1])
            raise AssertionError(f"inverse mismatch log_n={log_n} rank={rank}: {int(bad.sum())} of {bad.numel()} elements, "
                                 f"rows {bad.any(dim=1).nonzero().flatten()[:8].tolist()} cols {bad.any(dim=0).nonzero().flatten()[:8].tolist()}")
    dist.barrier()
    if rank == 0:
        print("dist gpu worker ok", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
