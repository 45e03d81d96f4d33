"""Worker for tests/test_distributed_cpu.py: world_size-2 (or more) gloo run of the multi-GPU host
layer with oracle-backed local kernels (the distributed LOGIC is what is under test here -- row
ownership, layouts, the exchange, the verdict collective; the HIP local kernels are covered by the
-m gpu suite against the same contract)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.c_oracle import COracle, ints_to_limbs, limbs_to_ints      # noqa: E402

par = importlib.import_module("arithmetic-circuits_amd.parallel")
synth = importlib.import_module("arithmetic-circuits_amd.synth")


def _ints(t):
    return limbs_to_ints(t.numpy().view(np.uint64).reshape(-1, 4))


def _tensor(vals):
    return torch.from_numpy(ints_to_limbs(vals).view(np.int64)).reshape(-1, 4)


class OracleOps(par.LocalOps):
    """The layout contract of acx_ntt_dist_step_dev (include/acx.h) restated with the CPU oracle on canonical
    elements (test double).  The 1/N of an inverse transform is split 1/C * 1/R over the two steps here, while
    the product folds it into step 0's twiddles: only end-to-end results are comparable, which is what is tested."""

    def __init__(self, orc):
        self.orc = orc
        self.modulus = orc.p

    def _ntt(self, vals, log_len, inverse):
        return limbs_to_ints(self.orc.ntt(ints_to_limbs(vals), log_len, inverse=inverse))

    def dist_step(self, src, dst, log_n, log_r, world, rank, inverse, step, shift, rows_t=False, mul=None, add=None):
        p = self.orc.p
        N, R = 1 << log_n, 1 << log_r
        C = N // R
        rw, cw = R // world, C // world
        w = self.orc.root_of_unity(log_n)
        if inverse:
            w = pow(w, -1, p)
        a = _ints(src)
        if mul is not None:                      # the contract of acx_ntt_dist_step_fused_dev: product on the way in
            a = [x * y % p for x, y in zip(a, _ints(mul))]
        if rows_t:                               # [k2][kl] -> [kl][k2]: the contract of ACX_DIST_ROWS_T
            assert inverse and step == 0
            a = [a[k2 * rw + kl] for kl in range(rw) for k2 in range(C)]
        out = [0] * (N // world)
        if not inverse and step == 0:            # COLS -> XCHG
            for i2l in range(cw):
                i2 = rank * cw + i2l
                col = a[i2l * R:(i2l + 1) * R]
                if shift is not None:
                    col = [x * pow(shift, i1 * C + i2, p) % p for i1, x in enumerate(col)]
                y = self._ntt(col, log_r, False)
                for k1 in range(R):
                    out[k1 * cw + i2l] = y[k1] * pow(w, i2 * k1, p) % p
        elif not inverse:                        # XCHG -> ROWS
            for kl in range(rw):
                vec = [a[s * rw * cw + kl * cw + i2l] for s in range(world) for i2l in range(cw)]
                out[kl * C:(kl + 1) * C] = self._ntt(vec, log_n - log_r, False)
        elif step == 0:                          # ROWS -> XCHG
            for kl in range(rw):
                k1 = rank * rw + kl
                y = self._ntt(a[kl * C:(kl + 1) * C], log_n - log_r, True)
                for i2 in range(C):
                    out[(i2 // cw) * rw * cw + kl * cw + (i2 % cw)] = y[i2] * pow(w, i2 * k1, p) % p
        else:                                    # XCHG -> COLS
            for i2l in range(cw):
                i2 = rank * cw + i2l
                x = self._ntt([a[k1 * cw + i2l] for k1 in range(R)], log_r, True)
                if shift is not None:
                    si = pow(shift, -1, p)
                    x = [v * pow(si, i1 * C + i2, p) % p for i1, v in enumerate(x)]
                out[i2l * R:(i2l + 1) * R] = x
        if add is not None:                      # ... and a vector of the output's layout added on the way out
            out = [(x + y) % p for x, y in zip(out, _ints(add))]
        dst.copy_(_tensor(out))

    def pointwise_h(self, a, b, c, out, log_n, shift):
        p = self.orc.p
        zinv = pow(pow(shift, 1 << log_n, p) - 1, -1, p)
        cs = _ints(c) if c is not None else [0] * a.shape[0]
        out.copy_(_tensor([(x * y - z) * zinv % p for x, y, z in zip(_ints(a), _ints(b), cs)]))

    def sub_o(self, h, o, log_n, shift):
        p = self.orc.p
        zinv = pow(pow(shift, 1 << log_n, p) - 1, -1, p)
        h.copy_(_tensor([(x - y * zinv) % p for x, y in zip(_ints(h), _ints(o))]))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = COracle("bn254")
    p = orc.p

    # ---- sharded R1CS check: verdict identical on all ranks and equal to the unsharded oracle
    s = synth.mulgraph(1 << 11, n_in=32, window=128, seed=99)
    mats, w = s.rows(), s.witness()
    n, m = s.circuit.n_rows, s.circuit.m

    class OracleRows(par.LocalRows):
        """A rank's rows checked by the CPU oracle: the test double of parallel.HipLocalRows."""

        def __init__(self, rows, m_, lm):
            self.rows, self.m, self.lm = np.asarray(rows, dtype=np.int64), m_, lm

        def prepare(self, witness):
            return witness

        def verify(self, wit, want_first=False, dots=None, h_log_n=0, h_shift=None):
            res, nbad, _ = orc.r1cs_residuals(len(self.lm[0][0]) - 1, self.m, *self.lm, wit)
            badrows = self.rows[res.any(axis=1)]
            if dots is not None:
                wi = limbs_to_ints(wit)
                vals = []
                # the contract of acx_r1cs_dots_h_dev: <A_i,w> / z, <B_i,w>, -<C_i,w> / z with z = shift^N - 1
                zinv = pow(pow(h_shift, 1 << h_log_n, p) - 1, -1, p) if h_log_n else None
                for k, (rowptr, col, val) in enumerate(self.lm):
                    v = limbs_to_ints(val)
                    f = 1 if zinv is None else (zinv, 1, p - zinv)[k]
                    vals += [sum(v[e] * wi[int(col[e])] for e in range(int(rowptr[i]), int(rowptr[i + 1]))) * f % p
                             for i in range(len(rowptr) - 1)]
                dots.copy_(_tensor(vals))
            return (torch.tensor([nbad, 0], dtype=torch.int64),
                    torch.tensor([int(badrows.min()) if nbad else (1 << 62)], dtype=torch.int64))

    bad = w.copy()
    for k in (40, 900, 2000):
        bad[k, 0] ^= np.uint64(1)
    _, want_bad, want_first = orc.r1cs_residuals(n, m, *mats, bad, want_residuals=False)
    sh = par.ShardedR1CS.from_slabs(mats, m, local_factory=OracleRows)
    assert sh.bounds[0] == 0 and sh.bounds[-1] == n and all(a <= b for a, b in zip(sh.bounds, sh.bounds[1:]))
    assert sh.verify(w) == (True, 0, par.U64_MAX)
    assert sh.verify(bad) == (False, want_bad, par.U64_MAX)                  # one collective: no first_bad
    assert sh.verify(bad, want_first=True) == (False, want_bad, want_first)
    # block-cyclic ownership, rows marshalled per rank from a row source (here: gathered from the host CSR)
    log_n, log_r = 11, 5
    source = lambda rows: tuple(par.gather_rows(mt, rows) for mt in mats)
    shc = par.ShardedR1CS.from_cyclic(source, n, m, log_n, log_r, local_factory=OracleRows)
    own = par.cyclic_rows(log_n, log_r, world, rank, ascending=True)          # from_cyclic's default: ascending row order
    assert np.array_equal(shc.rows, own) and len(own) == (1 << log_n) // world and shc.rows_t
    assert np.all(np.diff(own) > 0) and sorted(own.tolist()) == sorted(par.cyclic_rows(log_n, log_r, world, rank).tolist())
    allrows = [torch.zeros(len(own), dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allrows, torch.from_numpy(own))
    assert sorted(torch.cat(allrows).tolist()) == list(range(1 << log_n))      # a partition of the padded domain
    assert shc.verify(w) == (True, 0, par.U64_MAX)
    assert shc.verify(bad, want_first=True) == (False, want_bad, want_first)

    # ---- distributed four-step NTT == single transform: forward, inverse, coset, odd digits
    for log_n, log_r in ((8, 4), (9, 4), (10, 6)):
        N = 1 << log_n
        x = synth.random_fr(N, 7, log_n)
        d = par.DistributedNTT(log_n, OracleOps(orc), log_r=log_r)
        xt = torch.from_numpy(x.view(np.int64))
        mine = xt[d.cols_indices()].contiguous()
        for shift in (None, 5):
            want = orc.ntt(x, log_n, shift=shift)
            out = d.forward(mine, shift=shift)
            got = out.numpy().view(np.uint64)
            assert np.array_equal(got, want[d.rows_indices()]), f"forward mismatch log_n={log_n} rank={rank} shift={shift}"
            back = d.inverse(out, shift=shift)
            assert torch.equal(back, mine), f"inverse mismatch log_n={log_n} shift={shift}"

    # ---- distributed h(x): residual dots in ROWS ownership -> 7 transforms -> h in COLS ownership
    log_n, log_r = 11, 5
    dn = par.DistributedNTT(log_n, OracleOps(orc), log_r=log_r)
    qh = par.DistributedQapH(shc, dn, orc.generator)
    h, ok = qh.run(w)
    want_h, want_ok = orc.qap_h(n, m, log_n, *mats, w)
    assert ok and want_ok
    assert np.array_equal(h.numpy().view(np.uint64), want_h[:1 << log_n][dn.cols_indices()]), "distributed h(x) mismatch"
    _, ok_bad = qh.run(bad)
    assert not ok_bad
    dist.barrier()
    if rank == 0:
        print("DIST_OK world", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
