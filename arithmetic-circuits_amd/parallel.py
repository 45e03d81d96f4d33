"""Multi-GPU host layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference is single-threaded Haskell with nothing distributed; the sharding
below follows from the maths of the path:

  * R1CS check (`verifyAssignment`, src/QAP.hs:276-282): constraint rows are independent.  Every
    rank holds its own rows as a device-resident system (marshalled locally: no rank ever builds
    the whole matrix), the witness is replicated, and the verdict is ONE all-reduce (SUM of the
    violated-row counts, the non-canonical-witness flag rides in the same message).  The smallest
    violated row costs a second all-reduce (MIN) and is only computed on request.
  * Large NTT (`FFT.interpolate`, src/QAP.hs:521-523, at N = 2^24): four-step decomposition
    N = R*C with ONE all-to-all between the two local steps.  Each local step is one kernel
    launch of libacx (`acx_ntt_dist_step_dev`): the transposes around the exchange are strides of
    that launch and the w_N^(i2*k1) twiddle is its closing multiplication.
  * h(x) (`verificationWitnessZk`, src/QAP.hs:309-327) over all GPUs: rows are owned
    block-cyclically (rank g: rows r with (r mod R) in block g) so that the residual kernel writes
    <A_i,w>, <B_i,w>, <C_i,w> straight into the layout the first inverse transform reads; six
    distributed transforms = six all-to-alls, nothing else moves.

This is the one-process-per-GPU launcher (what `torch.distributed.run` starts for bench.py): the collectives live here,
above the C ABI, and libacx sees one GPU per process (a C host does the same with rcclCommInitRank / ncclAllToAll:
INTEGRATION.md section 5).  The single-process alternative -- a device list, the sharding and the collectives INSIDE libacx -- is
`acx_mgpu_*` (engine.MultiGpu).

Three seams keep the sharding logic testable without several GPUs, each with exactly one product implementation here:
`LocalRows` (a rank's rows: `HipLocalRows` = a device-resident acx_r1cs), `LocalOps` (a rank's transform steps: `HipOps`) and
`Collectives` (all-reduce / all-to-all on `torch.distributed`, backend nccl = RCCL).  The CPU test-suite (tests/dist_worker.py)
supplies oracle-backed rows and steps over gloo; the one-GPU tests supply a host-staged exchange (tests/helpers_dist.py)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .engine import Context, R1CS

U64_MAX = (1 << 64) - 1


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


# ------------------------------------------------------------------------------------ collectives
class Collectives:
    """The two collectives of the path on a process group: torch.distributed, i.e. RCCL over xGMI with the nccl backend.
    all_to_all returns a Work handle when the exchange is still in flight (RCCL runs it on its own stream), else None."""

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = _world(group)

    def all_reduce(self, t: torch.Tensor, op) -> None:
        dist.all_reduce(t, op=op, group=self.group)

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor):
        return dist.all_to_all_single(recv, send, group=self.group, async_op=True)

    def overlaps(self, like: torch.Tensor) -> bool:
        """True when all_to_all returns with the exchange still in flight on another stream."""
        return bool(like.is_cuda)


# ------------------------------------------------------------------------------------ row ownership
def shard_bounds(rowptrs: Sequence[np.ndarray], world: int) -> List[int]:
    """Split rows [0, n) into `world` contiguous slabs balanced by nnz of A+B+C (Split gates make
    257-row bursts of very uneven length, test/Test/Circuit/Arithmetic.hs:123).  Returns world+1
    boundaries."""
    n = len(rowptrs[0]) - 1
    cost = np.zeros(n + 1, dtype=np.int64)
    for rp in rowptrs:
        cost += np.asarray(rp, dtype=np.int64)
    cost += np.arange(n + 1, dtype=np.int64)          # every row costs at least its epilogue
    total = int(cost[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(cost, total * r // world, side="left")))
    bounds.append(n)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def slice_rows(mat, lo: int, hi: int):
    rowptr, col, val = mat
    e0, e1 = int(rowptr[lo]), int(rowptr[hi])
    return (np.asarray(rowptr[lo:hi + 1], dtype=np.uint32) - np.uint32(e0)), col[e0:e1], val[e0:e1]


def gather_rows(mat, rows: np.ndarray):
    """CSR of the given rows (any order; indices >= n give empty rows) of a host CSR triple."""
    rowptr, col, val = mat
    rp = np.asarray(rowptr, dtype=np.int64)
    n = rp.shape[0] - 1
    rows = np.asarray(rows, dtype=np.int64)
    live = rows < n
    loc = np.where(live, rows, 0)
    lens = np.where(live, rp[loc + 1] - rp[loc], 0)
    new_rp = np.concatenate([[0], np.cumsum(lens)])
    owner = np.repeat(np.arange(rows.shape[0], dtype=np.int64), lens)
    src = rp[loc][owner] + (np.arange(int(new_rp[-1]), dtype=np.int64) - new_rp[:-1][owner])
    return new_rp.astype(np.uint32), np.ascontiguousarray(col[src]), np.ascontiguousarray(val[src])


def cyclic_rows(log_n: int, log_r: int, world: int, rank: int, ascending: bool = False) -> np.ndarray:
    """Global row numbers owned by `rank` under the block-cyclic ownership of SURVEY.md 8(e): row = (rank*R/W + kl) + k2*R.
    N/W entries (rows >= n are padding).  Default: local ROWS-layout order [kl][k2] (what a forward transform produces);
    ascending: [k2][kl], the rows in increasing order -- the order a rank LOADS its rows in, because the gathers of the
    rows in flight then stay inside a window 8x narrower; the dot products come out as the transposed ROWS block, which the
    first inverse step reads through its strides (ACX_DIST_ROWS_T)."""
    R, C = 1 << log_r, 1 << (log_n - log_r)
    rw = R // world
    kl = np.arange(rw, dtype=np.int64).reshape(-1, 1)
    k2 = np.arange(C, dtype=np.int64).reshape(1, -1)
    rows = rank * rw + kl + k2 * R
    return (rows.T if ascending else rows).reshape(-1)


def wire_range(wire_begin: int, wire_count: int, world: int, rank: int) -> Tuple[int, int]:
    """Per-wire polynomials (`createPolynomialsFFT`, src/QAP.hs:512-525) shard by WIRE with no communication:
    rank's contiguous part (first wire, count) of a wire range, the split acx_mgpu_qap_columns uses inside one process.  A
    one-process-per-GPU host loads the whole system on every rank (a column's interpolation needs every row) and calls
    R1CS.qap_columns(matrix, *wire_range(...)) on its own part; the parts tile the range in rank order."""
    w0, w1 = wire_count * rank // world, wire_count * (rank + 1) // world
    return wire_begin + w0, w1 - w0


RowSource = Callable[[np.ndarray], Tuple[tuple, tuple, tuple]]


class LocalRows:
    """A rank's rows of a sharded system: the seam between the sharding logic and the per-rank kernels.
    `rows` = the rank's global row numbers in local order."""
    rows: np.ndarray
    m: int

    def prepare(self, witness: np.ndarray):
        """the (replicated) witness in whatever form verify() takes"""
        raise NotImplementedError

    def verify(self, w, want_first: bool = False, dots: Optional[torch.Tensor] = None, h_log_n: int = 0,
               h_shift: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Local check on a prepared witness: (verdict = [violated rows, non-canonical flag], first = [smallest violated GLOBAL
        row or 2^62]) as int64 tensors, not yet reduced over the ranks; dots (3 * len(rows) elements) receives <A_i,w>, <B_i,w>,
        <C_i,w> of the local rows in local order when given.  h_log_n != 0: the dots are stored FOR h(x) over 2^h_log_n points
        (the global transform size): <A_i,w> / z, <B_i,w>, -<C_i,w> / z with z = h_shift^N - 1 (acx_r1cs_dots_h_dev)."""
        raise NotImplementedError


class HipLocalRows(LocalRows):
    """The product implementation: the rank's rows as a device-resident acx_r1cs, checked by the HIP residual kernel."""

    def __init__(self, ctx: Context, rows: np.ndarray, m: int, local_mats):
        if ctx is None:
            raise RuntimeError("a sharded system needs a GPU Context (libacx has no CPU fallback)")
        self.ctx, self.m = ctx, m
        self.rows = np.asarray(rows, dtype=np.int64)
        self.r1cs = R1CS.load(ctx, self.rows.shape[0], m, *local_mats)
        dev = f"cuda:{ctx.device}"
        self._res = torch.zeros(2, dtype=torch.int64, device=dev)
        self._flag = torch.zeros(2, dtype=torch.int32, device=dev)       # [0] = non-canonical witness
        self._stream = torch.cuda.ExternalStream(ctx.stream)
        self._monotone = self.rows.shape[0] < 2 or bool(np.all(np.diff(self.rows) > 0))

    def prepare(self, witness: np.ndarray) -> torch.Tensor:
        """Upload + convert the (replicated) witness; canonicity is checked on the device like the single-GPU path
        does (acx_r1cs_verify): the flag travels with the verdict's all-reduce."""
        ctx = self.ctx
        w = torch.from_numpy(np.ascontiguousarray(witness, dtype=np.uint64).view(np.int64)).to(f"cuda:{ctx.device}")
        self._flag.zero_()
        torch.cuda.synchronize()
        ctx.dev_from_canonical(self.m, w.data_ptr(), w.data_ptr(), self._flag.data_ptr())
        return w

    def verify(self, w: torch.Tensor, want_first: bool = False, dots: Optional[torch.Tensor] = None, h_log_n: int = 0, h_shift=None):
        none = 1 << 62
        res_vec = None
        # libacx launches on its own stream: order it after whatever the caller's stream still has in flight on `w` / `dots`
        # (e.g. the fill of a freshly allocated torch.zeros buffer), as HipOps._fenced does for the transform steps
        self._stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._stream):
            self._res.copy_(torch.tensor([0, -1], dtype=torch.int64), non_blocking=False)
            if want_first and not self._monotone:
                res_vec = torch.empty((self.rows.shape[0], 4), dtype=torch.int64, device=self._res.device)
            if h_log_n and dots is not None and res_vec is None:
                self.r1cs.dots_h_dev(w.data_ptr(), self._res.data_ptr(), dots.data_ptr(), h_log_n, h_shift)
            else:
                assert not h_log_n, "scaled dots come without the residual vector"
                self.r1cs.verify_dev(w.data_ptr(), self._res.data_ptr(), d_dots=dots.data_ptr() if dots is not None else 0,
                                     d_residuals=res_vec.data_ptr() if res_vec is not None else 0)
            verdict = torch.stack([self._res[0], self._flag[0].to(torch.int64)])
            first = torch.full((1,), none, dtype=torch.int64, device=self._res.device)
            if want_first and self._monotone:
                # the kernel's first_bad is the smallest LOCAL row (unsigned, UINT64_MAX = none); local order is
                # increasing in the global row number here, so it maps straight to the global row
                pos = self._res[1:2]
                rows = torch.from_numpy(self.rows).to(pos.device)
                first = torch.where(pos < 0, first, rows[pos.clamp(min=0, max=rows.shape[0] - 1)])
            elif want_first:
                # local order is not global order (ROWS-order ownership): take the residual vector (on request only)
                self.ctx.dev_to_canonical(res_vec.shape[0], res_vec.data_ptr(), res_vec.data_ptr())
                bad = (res_vec != 0).any(dim=1)
                rows = torch.from_numpy(self.rows).to(bad.device)
                first = torch.where(bad, rows, torch.full_like(rows, none)).min().reshape(1)
        self._stream.synchronize()
        return verdict, first


LocalFactory = Callable[[np.ndarray, int, list], LocalRows]


class ShardedR1CS:
    """A constraint system whose rows are sharded over the ranks of a process group.

    Ownership: `rows` = this rank's global row numbers in local order (`from_slabs`: a contiguous nnz-balanced slab;
    `from_cyclic`: the block-cyclic ownership the distributed h(x) pipeline needs).  Only those rows are marshalled and
    uploaded on this rank.  `ctx` selects the product's HipLocalRows; `local_factory(rows, m, local_mats)` supplies
    another LocalRows, `collectives` another Collectives."""

    def __init__(self, local: LocalRows, n: int, m: int, collectives: Optional[Collectives] = None):
        self.local = local
        self.coll = collectives if collectives is not None else Collectives()
        self.world, self.rank = self.coll.world, self.coll.rank
        self.n, self.m = n, m
        self.rows_t = False

    @property
    def rows(self) -> np.ndarray:
        return self.local.rows

    @property
    def r1cs(self):
        return getattr(self.local, "r1cs", None)

    # -- constructors ---------------------------------------------------------------------------
    @classmethod
    def from_source(cls, source: RowSource, rows: np.ndarray, n: int, m: int, ctx: Optional[Context] = None,
                    local_factory: Optional[LocalFactory] = None, collectives: Optional[Collectives] = None) -> "ShardedR1CS":
        """Rank-local construction: `source(rows)` returns the CSR triples of exactly these global rows."""
        mats = list(source(rows))
        rows = np.asarray(rows, dtype=np.int64)
        local = local_factory(rows, m, mats) if local_factory is not None else HipLocalRows(ctx, rows, m, mats)
        return cls(local, n, m, collectives)

    @classmethod
    def from_slabs(cls, mats, m: int, collectives: Optional[Collectives] = None, **kw) -> "ShardedR1CS":
        """Contiguous slabs balanced by nnz.  `mats` is the full host CSR triple (cheap for the sizes where a
        single host can hold it; large jobs use `from_source`)."""
        coll = collectives if collectives is not None else Collectives()
        n = len(mats[0][0]) - 1
        bounds = shard_bounds([mt[0] for mt in mats], coll.world)
        lo, hi = bounds[coll.rank], bounds[coll.rank + 1]
        self = cls.from_source(lambda rows: [slice_rows(mt, lo, hi) for mt in mats], np.arange(lo, hi, dtype=np.int64), n, m,
                               collectives=coll, **kw)
        self.bounds = bounds
        return self

    @classmethod
    def from_cyclic(cls, source: RowSource, n: int, m: int, log_n: int, log_r: int, ascending: bool = True,
                    collectives: Optional[Collectives] = None, **kw) -> "ShardedR1CS":
        """Block-cyclic ownership for the distributed h(x); `ascending` (default) loads the rank's rows in increasing order
        and marks the dot products as the transposed ROWS block (DistributedQapH passes that on to the first inverse steps)."""
        coll = collectives if collectives is not None else Collectives()
        self = cls.from_source(source, cyclic_rows(log_n, log_r, coll.world, coll.rank, ascending), n, m, collectives=coll, **kw)
        self.rows_t = ascending
        return self

    # -- verifyAssignment -------------------------------------------------------------------------
    def verify(self, witness: np.ndarray, want_first: bool = False) -> Tuple[bool, int, int]:
        """verifyAssignment over all shards: (ok, n_bad, first_bad) identical on every rank.  ONE collective;
        first_bad (smallest violated global row) costs a second one and is U64_MAX unless want_first."""
        verdict, first = self.local.verify(self.local.prepare(witness), want_first)
        return self._reduce(verdict, first, want_first)

    def verify_dev(self, w, want_first: bool = False, dots: Optional[torch.Tensor] = None):
        """Local launch on a prepared (device-resident) witness; returns the (not yet reduced) verdict tensors."""
        return self.local.verify(w, want_first, dots)

    def dots(self, w, out: torch.Tensor, h_log_n: int = 0, h_shift: Optional[int] = None):
        """<A_i,w>, <B_i,w>, <C_i,w> of the local rows into out (3 * rows elements), plus the local verdict; h_log_n: stored for
        h(x) over 2^h_log_n points on the coset h_shift * <omega> (LocalRows.verify)."""
        return self.local.verify(w, False, out, h_log_n, h_shift)

    def _reduce(self, verdict: torch.Tensor, first: torch.Tensor, want_first: bool) -> Tuple[bool, int, int]:
        if self.world > 1:
            self.coll.all_reduce(verdict, dist.ReduceOp.SUM)          # THE verdict collective
            if want_first:
                self.coll.all_reduce(first, dist.ReduceOp.MIN)
        n_bad, noncanon = int(verdict[0]), int(verdict[1])
        if noncanon:
            from ._lib import AcxError, STATUS
            raise AcxError(STATUS["NONCANONICAL"], "element >= p")
        fb = int(first[0])
        return n_bad == 0, n_bad, (fb if (n_bad and want_first) else U64_MAX)


# ------------------------------------------------------------------------------------ distributed NTT
class LocalOps:
    """Per-rank kernels the distributed pipeline is built from.  Tensors are int64 views of field
    elements, shape (count, 4), in whatever element format the implementation uses.  `modulus`: the field's prime (the
    pipeline inverts its coset generator)."""

    modulus: int = 0

    def dist_step(self, src: torch.Tensor, dst: torch.Tensor, log_n: int, log_r: int, world: int, rank: int,
                  inverse: bool, step: int, shift: Optional[int], rows_t: bool = False,
                  mul: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None) -> None:
        """rows_t (inverse step 0 only): src is the transposed ROWS block [k2][kl] (include/acx.h, ACX_DIST_ROWS_T).
        mul: the step transforms src[i] * mul[i]; add: dst[k] = X[k] + add[k] (acx_ntt_dist_step_fused_dev)."""
        raise NotImplementedError

    def pointwise_h(self, a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor], out: torch.Tensor, log_n: int, shift: int) -> None:
        """out = (a*b - c) / (shift^N - 1) elementwise (src/QAP.hs:325-327 on the coset); c = None: a*b / (shift^N - 1)."""
        raise NotImplementedError

    def sub_o(self, h: torch.Tensor, o: torch.Tensor, log_n: int, shift: int) -> None:
        """h -= o / (shift^N - 1) elementwise: O(x) enters the quotient in coefficient form."""
        raise NotImplementedError


class HipOps(LocalOps):
    """libacx kernels on dev-format CUDA tensors (the product path).  libacx launches on its
    context's own HIP stream; each call is fenced against torch's current stream in both
    directions so that RCCL collectives (on torch's stream) order correctly around it."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.modulus = ctx.p
        self._ext = torch.cuda.ExternalStream(ctx.stream)

    def _fenced(self, fn):
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != self._ext.cuda_stream:
            self._ext.wait_stream(cur)
        fn()
        if cur.cuda_stream != self._ext.cuda_stream:
            cur.wait_stream(self._ext)

    def dist_step(self, src, dst, log_n, log_r, world, rank, inverse, step, shift, rows_t=False, mul=None, add=None):
        assert src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()
        assert all(t is None or (t.is_cuda and t.is_contiguous()) for t in (mul, add))
        self._fenced(lambda: self.ctx.ntt_dist_step_dev(src.data_ptr(), dst.data_ptr(), log_n, log_r, world, rank, inverse, step, shift, rows_t,
                                                        d_mul=mul.data_ptr() if mul is not None else 0,
                                                        d_add=add.data_ptr() if add is not None else 0))

    def pointwise_h(self, a, b, c, out, log_n, shift):
        self._fenced(lambda: self.ctx.qap_pointwise_dev(a.data_ptr(), b.data_ptr(), c.data_ptr() if c is not None else None,
                                                        out.data_ptr(), a.shape[0], log_n, shift))

    def sub_o(self, h, o, log_n, shift):
        self._fenced(lambda: self.ctx.qap_sub_o_dev(h.data_ptr(), o.data_ptr(), h.shape[0], log_n, shift))


class DistributedNTT:
    """Length-N = 2^log_n transform over `world` ranks, four-step with index split i = i1*C + i2,
    k = k1 + k2*R  (R = 2^log_r rows, C = N/R columns):

        X[k1 + k2 R] = sum_{i2} w_C^{i2 k2} * w_N^{i2 k1} * ( sum_{i1} w_R^{i1 k1} x[i1 C + i2] )

    Local layouts (include/acx.h, acx_ntt_dist_step_dev), N/W elements each:
        COLS [i2l][i1]   x[i1*C + g*C/W + i2l]          ROWS [kl][k2]   X[(g*R/W + kl) + k2*R]
    forward(): COLS -> ROWS, inverse(): ROWS -> COLS.  One all-to-all each way (N*32*(W-1)/W bytes over xGMI,
    every link busy); a pipeline of transforms (the 6 NTTs of h(x)) alternates the two layouts, so nothing is
    ever re-ordered in between."""

    def __init__(self, log_n: int, ops: LocalOps, group=None, log_r: Optional[int] = None, force_collective: bool = False,
                 collectives: Optional[Collectives] = None):
        self.coll = collectives if collectives is not None else Collectives(group)
        self.group = self.coll.group
        self.world, self.rank = self.coll.world, self.coll.rank
        self.force_collective = force_collective        # issue the collective even with one rank: a one-rank RCCL communicator still runs the exchange
        self.log_n = log_n
        self.log_r = log_r if log_r is not None else log_n // 2
        self.log_c = log_n - self.log_r
        self.R, self.C = 1 << self.log_r, 1 << self.log_c
        if self.R % self.world or self.C % self.world:
            raise ValueError("world size must divide both factors of N")
        self.local = (1 << log_n) // self.world
        self.ops = ops
        self._pool = []                                  # (send, recv) pairs, one per transform in flight

    # -- layout helpers (tests / single-rank users) ---------------------------------------------
    def cols_indices(self) -> np.ndarray:
        """Natural index i of every element of this rank's COLS block, in local order."""
        cw = self.C // self.world
        i2 = (self.rank * cw + np.arange(cw, dtype=np.int64)).reshape(-1, 1)
        i1 = np.arange(self.R, dtype=np.int64).reshape(1, -1)
        return (i1 * self.C + i2).reshape(-1)

    def rows_indices(self) -> np.ndarray:
        """Natural index k = k1 + k2*R of every element of this rank's ROWS block, in local order."""
        return cyclic_rows(self.log_n, self.log_r, self.world, self.rank)

    def _buffers(self, like: torch.Tensor, slot: int = 0):
        while len(self._pool) <= slot:
            self._pool.append(None)
        pair = self._pool[slot]
        if pair is None or pair[0].device != like.device or pair[0].dtype != like.dtype:
            send = torch.empty((self.local, 4), dtype=like.dtype, device=like.device)
            pair = self._pool[slot] = (send, torch.empty_like(send))
        return pair

    def _exchanges(self) -> bool:
        return self.world > 1 or self.force_collective

    def overlapped(self, like: torch.Tensor) -> bool:
        """True when begin() returns with the exchange still in flight: RCCL runs the collective on its own stream, so
        the caller's next local step (another transform of the pipeline) overlaps it."""
        return bool(self._exchanges() and self.coll.overlaps(like))

    def stream_context(self):
        """The HIP stream every step of a pipeline is issued on (libacx's own, for HipOps): collectives are ordered
        against it.  Fenced against the caller's current stream on entry and exit, so tensors produced or consumed by
        torch operations outside the block need no further care."""
        import contextlib
        ext = getattr(self.ops, "_ext", None)
        if ext is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def fenced():
            outer = torch.cuda.current_stream()
            if outer.cuda_stream == ext.cuda_stream:
                yield
                return
            ext.wait_stream(outer)
            with torch.cuda.stream(ext):
                yield
            outer.wait_stream(ext)
        return fenced()

    def begin(self, x: torch.Tensor, inverse: bool, shift: Optional[int] = None, slot: int = 0, rows_t: bool = False,
              mul: Optional[torch.Tensor] = None):
        """First local step of one transform and the START of its all-to-all (src/QAP.hs:512-525's transform, sharded).
        Returns a token for finish().  Transforms in flight at the same time need different slots (buffer pairs).
        mul: the transform of the pointwise product x * mul (same layout), formed as the step loads its points."""
        assert x.shape == (self.local, 4) and x.is_contiguous()
        send, recv = self._buffers(x, slot)
        a = (self.log_n, self.log_r, self.world, self.rank, inverse)
        self.ops.dist_step(x, send, *a, 0, shift, rows_t, mul=mul)
        work, got = None, send
        if self._exchanges():
            # enqueued behind the local step on the current stream; with RCCL it returns at once and the xGMI links work while
            # the SIMDs run the next transform's local step
            work = self.coll.all_to_all(recv, send)
            got = recv
        return (work, got, a, shift)

    def finish(self, token, out: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Wait (stream-side) for the exchange, then the second local step.  add: a vector in the output's layout added
        behind the step's closing multiplication."""
        work, got, a, shift = token
        if work is not None:
            work.wait()
        if out is None:
            out = torch.empty_like(got)
        self.ops.dist_step(got, out, *a, 1, shift, add=add)
        return out

    def _run(self, x: torch.Tensor, out: Optional[torch.Tensor], inverse: bool, shift: Optional[int], rows_t: bool = False) -> torch.Tensor:
        with self.stream_context():
            return self.finish(self.begin(x, inverse, shift, rows_t=rows_t), out)

    def forward(self, cols: torch.Tensor, out: Optional[torch.Tensor] = None, shift: Optional[int] = None) -> torch.Tensor:
        return self._run(cols, out, False, shift)

    def inverse(self, rows: torch.Tensor, out: Optional[torch.Tensor] = None, shift: Optional[int] = None, rows_t: bool = False) -> torch.Tensor:
        """rows_t: `rows` is the transposed ROWS block [k2][kl] (the dots of rows loaded in ascending order)."""
        return self._run(rows, out, True, shift, rows_t)


class DistributedQapH:
    """`verificationWitness` (src/QAP.hs:292-327, delta = 0) over all GPUs: h = (L*R - O) / (x^N - 1).

    Rows are owned block-cyclically (ShardedR1CS.from_cyclic), so the residual kernel's <A_i,w>, <B_i,w>, <C_i,w>
    ARE the ROWS layout of three evaluation vectors -- stored as <A_i,w> / z, <B_i,w>, -<C_i,w> / z (z = g^N - 1 on the
    coset): the factors ride on the residual launch's stores.  Then 3 inverse transforms (-> coefficients, COLS), 2 forward
    coset transforms (L / z and R -> ROWS) and 1 inverse coset transform that takes their PRODUCT as its first step loads
    the points and adds -O / z (coefficient form, COLS) behind the closing multiplication of its second step: h's
    coefficients in COLS layout (rank g holds h[i1*C + g*C/W + i2l]).  O(x) never needs its coset evaluations -- the
    transforms are linear and icoset(coset(O)) = O -- so there are six all-to-alls, not seven, one verdict all-reduce, and
    no elementwise pass outside the transforms (history: profiles/HISTORY.md "h(x) in round 3")."""

    def __init__(self, sharded: ShardedR1CS, ntt: DistributedNTT, generator: int):
        assert sharded.rows.shape[0] == ntt.local
        self.sharded, self.ntt, self.g = sharded, ntt, generator
        self._bufs = None

    def run(self, w) -> Tuple[torch.Tensor, bool]:
        """w: the replicated witness (device tensor in dev format on the product path)."""
        nt, L = self.ntt, self.ntt.local
        if self._bufs is None:
            dev = w.device if isinstance(w, torch.Tensor) else "cpu"
            # no fill: the local system has exactly N/W rows (padding rows are empty rows), so the residual launch writes every
            # element of the three dot-product vectors
            self._bufs = (torch.empty((3 * L, 4), dtype=torch.int64, device=dev),
                          torch.empty((3 * L, 4), dtype=torch.int64, device=dev))
        dots, tmp = self._bufs
        verdict, first = self.sharded.dots(w, dots, h_log_n=nt.log_n, h_shift=self.g)     # rows of padding give 0
        part = lambda t, k: t[k * L:(k + 1) * L]
        # Software pipeline over the three vectors: the exchange of vector k runs (on RCCL's stream) under the local
        # steps of vector k+1, and a vector's forward transform starts as soon as its inverse one is complete -- of
        # the six all-to-alls only the last one has no local work to hide behind.  Without an overlapping backend
        # (one rank, gloo) begin() completes the exchange itself and this is the plain sequence.
        # Nobody needs the plain coefficients of L and R: their coset factor g^i rides on the closing multiplication of their
        # INVERSE transform (an inverse coset transform with shift 1/g multiplies by g^i) instead of on the load of the forward
        # one -- one product per element less on the steps that have it.  O stays in plain coefficient form.
        ginv = pow(self.g, -1, self.ntt.ops.modulus)
        with nt.stream_context():
            inv = [nt.begin(part(dots, k), True, ginv if k < 2 else None, slot=k, rows_t=self.sharded.rows_t) for k in range(3)]
            fwd = []
            for k in range(3):
                nt.finish(inv[k], out=part(tmp, k))
                if k < 2:
                    fwd.append(nt.begin(part(tmp, k), False, None, slot=k))
            for k in range(2):
                nt.finish(fwd[k], out=part(dots, k))
            # (L / z) * R on the way in, -O / z on the way out (O's coefficients are in COLS ownership, like h)
            h = nt.finish(nt.begin(dots[:L], True, self.g, slot=0, mul=dots[L:2 * L]), out=tmp[:L], add=part(tmp, 2))
        ok, _, _ = self.sharded._reduce(verdict, first, False)
        return h, ok
