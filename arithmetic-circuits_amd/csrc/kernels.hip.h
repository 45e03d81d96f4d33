// kernels.hip.h -- gfx950 kernels of the R1CS / QAP hot path.  All new code: the reference
// (pure Haskell) has no kernels; each kernel names the reference computation it performs.
#pragma once
#include <utility>
#include "fr.hip.h"
#include "mem.hip.h"
#include "ntt_pass.hip.h"

namespace acx {

constexpr int kBlock = 256;

// Device CSR view (values in dev format: 2 x uint4 per entry).
struct CsrDev {
    const u32* rowptr;
    const u32* col;
    const uint4* val;
};

// ---------------------------------------------------------------------------------------------
// K7: canonical <-> dev (lazy Montgomery) conversion at the ABI edge.
// to_dev validates canonicity (galois-field keeps residues canonical; a host that passes >= p
// gets ACX_ERR_NONCANONICAL): *err is set to 1 if any element >= p.
template <class F, bool TO_DEV>
__global__ __launch_bounds__(kBlock) void k_convert(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                   u64 count, u32* __restrict__ err) {
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < count; i += (u64)gridDim.x * kBlock) {
        Fe x = fe_load(in + 2 * i);
        if (TO_DEV) {
            if (err != nullptr && !fe_lt_p<F>(x)) atomicOr(err, 1u);
            fe_store(out + 2 * i, fe_to_mont<F>(x));
        } else {
            fe_store(out + 2 * i, fe_from_mont<F>(x));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: R1CS residual check = `verifyAssignment` (/root/reference/src/QAP.hs:276-327) in the
// evaluation domain: r_i = <A_i,w> * <B_i,w> - <C_i,w> for every constraint row i.
// One row per lane; a row's entries are contiguous in the CSR value stream.
// <M_row, w> with deferred reduction: raw limb products of up to kWideTerms entries are summed in
// 64-bit column accumulators and Montgomery-reduced once (81 mads per entry + ~100 per row
// instead of 171 per entry).  UNIT = every stored value of this matrix is the field's 1 (the C
// matrix of every gate the reference emits, src/QAP.hs:371-474): the dot is a plain sum of
// witness entries and the value stream is never read.
template <class F, bool UNIT>
__device__ __forceinline__ Fe csr_range_dot(const CsrDev& M, const uint4* __restrict__ w, u32 e0, u32 e1) {
    Fe acc = fe_zero();
    if (UNIT) {
        for (u32 e = e0; e < e1; ++e) {
            const Fe x = fe_load(w + 2 * (u64)M.col[e]);
            acc = (e == e0) ? x : fe_add<F>(acc, x);
        }
        return acc;
    }
    for (u32 base = e0; base < e1; base += kWideTerms) {
        const u32 end = (e1 - base > (u32)kWideTerms) ? base + kWideTerms : e1;
        Wide wide;
        wide_zero(wide);
        for (u32 e = base; e < end; ++e) {
            const Fe v = fe_load(M.val + 2 * (u64)e);
            const Fe x = fe_load(w + 2 * (u64)M.col[e]);
            wide_mac(wide, v, x);
        }
        const Fe part = wide_reduce<F>(wide);
        acc = (base == e0) ? part : fe_add<F>(acc, part);
    }
    return acc;
}
template <class F, bool UNIT>
__device__ __forceinline__ Fe csr_row_dot(const CsrDev& M, const uint4* __restrict__ w, u64 row) {
    return csr_range_dot<F, UNIT>(M, w, M.rowptr[row], M.rowptr[row + 1]);
}

// ---- SELL-64 device layout of a constraint matrix -------------------------------------------------
// Rows are grouped in slices of 64 (one wavefront).  Inside a slice, slot j of all 64 rows is stored
// together: a wave reading "entry j of my row" issues perfectly coalesced 1 KiB loads and every
// 128-byte line of the stream is fetched from HBM exactly once.  (A row-per-lane walk over plain
// CSR touched each line in three loop iterations far apart: measured 2.1x HBM over-fetch,
// profiles/r01_r1cs_direct_2p22_overfetch.txt; staging CSR through LDS removed the over-fetch but
// its barriers cost more than it saved, profiles/r01_r1cs_lds_staged_2p22.txt.)  Rows are sorted by
// length inside windows of kSellWindow rows so that slices are uniform: no padding to stream and no
// lane idles in the multiply loop.  Rows longer than kSellMaxLen in any matrix (the 2^j row of a
// Split gate, src/QAP.hs:447-459) stay in CSR and are handled by k_r1cs_residual_rows.
constexpr int kSlice = 64;
constexpr int kSellMaxLen = 6;   // = kWideTerms: one deferred reduction per row and matrix, no partial accumulator
constexpr int kSellWindow = 4096;
constexpr u32 kNoRow = 0xffffffffu;

struct SellDev {
    const u32* slice_ofs;  // [n_slices + 1] slot offsets
    const uint2* tail;     // [slots * 64]: { limb 8 of the value, column }; column kNoRow marks padding
    const uint4* val;      // [slots * 2 * 64]: limbs 0-3 / 4-7 of slot q, lane l at (2q + h) * 64 + l
};

// Gather one matrix from its (device, dev-format) CSR into the SELL arrays.  Values are stored as
// 29-bit limbs (40 bytes per entry with the column index instead of 36): the residual kernel is
// bound by integer VALU issue, not by HBM, and this removes the limb split from its inner loop.
__global__ __launch_bounds__(kBlock) void k_build_sell(CsrDev M, const u32* __restrict__ perm,
                                                      const u32* __restrict__ slice_ofs, u32 n_slices,
                                                      uint2* __restrict__ tail, uint4* __restrict__ val) {
    const u32 slice = blockIdx.x * (kBlock / kSlice) + (threadIdx.x / kSlice);
    const u32 lane = threadIdx.x % kSlice;
    if (slice >= n_slices) return;
    const u32 q0 = slice_ofs[slice], q1 = slice_ofs[slice + 1];
    const u32 row = perm[slice * kSlice + lane];
    u32 e0 = 0, len = 0;
    if (row != kNoRow) { e0 = M.rowptr[row]; len = M.rowptr[row + 1] - e0; }
    for (u32 q = q0; q < q1; ++q) {
        const u32 j = q - q0;
        Fe v = fe_zero();
        u32 c = kNoRow;
        if (j < len) { v = fe_load(M.val + 2 * (u64)(e0 + j)); c = M.col[e0 + j]; }
        tail[(u64)q * kSlice + lane] = make_uint2(v.l[8], c);
        val[(2 * (u64)q) * kSlice + lane] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        val[(2 * (u64)q + 1) * kSlice + lane] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    }
}

// The same for a matrix whose SELL rows carry only small coefficients (|c| <= kSmallCoeffMax, src/Circuit/Expr.hs
// compiles programs to +-1, +-2 and small constants): the slot is {signed coefficient, column} and there is no value
// stream at all -- 8 bytes per entry instead of 40.  The coefficient is recovered from the Montgomery CSR value.
template <class F>
__global__ __launch_bounds__(kBlock) void k_build_sell_small(CsrDev M, const u32* __restrict__ perm,
                                                            const u32* __restrict__ slice_ofs, u32 n_slices,
                                                            uint2* __restrict__ tail, u32* __restrict__ err) {
    const u32 slice = blockIdx.x * (kBlock / kSlice) + (threadIdx.x / kSlice);
    const u32 lane = threadIdx.x % kSlice;
    if (slice >= n_slices) return;
    const u32 q0 = slice_ofs[slice], q1 = slice_ofs[slice + 1];
    const u32 row = perm[slice * kSlice + lane];
    u32 e0 = 0, len = 0;
    if (row != kNoRow) { e0 = M.rowptr[row]; len = M.rowptr[row + 1] - e0; }
    for (u32 q = q0; q < q1; ++q) {
        const u32 j = q - q0;
        i32 cf = 0;
        u32 c = kNoRow;
        if (j < len) {
            const Fe v = fe_from_mont<F>(fe_load(M.val + 2 * (u64)(e0 + j)));   // canonical
            u32 hi = 0, hin = 0;
            Fe neg;                                              // p - v
            i32 br = 0;
#pragma unroll
            for (int k = 0; k < kLimbs; ++k) {
                const i32 t = (i32)F::P[k] - (i32)v.l[k] + br;
                neg.l[k] = (u32)t & kLimbMask;
                br = t >> kLimbBits;
            }
#pragma unroll
            for (int k = 1; k < kLimbs; ++k) { hi |= v.l[k]; hin |= neg.l[k]; }
            if (hi == 0 && v.l[0] <= (u32)kSmallCoeffMax) cf = (i32)v.l[0];
            else if (hin == 0 && neg.l[0] <= (u32)kSmallCoeffMax) cf = -(i32)neg.l[0];
            else atomicAdd(err, 1u);                             // the host classified this matrix as small: cannot happen
            c = M.col[e0 + j];
        }
        tail[(u64)q * kSlice + lane] = make_uint2((u32)cf, c);
    }
}

#ifdef ACX_K2_TRACE
// Development build (tools/k2_trace.py): where a wave of the residual kernel spends its life.  Per role (A wave / B-C-closing
// wave) the SUM over all waves of: [0] waves, [1] cycles until the descriptor is in registers, [2] until the slice offsets are,
// [3] until the first slot's stream words have arrived, [4] until the dot product of the first matrix is reduced, [5] total.
constexpr u32 kK2TraceWaves = 1u << 16;               // one record per workgroup and role: plain stores (same-address atomics from
__device__ unsigned long long g_k2_trace[2][kK2TraceWaves][6];   // 65536 waves serialise at ~10 per us and would distort what is measured)
__device__ __forceinline__ u64 k2_now() { return __builtin_readcyclecounter(); }
struct K2Trace { u64 t0, t_desc, t_ofs, t_first, t_dot; };
#define K2_TRACE_ARG , K2Trace* tr
#define K2_TRACE_PASS(x) , x
#else
#define K2_TRACE_ARG
#define K2_TRACE_PASS(x)
#endif

// slots [q0, q1) of one slice.  UNIT: c_first = the column word of slot q0 when the caller has loaded it already (the closer
// wave requests it at its start, under <B,w>: the dot product of a unit C is then one gather away instead of a stream round
// trip plus a gather), have_first says so.
template <class F, bool UNIT>
__device__ __forceinline__ Fe sell_dot_range(const SellDev& M, const uint4* __restrict__ w, u32 q0, u32 q1, u32 lane, u32 c_first,
                                             bool have_first K2_TRACE_ARG) {
#ifdef ACX_K2_TRACE
    if (tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(q0), "s"(q1)); tr->t_ofs = k2_now(); }
#endif
    Fe acc = fe_zero();
    if (UNIT) {
        for (u32 q = q0; q < q1; ++q) {
            const u32 c = (have_first && q == q0) ? c_first : gload(&M.tail[(u64)q * kSlice + lane]).y;
            if (c != kNoRow) {
                const Fe x = fe_gload(w + 2 * (u64)c);
                acc = (q == q0) ? x : fe_add<F>(acc, x);    // rows are sorted: padding never precedes data
            }
        }
        return acc;
    }
    // Software pipeline, one slot deep.  Ablation on 2^22 rows (profiles/r01_r1cs_ablation.txt):
    // removing the multiplies saves 4 us of 345, removing the witness gathers 140, removing the
    // value stream 95 -- the kernel is bound by memory latency x concurrency (a divergent gather
    // costs one L1 tag lookup per lane), not by VALU or HBM bytes.  So: the gather of slot q is
    // issued first, then the stream loads of slot q+1, and only the gather is waited for (vmcnt
    // retires in order), which keeps a value-stream request in flight during every multiply.
    // Round 3 re-measured three levers on THIS kernel, same box, with counters (profiles/r03_r1cs.txt; the variants live in
    // git history at 8562ec3): the gather one slot ahead as well (92-100 VGPRs: 5 / 4 waves per SIMD: +3 / +6 %), the first
    // 1024 wires from an LDS copy (per-workgroup and persistent forms: +4 ... +9 %; L2 requests -17 %, time up), 36-byte
    // entries with a 4-byte unit-C stream (HBM bytes -9.3 %, time unchanged).  None is kept.
    static_assert(kSellMaxLen <= kWideTerms, "a SELL row is reduced once");
    if (q0 == q1) return acc;
    Wide wide;
    uint2 t = nt_load(&M.tail[(u64)q0 * kSlice + lane]);
    uint4 lo = nt_load(&M.val[(2 * (u64)q0) * kSlice + lane]);
    uint4 hi = nt_load(&M.val[(2 * (u64)q0 + 1) * kSlice + lane]);
#ifdef ACX_K2_TRACE
    if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::"v"(t.y), "v"(hi.w)); tr->t_first = k2_now(); }
#endif
    for (u32 q = q0; q < q1; ++q) {
        const uint4* px = w + 2 * (u64)(t.y == kNoRow ? 0u : t.y);   // padding: value 0 * w[0]
        const uint4 xlo = gload(px), xhi = gload(px + 1);
        Fe v;
        v.l[0] = lo.x; v.l[1] = lo.y; v.l[2] = lo.z; v.l[3] = lo.w;
        v.l[4] = hi.x; v.l[5] = hi.y; v.l[6] = hi.z; v.l[7] = hi.w;
        v.l[8] = t.x;
        if (q + 1 < q1) {
            t = nt_load(&M.tail[(u64)(q + 1) * kSlice + lane]);
            lo = nt_load(&M.val[(2 * (u64)(q + 1)) * kSlice + lane]);
            hi = nt_load(&M.val[(2 * (u64)(q + 1) + 1) * kSlice + lane]);
        }
        const u32 xw[8] = {xlo.x, xlo.y, xlo.z, xlo.w, xhi.x, xhi.y, xhi.z, xhi.w};
        const Fe x = fe_unpack(xw);
        if (q == q0) wide_mul(wide, v, x); else wide_mac(wide, v, x);
    }
    return wide_reduce<F>(wide);
}

template <class F, bool UNIT>
__device__ __forceinline__ Fe sell_dot(const SellDev& M, const uint4* __restrict__ w, u32 slice, u32 lane K2_TRACE_ARG) {
    const u32 q0 = sload(M.slice_ofs + slice), q1 = sload(M.slice_ofs + slice + 1);   // wave-uniform: scalar loads
    return sell_dot_range<F, UNIT>(M, w, q0, q1, lane, kNoRow, false K2_TRACE_PASS(tr));
}

// <M_row, w> for a small-coefficient matrix: nine signed columns, one v_mad_i64_i32 per limb and entry, one exact
// reduction per row (small_reduce).  The register budget the deferred-reduction path spends on its 17 64-bit columns
// pays here for a deeper pipeline: the column word of slot q+2 and the witness gather of slot q+1 are in flight while
// slot q is accumulated.
template <class F>
__device__ __forceinline__ Fe sell_dot_small(const SellDev& M, const uint4* __restrict__ w, u32 slice, u32 lane) {
    const u32 q0 = sload(M.slice_ofs + slice), q1 = sload(M.slice_ofs + slice + 1);   // wave-uniform: scalar loads
    static_assert(kSellMaxLen <= kWideTerms, "column bound of small_reduce");
    if (q0 == q1) return fe_zero();
    i64 acc[kLimbs];
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) acc[k] = 0;
    uint2 t = nt_load(&M.tail[(u64)q0 * kSlice + lane]);
    uint2 tn = t;
    if (q0 + 1 < q1) tn = nt_load(&M.tail[(u64)(q0 + 1) * kSlice + lane]);
    const uint4* px = w + 2 * (u64)(t.y == kNoRow ? 0u : t.y);
    uint4 xlo = gload(px), xhi = gload(px + 1);
    for (u32 q = q0; q < q1; ++q) {
        const i32 cf = t.y == kNoRow ? 0 : (i32)t.x;
        const u32 xw[8] = {xlo.x, xlo.y, xlo.z, xlo.w, xhi.x, xhi.y, xhi.z, xhi.w};
        if (q + 1 < q1) {
            t = tn;
            const uint4* pn = w + 2 * (u64)(t.y == kNoRow ? 0u : t.y);
            xlo = gload(pn);
            xhi = gload(pn + 1);
            if (q + 2 < q1) tn = nt_load(&M.tail[(u64)(q + 2) * kSlice + lane]);
        }
        const Fe x = fe_unpack(xw);
#pragma unroll
        for (int k = 0; k < kLimbs; ++k) acc[k] += (i64)cf * (i64)(i32)x.l[k];
    }
    return small_reduce<F>(acc);
}

struct ResidualOut {
    unsigned long long* result;  // {n_bad, first_bad}
    uint4* residuals;            // [n] or null
    uint4* dots;                 // [3 * stride] or null
    u64 dots_stride;
    u64 row_offset;
    // first_bad numbering.  map_log_r == 0: row + row_offset (a contiguous slab of a larger system).  Otherwise the rows are a
    // block-cyclic shard in ascending order (acx_mgpu_*, mgpu.inc.h): runs of 2^map_log_run consecutive rows, one run out
    // of every 2^map_log_r -- local row j is global row row_offset + (j mod run) + (j / run) * 2^map_log_r.  Only
    // evaluated on the (rare) violated-row path.
    u32 map_log_run, map_log_r;
    // null, or two dev elements {sa, sc}: the STORED <A_i,w> is multiplied by sa and the stored <C_i,w> by sc (the residual is
    // formed from the plain values).  h(x) lets 1/z and -1/z ride on the dots: (sa L) R + sc O = (L R - O) / z needs neither a
    // pointwise pass over the product nor a scaled subtraction afterwards.
    const uint4* dot_scale;
};

// per-lane epilogue shared by the SELL and the CSR-rows kernels.  Violations are the rare case: a
// wave without any skips the reduction entirely (one ballot); otherwise one atomic pair per wave.
template <class F>
__device__ __forceinline__ void residual_epilogue(const Fe& a, const Fe& b, const Fe& c, u32 row, bool live,
                                                  const ResidualOut& out) {
    bool bad = false;
    if (live) {
        const Fe r = fe_sub<F>(fe_mul<F>(a, b), c);
        bad = !fe_is_zero<F>(r);
        if (out.residuals != nullptr) fe_store(out.residuals + 2 * (u64)row, r);
        if (out.dots != nullptr) {
            const bool scaled = out.dot_scale != nullptr;          // uniform
            fe_store(out.dots + 2 * (u64)row, scaled ? fe_mul<F>(a, fe_load(out.dot_scale)) : a);
            fe_store(out.dots + 2 * (out.dots_stride + row), b);
            fe_store(out.dots + 2 * (2 * out.dots_stride + row), scaled ? fe_mul<F>(c, fe_load(out.dot_scale + 2)) : c);
        }
    }
    const unsigned long long mask = __ballot(bad);
    if (mask == 0) return;                                   // wave-uniform
    unsigned long long my_first = ~0ull;
    if (bad) {
        my_first = (u64)row + out.row_offset;
        if (out.map_log_r != 0)
            my_first = out.row_offset + (row & ((1u << out.map_log_run) - 1u)) + ((u64)(row >> out.map_log_run) << out.map_log_r);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(my_first, off, 64);
        my_first = o < my_first ? o : my_first;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out.result[0], (unsigned long long)__popcll(mask));
        atomicMin(&out.result[1], my_first);
    }
}

// One system of a (possibly batched) launch.
struct SellSystem {
    SellDev A, B, C;
    const u32* perm;       // [n_slices * 64] original row of each sorted position, kNoRow = none
    const uint4* w;        // witness, dev format (32 bytes per element)
    u32 n_slices;
    u32 unit_c;
    u32 small;             // bit k: matrix k (A, B, C) is stored in the small-coefficient form
    ResidualOut out;
};

// The system descriptor of a launch BY VALUE, in scalar registers: `systems` (batched launch) is read through the constant
// address space -- uniform s_load instructions -- or the kernel argument `one` is taken.  Binding a reference to
// `systems != nullptr ? systems[blockIdx.y] : one` instead makes every field access a flat load into vector registers, with
// the pointers living in VGPRs and every wait on them counted against both vmcnt and lgkmcnt: 138 against 117 us on 2^21 rows.
typedef __attribute__((address_space(4))) const unsigned long long c_u64;
static_assert(sizeof(SellSystem) % 8 == 0, "descriptor is copied in 8-byte words");
__device__ __forceinline__ SellSystem sell_system_of_launch(const SellSystem* systems, const SellSystem& one) {
    if (systems == nullptr) return one;
    SellSystem S;
    c_u64* src = (c_u64*)(unsigned long long)(systems + blockIdx.y);
    unsigned long long* dst = (unsigned long long*)&S;
#pragma unroll
    for (unsigned i = 0; i < sizeof(SellSystem) / 8; ++i) dst[i] = src[i];
    return S;
}

// K2: r_i = <A_i,w> * <B_i,w> - <C_i,w> for every row (verifyAssignment, src/QAP.hs:276-327, in the evaluation domain);
// blockIdx.y selects the system of a batched launch.  XCD-aware slice order: workgroup b runs on XCD b % 8 (observed
// dispatch order), so XCD x gets the contiguous slice range [x*T/8, (x+1)*T/8): the rows in flight on an XCD, and the
// witness window they gather from, stay inside its private 4 MiB L2.  gridDim.x = slices rounded up to a multiple of 8.
// SPEC = 0: no matrix of the launch is in the small-coefficient form (the full-width path alone).
// SPEC = 1: every system of the launch has small-coefficient A and B and a unit C (the shape of a compiled program);
//           the instance then carries none of the deferred-reduction path's registers (58 VGPRs, 8 waves per SIMD).
// SPEC = 2: anything else; the form of each matrix is a run-time flag of its system.
// (Rounds 1-2 also carried `k_r1cs_sell`, one wave walking A, B, C of four slices per workgroup: 120.3 us against 114.7 us
// per bench launch, profiles/r02_r1cs_experiments.txt item 15; removed in round 3, code in git history.)
// K2, wave-specialised form: a slice takes TWO waves.  Wave 0 forms <A,w>, parks it in LDS and leaves; wave 1 forms <B,w>
// (which waits in LDS meanwhile: 77 VGPRs = 6 waves per SIMD), then <C,w> -- for the unit C of every gate the reference
// emits that is gathers and additions only -- and does the closing a*b - c test.  Against one wave walking A, B, C in turn
// (the removed k_r1cs_sell): a slice's streams are in flight together, waves of different instruction mix share every SIMD, and a
// launch too small to fill the chip (configs[1] taken literally: 2^16 rows = 1024 slices) costs less than the sum of three
// dot products' latencies.  The closer is the wave with the LONGER job, so it never sits in the barrier holding a wave slot
// (the first form of this kernel, three waves with the closing test on the wave that finishes first, lost 15 % to that).
// Same box, alternating processes (tools/k2_ab.py, bench workload, us per launch): one wave per slice 120.3, three waves
// (A + closing | B | C) 118.1, two waves (A + closing | B, C) 115.2, this form 114.7.
template <class F, int SPEC = 0>
__global__ __launch_bounds__(2 * kSlice) void k_r1cs_sell_split(const SellSystem* __restrict__ systems, SellSystem one) {
#ifdef ACX_K2_TRACE
    K2Trace trace{k2_now(), 0, 0, 0, 0};
    K2Trace* tr = &trace;
#endif
    const SellSystem S = sell_system_of_launch(systems, one);               // batched : single
    const u32 per_xcd = (S.n_slices + 7) / 8;
    const u32 slice = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (blockIdx.x >= 8 * per_xcd || slice >= S.n_slices) return;          // uniform over the workgroup
#ifdef ACX_K2_TRACE
    trace.t_desc = k2_now();
#endif
    const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kSlice), lane = threadIdx.x % kSlice;
    constexpr bool kMixed = SPEC == 2;
    __shared__ u32 park[2][kLimbs][kSlice];
    Fe b = fe_zero(), c = b;
    u32 row = kNoRow;
    if (wv == 0) {
        const Fe a = (SPEC == 1 || (kMixed && (S.small & 1u))) ? sell_dot_small<F>(S.A, S.w, slice, lane) : sell_dot<F, false>(S.A, S.w, slice, lane K2_TRACE_PASS(tr));
#ifdef ACX_K2_TRACE
        asm volatile("" ::"v"(a.l[0]));
        trace.t_dot = k2_now();
#endif
#pragma unroll
        for (int i = 0; i < kLimbs; ++i) park[0][i][lane] = a.l[i];
    } else {
        row = gload(S.perm + slice * kSlice + lane);                        // needed last: issued first
        // A unit C (every gate the reference emits) is one column word and one gather per entry.  Its slot range and the
        // column word of its first slot are requested NOW, under <B,w>: a wave's life is a chain of ~3000-cycle memory
        // round trips (tools/k2_trace.py: 20.4k cycles for this wave, of which the C phase's own offsets + column word +
        // gather were ~5.5k), and this takes two of them out of the chain for one more register.
        const bool unit_c = SPEC == 1 || S.unit_c;
        u32 qc0 = 0, qc1 = 0, c_first = kNoRow;
        if (unit_c) {
            qc0 = sload(S.C.slice_ofs + slice); qc1 = sload(S.C.slice_ofs + slice + 1);
            if (qc0 < qc1) c_first = gload(&S.C.tail[(u64)qc0 * kSlice + lane]).y;
        }
        b = (SPEC == 1 || (kMixed && (S.small & 2u))) ? sell_dot_small<F>(S.B, S.w, slice, lane) : sell_dot<F, false>(S.B, S.w, slice, lane K2_TRACE_PASS(tr));
#ifdef ACX_K2_TRACE
        asm volatile("" ::"v"(b.l[0]));
        trace.t_dot = k2_now();
#endif
#pragma unroll
        for (int i = 0; i < kLimbs; ++i) park[1][i][lane] = b.l[i];         // own wave's LDS traffic is ordered: no barrier
        c = unit_c ? sell_dot_range<F, true>(S.C, S.w, qc0, qc1, lane, c_first, true K2_TRACE_PASS(nullptr))
            : (kMixed && (S.small & 4u)) ? sell_dot_small<F>(S.C, S.w, slice, lane) : sell_dot<F, false>(S.C, S.w, slice, lane K2_TRACE_PASS(nullptr));
    }
#ifdef ACX_K2_TRACE
    auto k2_flush = [&](u32 role) {
        if (lane == 0) {
            const u64 t_end = k2_now();
            unsigned long long* rec = g_k2_trace[role][(blockIdx.y * gridDim.x + blockIdx.x) & (kK2TraceWaves - 1)];
            rec[0] = 1ull;
            rec[1] = trace.t_desc - trace.t0;
            rec[2] = trace.t_ofs - trace.t0;
            rec[3] = trace.t_first - trace.t0;
            rec[4] = trace.t_dot - trace.t0;
            rec[5] = t_end - trace.t0;
        }
    };
    if (wv == 0) k2_flush(0);
#endif
    __syncthreads();
    if (wv == 0) return;
    Fe a;
#pragma unroll
    for (int i = 0; i < kLimbs; ++i) { a.l[i] = park[0][i][lane]; b.l[i] = park[1][i][lane]; }
    residual_epilogue<F>(a, b, c, row, row != kNoRow, S.out);
#ifdef ACX_K2_TRACE
    k2_flush(1);
#endif
}

// CSR path for the listed rows only (rows too long for the SELL layout: the 2^j row of a Split gate has 257 entries,
// src/QAP.hs:447-459).  ONE WAVE PER ROW: lane l takes entries e0 + l, e0 + l + 64, ... (coalesced column / value loads),
// sums up to kWideTerms raw products per deferred reduction, and the 64 partial sums are folded with xor-shuffles.
// (One LANE per row -- the first version -- walked 257 dependent gathers serially: 150 us for the 87 long rows of a
// 6000-gate circuit in the reference's 50:10:1 gate mix, 95 % of its verification.)
template <class F, bool UNIT>
__device__ __forceinline__ Fe long_row_dot(const CsrDev& M, const uint4* __restrict__ w, u32 row, u32 sub, u32 G) {
    const u32 e0 = M.rowptr[row], e1 = M.rowptr[row + 1];
    Fe acc = fe_zero();
    bool any = false;
    for (u32 base = e0 + sub; base < e1; base += G * kWideTerms) {
        Fe part = fe_zero();
        if (UNIT) {
            bool first = true;
#pragma unroll 1
            for (int j = 0; j < kWideTerms; ++j) {
                const u32 e = base + G * j;
                if (e < e1) {
                    const Fe x = fe_load(w + 2 * (u64)M.col[e]);
                    part = first ? x : fe_add<F>(part, x);
                    first = false;
                }
            }
        } else {
            Wide wide;
            wide_zero(wide);
#pragma unroll 1
            for (int j = 0; j < kWideTerms; ++j) {
                const u32 e = base + G * j;
                if (e < e1) wide_mac(wide, fe_load(M.val + 2 * (u64)e), fe_load(w + 2 * (u64)M.col[e]));
            }
            part = wide_reduce<F>(wide);
        }
        acc = any ? fe_add<F>(acc, part) : part;
        any = true;
    }
#pragma unroll 1
    for (int off = (int)G / 2; off > 0; off >>= 1) {              // fold the G partial sums of the row's lane group
        Fe o;
#pragma unroll
        for (int k = 0; k < kLimbs; ++k) o.l[k] = (u32)__shfl_xor((int)acc.l[k], off, kSlice);
        acc = fe_add<F>(acc, o);
    }
    return acc;                                                   // every lane of the group holds the row's dot product
}

// G lanes per row (a power of two, 2 .. 64; one launch per tier of row lengths): rows of 7 .. 12 entries -- affine sides
// that are sums of several wires -- take two lanes each and cost about what a SELL row costs, 13 .. 24 four, up to 48 eight;
// longer rows eight lanes with several reductions per lane when there are many of them (throughput), a whole wave each when
// there are few (the Split gates of a circuit: latency).
template <class F, bool UNIT_C>
__global__ __launch_bounds__(kBlock) void k_r1cs_residual_rows(CsrDev A, CsrDev B, CsrDev C,
                                                              const uint4* __restrict__ w,
                                                              const u32* __restrict__ rows, u32 n_rows, u32 G,
                                                              ResidualOut out, const SellSystem* __restrict__ many) {
    if (many != nullptr) {                                       // acx_r1cs_verify_many: blockIdx.y = witness of the same system
        w = many[blockIdx.y].w;
        out = many[blockIdx.y].out;
    }
    const u32 i = (blockIdx.x * kBlock + threadIdx.x) / G, sub = threadIdx.x % G;
    const bool have = i < n_rows;
    // groups past the end run on the last row (uniform control flow for the shuffles) and report nothing
    const u32 row = rows[have ? i : n_rows - 1];
    const Fe a = long_row_dot<F, false>(A, w, row, sub, G);
    const Fe b = long_row_dot<F, false>(B, w, row, sub, G);
    const Fe c = long_row_dot<F, UNIT_C>(C, w, row, sub, G);
    residual_epilogue<F>(a, b, c, row, have && sub == 0, out);
}

// ---------------------------------------------------------------------------------------------
// NTT (replaces galois-fft `FFT.fft` / `FFT.interpolate`; call sites src/QAP.hs:521-524).

// tw[j] = base^j for j < count
template <class F>
__global__ __launch_bounds__(kBlock) void k_pow_table(uint4* __restrict__ tw, u64 count, FeArg base_arg) {
    const Fe base = fe_from_arg(base_arg);
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < count; j += (u64)gridDim.x * kBlock)
        fe_store(tw + 2 * j, fe_pow<F>(base, j));
}

// tw[j] = first * base^j for j < count (inter-pass twiddles with the 1/N of an inverse transform folded in)
template <class F>
__global__ __launch_bounds__(kBlock) void k_pow_table_scaled(uint4* __restrict__ tw, u64 count, FeArg base_arg, FeArg first_arg) {
    const Fe base = fe_from_arg(base_arg), first = fe_from_arg(first_arg);
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < count; j += (u64)gridDim.x * kBlock)
        fe_store(tw + 2 * j, fe_mul<F>(fe_pow<F>(base, j), first));
}

// limb-form table for k_ntt_r4: entry j = 3 x uint4 holding the nine 29-bit limbs of base^j (no unpacking in the kernel)
template <class F>
__global__ __launch_bounds__(kBlock) void k_pow_table_limbs(uint4* __restrict__ tw, u64 count, FeArg base_arg) {
    const Fe base = fe_from_arg(base_arg);
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < count; j += (u64)gridDim.x * kBlock) {
        const Fe w = fe_pow<F>(base, j);
        tw[kLimbEntryQuads * j] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]);
        tw[kLimbEntryQuads * j + 1] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]);
        tw[kLimbEntryQuads * j + 2] = make_uint4(w.l[8], 0u, 0u, 0u);
    }
}

// Closing-factor table of one local step of the distributed four-step transform, in the step's STORE order (k_ntt_r4
// tw_mode 3): out[off] = first * w^(e1(off)) * g^(e2(off)), both powers from two-level tables (null = factor absent).
//   XCHG layout (step 0: the twiddle w_N^(+-i2 k1), and g^i2 of a forward coset transform):
//       off = (peer * rw + kl) * cw + i2l;  forward: k1 = peer * rw + kl, i2 = rank * cw + i2l;  inverse: k1 = rank * rw + kl, i2 = peer * cw + i2l
//   COLS layout (inverse step 1: the coset factor g^-(i1 C + i2)):  off = i2l * R + i1, i2 = rank * cw + i2l
struct DistTable {
    const uint4 *w_lo, *w_hi;      // w_N^(+-j), j < 1024 (with 1/N folded in for an inverse transform) and w_N^(+-1024 j); null: no twiddle
    const uint4 *g_lo, *g_hi;      // g^j, g^(1024 j) (or powers of 1/g); null: no coset factor
    u32 log_n, log_r, log_w, rank;
    u32 inverse, cols_layout;
};
template <class F>
__global__ __launch_bounds__(kBlock) void k_dist_table(DistTable T, uint4* __restrict__ out, u64 count) {
    const u32 log_c = T.log_n - T.log_r, log_rw = T.log_r - T.log_w, log_cw = log_c - T.log_w;
    const u64 mask = (1ull << T.log_n) - 1;
    for (u64 off = (u64)blockIdx.x * kBlock + threadIdx.x; off < count; off += (u64)gridDim.x * kBlock) {
        u64 e_w = 0, e_g = 0;
        if (T.cols_layout) {
            const u64 i1 = off & ((1ull << T.log_r) - 1), i2 = ((u64)T.rank << log_cw) + (off >> T.log_r);
            e_g = (i1 << log_c) + i2;
        } else {
            const u64 i2l = off & ((1ull << log_cw) - 1), kl = (off >> log_cw) & ((1ull << log_rw) - 1), peer = off >> (log_cw + log_rw);
            const u64 k1 = T.inverse ? (((u64)T.rank << log_rw) + kl) : ((peer << log_rw) + kl);
            const u64 i2 = T.inverse ? ((peer << log_cw) + i2l) : (((u64)T.rank << log_cw) + i2l);
            e_w = (i2 * k1) & mask;
            e_g = i2;
        }
        Fe f;
        bool have = false;
        if (T.w_lo != nullptr) { f = two_level_pow<F>(T.w_lo, T.w_hi, e_w); have = true; }
        if (T.g_lo != nullptr) {
            const Fe g = two_level_pow<F>(T.g_lo, T.g_hi, e_g);
            f = have ? fe_mul<F>(f, g) : g;
            have = true;
        }
        fe_store(out + 2 * off, have ? f : fe_one_mont<F>());
    }
}

// ---- K3/K4: tiled multi-pass NTT (pass descriptor and planning constants: ntt_pass.hip.h) ----------
template <class F>
__global__ __launch_bounds__(kBlock) void k_ntt_tile(NttPass P) {
    __shared__ u32 lds[kLimbs][kTileElems];
    const u32 S = 1u << P.log_s, T = 1u << P.log_t, elems = S * T;
    // tile -> outer indices
    u64 base_in = 0, base_out = 0, K0 = 0, I0 = 0;
    {
        u64 t = blockIdx.x;
        for (u32 d = 0; d < P.n_outer; ++d) {
            const u64 idx = t % P.outer[d].count;
            t /= P.outer[d].count;
            base_in += idx * P.outer[d].stride_in;
            base_out += idx * P.outer[d].stride_out;
            K0 += idx * P.outer[d].k_w;
            I0 += idx * P.outer[d].i_w;
        }
    }
    // load: element (point d, column c), placed at bit-reversed point position
    for (u32 e = threadIdx.x; e < elems; e += kBlock) {
        const u32 c = e & (T - 1), d = e >> P.log_t;
        const u64 off = base_in + (u64)d * P.stride_t_in + (u64)c * P.stride_c_in;
        Fe x = fe_load(P.src + 2 * off);
        if (P.scale_on_load) x = fe_mul<F>(x, two_level_pow<F>(P.sc_lo, P.sc_hi, off & P.idx_mask));
        const u32 pos = ((P.log_s ? (__brev(d) >> (32 - P.log_s)) : 0u) << P.log_t) | c;
#pragma unroll
        for (int k = 0; k < kLimbs; ++k) lds[k][pos] = x.l[k];
    }
    __syncthreads();
    // radix-2 DIT stages
    for (u32 lh = 0; lh < P.log_s; ++lh) {
        const u32 h = 1u << lh;
        for (u32 b = threadIdx.x; b < elems / 2; b += kBlock) {
            const u32 c = b & (T - 1), q = b >> P.log_t;
            const u32 j = q & (h - 1), grp = q >> lh;
            const u32 i0 = (((grp << (lh + 1)) + j) << P.log_t) | c, i1 = i0 + (h << P.log_t);
            Fe u, v;
#pragma unroll
            for (int k = 0; k < kLimbs; ++k) { u.l[k] = lds[k][i0]; v.l[k] = lds[k][i1]; }
            Fe t;
            if (lh == 0) {
                // w_2^0 = 1: no multiplication, but v must become a strict product-like value < 2p:
                // after load every value is strict and < 2p, so it already is.
                t = v;
            } else {
                t = fe_mul<F>(v, fe_load(P.sub_tw + 2 * (u64)(j << (P.log_s - 1 - lh))));
            }
            const Fe a = fe_add_lazy(u, t), s = fe_sub_lazy<F>(u, t);
#pragma unroll
            for (int k = 0; k < kLimbs; ++k) { lds[k][i0] = a.l[k]; lds[k][i1] = s.l[k]; }
        }
        __syncthreads();
    }
    // store with the reducing multiplication
    const Fe scale = fe_from_arg(P.scale);
    for (u32 e = threadIdx.x; e < elems; e += kBlock) {
        const u32 c = e & (T - 1), d = e >> P.log_t;      // d = output digit k_p
        Fe x;
#pragma unroll
        for (int k = 0; k < kLimbs; ++k) x.l[k] = lds[k][e];
        const u64 off = base_out + (u64)d * P.stride_t_out + (u64)c * P.stride_c_out;
        Fe f;
        if (P.tw_mode != 0) {
            const u64 K = K0 + (u64)d * P.t_kw + (u64)c * P.c_kw, I = I0 + (u64)c * P.c_iw;
            const u64 E = I * K;
            f = (P.tw_mode == 1) ? fe_load(P.tw_lo + 2 * (E >> P.tw_shift))
                                 : two_level_pow<F>(P.tw_lo, P.tw_hi, E & P.tw_mask);
        } else if (P.scale_mode == 2) {
            f = fe_mul<F>(scale, two_level_pow<F>(P.sc_lo, P.sc_hi, off & P.idx_mask));
        } else {
            f = scale;
        }
        fe_store(P.dst + 2 * off, fe_mul<F>(x, f));
    }
}

// ---------------------------------------------------------------------------------------------
// K5: h on the coset: out[i] = (a[i]*b[i] - c[i]) * zinv   (src/QAP.hs:325-327 in evaluation form).  c == nullptr:
// out[i] = a[i]*b[i]*zinv -- the pipeline then subtracts zinv * O(x) in the COEFFICIENT domain after the inverse coset
// transform (the transform is linear and coset-NTT followed by inverse-coset-NTT is the identity on O's coefficients, so
// O never needs its coset evaluations: six transforms per h(x) instead of seven).
template <class F>
__global__ __launch_bounds__(kBlock) void k_pointwise_h(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                       const uint4* __restrict__ c, uint4* __restrict__ out, u64 n,
                                                       FeArg zinv_arg, u32 zero_top) {
    const Fe zinv = fe_from_arg(zinv_arg);
    if (zero_top && blockIdx.x == 0 && threadIdx.x == 0) fe_store(out + 2 * n, fe_zero());   // h has N+1 coefficients; the
                                                                                             // transform that follows leaves it alone
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock) {
        Fe t = fe_mul<F>(fe_load(a + 2 * i), fe_load(b + 2 * i));
        if (c != nullptr) t = fe_sub<F>(t, fe_load(c + 2 * i));
        fe_store(out + 2 * i, fe_mul<F>(t, zinv));
    }
}

// the two scalar corrections of the zero-knowledge quotient: h[0] -= sub0, h[top_index] = top (top_index = ~0: the caller
// appends the top coefficient itself -- the sharded pipeline, whose coefficient N lies outside every shard's block)
template <class F>
__global__ void k_h_fix(uint4* __restrict__ h, u64 top_index, FeArg sub0, FeArg top) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        fe_store(h, fe_sub<F>(fe_load(h), fe_from_arg(sub0)));
        if (top_index != ~0ull) fe_store(h + 2 * top_index, fe_from_arg(top));
    }
}

// h += ax * x + ay * y + az * z elementwise; null vectors are skipped.  The zero-knowledge shift d1 * R0 + d2 * L0
// (src/QAP.hs:315-323) and the coefficient-domain subtraction of zinv * O0 (k_pointwise_h) in one pass.
template <class F>
__global__ __launch_bounds__(kBlock) void k_axpy3(uint4* __restrict__ h, const uint4* __restrict__ x, const uint4* __restrict__ y,
                                                 const uint4* __restrict__ z, u64 n, FeArg ax_arg, FeArg ay_arg, FeArg az_arg) {
    const Fe ax = fe_from_arg(ax_arg), ay = fe_from_arg(ay_arg), az = fe_from_arg(az_arg);
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock) {
        Fe t = fe_load(h + 2 * i);
        if (x != nullptr) t = fe_add<F>(t, fe_mul<F>(ax, fe_load(x + 2 * i)));
        if (y != nullptr) t = fe_add<F>(t, fe_mul<F>(ay, fe_load(y + 2 * i)));
        if (z != nullptr) t = fe_add<F>(t, fe_mul<F>(az, fe_load(z + 2 * i)));
        fe_store(h + 2 * i, t);
    }
}

// h[i] += scale * base^i * x[i] with base^i from the two-level table lo[i & 1023] * hi[i >> 10] (hi == nullptr: lo[i]).
// The h(x) pipeline's subtraction of O / z when O's coefficients still carry the coset factor g^i of the fused inverse
// transform (base = 1/g, scale = -1/z): all three inverse transforms stay one batched launch.
template <class F>
__global__ __launch_bounds__(kBlock) void k_axpy_geo(uint4* __restrict__ h, const uint4* __restrict__ x, u64 n,
                                                    const uint4* __restrict__ lo, const uint4* __restrict__ hi, FeArg scale_arg) {
    const Fe scale = fe_from_arg(scale_arg);
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock) {
        Fe f = fe_mul<F>(scale, fe_load(lo + 2 * (i & 1023u)));
        if (hi != nullptr) f = fe_mul<F>(f, fe_load(hi + 2 * (i >> 10)));
        fe_store(h + 2 * i, fe_add<F>(fe_load(h + 2 * i), fe_mul<F>(f, fe_load(x + 2 * i))));
    }
}

// ---------------------------------------------------------------------------------------------
// Naive-roots path: `createPolynomials` (src/QAP.hs:486-508) = Lagrange interpolation on ARBITRARY
// distinct roots with target T(x) = prod (x - r_i).  The reference calls its own version "terrible
// complexity" and uses it at test sizes only (roots 7,8,9 in test/Test/QAP.hs:73); these kernels
// are plain O(n^2) and are not tuned.  n <= 4096.
//
// T coefficients, low to high, n + 1 of them (monic).  One workgroup; n sequential steps.
template <class F>
__global__ __launch_bounds__(1024) void k_poly_from_roots(const uint4* __restrict__ roots, u32 n,
                                                         uint4* __restrict__ coef, uint4* __restrict__ tmp) {
    // coef holds the running product of degree d (d+1 coefficients); multiply by (x - r_d)
    for (u32 i = threadIdx.x; i <= n; i += blockDim.x) fe_store(coef + 2 * (u64)i, i == 0 ? fe_one_mont<F>() : fe_zero());
    __syncthreads();
    for (u32 d = 0; d < n; ++d) {
        const Fe r = fe_load(roots + 2 * (u64)d);
        for (u32 i = threadIdx.x; i <= d + 1; i += blockDim.x) {
            // new[i] = old[i-1] - r * old[i]
            const Fe lo = (i >= 1) ? fe_load(coef + 2 * (u64)(i - 1)) : fe_zero();
            const Fe hi = (i <= d) ? fe_mul<F>(r, fe_load(coef + 2 * (u64)i)) : fe_zero();
            fe_store(tmp + 2 * (u64)i, fe_sub<F>(lo, hi));
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i <= d + 1; i += blockDim.x) {
            coef[2 * (u64)i] = tmp[2 * (u64)i];
            coef[2 * (u64)i + 1] = tmp[2 * (u64)i + 1];
        }
        __syncthreads();
    }
}

// a^(p-2): exponent given as 8 x u32 words (host-computed p - 2)
struct Exp256 { u32 w[8]; };
template <class F>
__device__ __forceinline__ Fe fe_inv_exp(const Fe& a, const Exp256& e) {
    Fe acc = fe_one_mont<F>(), base = a;
    for (int i = 0; i < 256; ++i) {
        if ((e.w[i >> 5] >> (i & 31)) & 1) acc = fe_mul<F>(acc, base);
        base = fe_mul<F>(base, base);
    }
    return acc;
}

// winv[i] = 1 / T'(r_i) = 1 / prod_{j != i} (r_i - r_j)   (`phis`, src/QAP.hs:503-504)
template <class F>
__global__ __launch_bounds__(kBlock) void k_bary_inv(const uint4* __restrict__ roots, u32 n, uint4* __restrict__ winv,
                                                    Exp256 pm2) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Fe ri = fe_load(roots + 2 * (u64)i);
    Fe acc = fe_one_mont<F>();
    for (u32 j = 0; j < n; ++j)
        if (j != i) acc = fe_mul<F>(acc, fe_sub<F>(ri, fe_load(roots + 2 * (u64)j)));
    fe_store(winv + 2 * (u64)i, fe_inv_exp<F>(acc, pm2));
}

// Q[i][k] = coefficient k of  winv[i] * T(x) / (x - r_i)   (synthetic division; `roots `quot` root x`
// scaled by 1/phi, src/QAP.hs:496-500), k < n.  One thread per i.
template <class F>
__global__ __launch_bounds__(kBlock) void k_build_q(const uint4* __restrict__ roots, const uint4* __restrict__ tcoef,
                                                   const uint4* __restrict__ winv, u32 n, uint4* __restrict__ Q) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Fe ri = fe_load(roots + 2 * (u64)i), wi = fe_load(winv + 2 * (u64)i);
    Fe q = fe_load(tcoef + 2 * (u64)n);                      // leading coefficient (1)
    for (u32 k = n; k-- > 0;) {                               // q_k = T[k+1] + r_i * q_{k+1}
        fe_store(Q + 2 * ((u64)i * n + k), fe_mul<F>(q, wi));
        q = fe_add<F>(fe_load(tcoef + 2 * (u64)k), fe_mul<F>(ri, q));
    }
}

// out[b][k] = sum_i vals[b * val_stride + i] * Q[i][k]      (Lagrange sum, src/QAP.hs:496-500)
template <class F>
__global__ __launch_bounds__(kBlock) void k_matvec_q(const uint4* __restrict__ vals, u64 val_stride,
                                                    const uint4* __restrict__ Q, u32 n, u64 batch,
                                                    uint4* __restrict__ out, u64 out_stride) {
    for (u64 t = (u64)blockIdx.x * kBlock + threadIdx.x; t < batch * n; t += (u64)gridDim.x * kBlock) {
        const u64 b = t / n;
        const u32 k = (u32)(t - b * n);
        Fe acc = fe_zero();
        for (u32 i = 0; i < n; ++i) {
            const Fe v = fe_load(vals + 2 * (b * val_stride + i));
            if (fe_is_zero<F>(v)) continue;                    // columns are sparse
            acc = fe_add<F>(acc, fe_mul<F>(v, fe_load(Q + 2 * ((u64)i * n + k))));
        }
        fe_store(out + 2 * (b * out_stride + k), acc);
    }
}

// c[k] = sum_{i+j=k} a[i] * b[j], a: na coefficients, b: nb  (schoolbook; poly `*`, src/QAP.hs:325)
template <class F>
__global__ __launch_bounds__(kBlock) void k_poly_mul(const uint4* __restrict__ a, u32 na, const uint4* __restrict__ b,
                                                    u32 nb, uint4* __restrict__ c) {
    const u32 k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= na + nb - 1) return;
    Fe acc = fe_zero();
    const u32 i0 = k >= nb ? k - nb + 1 : 0, i1 = k < na ? k : na - 1;
    for (u32 i = i0; i <= i1; ++i) acc = fe_add<F>(acc, fe_mul<F>(fe_load(a + 2 * (u64)i), fe_load(b + 2 * (u64)(k - i))));
    fe_store(c + 2 * (u64)k, acc);
}

// y[i] = y[i] * sy + x[i] * sx   (sy, sx constants; used for L = L0 + delta*T and P = L*R - O)
template <class F>
__global__ __launch_bounds__(kBlock) void k_poly_axpby(uint4* __restrict__ y, const uint4* __restrict__ x, u32 n,
                                                      FeArg sy_arg, FeArg sx_arg) {
    const Fe sy = fe_from_arg(sy_arg), sx = fe_from_arg(sx_arg);
    for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const Fe t = fe_add<F>(fe_mul<F>(fe_load(y + 2 * (u64)i), sy), fe_mul<F>(fe_load(x + 2 * (u64)i), sx));
        fe_store(y + 2 * (u64)i, t);
    }
}

// Long division of P (np coefficients, destroyed: becomes the remainder in its low n coefficients)
// by the MONIC T of degree n (`quotRem`, src/QAP.hs:327).  quot gets np - n coefficients.
// One workgroup; np - n sequential steps.
template <class F>
__global__ __launch_bounds__(1024) void k_poly_divrem_monic(uint4* __restrict__ P, u32 np, const uint4* __restrict__ T,
                                                           u32 n, uint4* __restrict__ quot) {
    if (np <= n) return;
    for (u32 s = np - n; s-- > 0;) {               // quotient coefficient s = P[s + n]
        const Fe q = fe_load(P + 2 * (u64)(s + n));
        if (threadIdx.x == 0) fe_store(quot + 2 * (u64)s, q);
        for (u32 j = threadIdx.x; j < n; j += blockDim.x) {
            const Fe t = fe_sub<F>(fe_load(P + 2 * (u64)(s + j)), fe_mul<F>(q, fe_load(T + 2 * (u64)j)));
            fe_store(P + 2 * (u64)(s + j), t);
        }
        __syncthreads();
    }
}

// flag[0] |= 1 if any of the n elements is nonzero mod p
template <class F>
__global__ __launch_bounds__(kBlock) void k_any_nonzero(const uint4* __restrict__ x, u32 n, u32* __restrict__ flag) {
    for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        if (!fe_is_zero<F>(fe_load(x + 2 * (u64)i))) atomicOr(flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// Witness generation on the device: `evalArithCircuit` = foldl' evalGate (src/Circuit/Arithmetic.hs:
// 106-145,221-235) restructured by dependency LEVEL: every gate of a level only reads wires written
// by earlier levels, so a level is one data-parallel launch.  Mul: out = <A_row,w> * <B_row,w>, the
// gate's own constraint row (src/QAP.hs:371-395); Equal: out = (inp /= 0), magic = inp^-1 (Fermat);
// Split: bit j of the canonical integer.
constexpr u32 kEvalLanes = 8;   // lanes per gate in k_eval_level_lanes
struct EvalGates {
    const u32* items;        // gate ids of this level
    u32 count;
    const uint8_t* kind;     // per gate
    const u32* row;          // per gate: constraint row of a Mul gate in the stored row order
    const u32* wire_ofs;     // per gate: offset into wires (n_gates + 1)
    const u32* wires;        // flat wire indices: Mul {out}, Equal {i, m, out}, Split {inp, outs...}
    const uint4* mul;        // per item, level order: Mul gate {out wire, first A entry, first B entry, nA | nB << 16}, else .w = ~0
    const u32* cols;         // per item, level order: kEvalLanes columns (entries 0-3 of the A row, 0-3 of the B row; k_eval_fill_cols)
    u32 defer_magic;         // Equal gates leave their magic wire to k_eval_magic (no gate reads one: HostCircuit::build_plan)
};

// one gate of any kind on one lane (everything except the recorded Mul gates of a level)
template <class F>
__device__ __forceinline__ void eval_gate_generic(const EvalGates& G, const CsrDev& A, const CsrDev& B, uint4* __restrict__ w, u32 g) {
    const u32* gw = G.wires + G.wire_ofs[g];
    const u32 kd = G.kind[g];
    if (kd == 0) {                                            // Mul
        const u32 row = G.row[g];
        const Fe a = csr_row_dot<F, false>(A, w, row), b = csr_row_dot<F, false>(B, w, row);
        fe_store(w + 2 * (u64)gw[0], fe_mul<F>(a, b));
    } else if (kd == 1) {                                     // Equal
        const Fe inp = fe_load(w + 2 * (u64)gw[0]);
        const bool z = fe_is_zero<F>(inp);
        fe_store(w + 2 * (u64)gw[2], z ? fe_zero() : fe_one_mont<F>());
        if (!G.defer_magic) fe_store(w + 2 * (u64)gw[1], z ? fe_zero() : fe_inv_divsteps<F>(inp));
    } else {                                                  // Split
        const Fe c = fe_from_mont<F>(fe_load(w + 2 * (u64)gw[0]));
        const u32 n_out = G.wire_ofs[g + 1] - G.wire_ofs[g] - 1;
        for (u32 j = 0; j < n_out; ++j) {
            const bool bit = j < 256 && ((c.l[j / kLimbBits] >> (j % kLimbBits)) & 1u);
            fe_store(w + 2 * (u64)gw[1 + j], bit ? fe_one_mont<F>() : fe_zero());
        }
    }
}

// The magic wires of ALL Equal gates (magic = inp^-1, 0 for inp = 0; src/Circuit/Arithmetic.hs:117-131), one lane per gate,
// after the last level: an inversion is ~20 000 dependent instructions, and inside the levels it would be the latency of
// every level that holds an Equal gate (the gate's OUTPUT, the only thing later gates may read, is a zero test).
template <class F>
__global__ __launch_bounds__(kSlice) void k_eval_magic(const u32* __restrict__ gates, u32 count, const u32* __restrict__ wire_ofs,
                                                       const u32* __restrict__ wires, uint4* __restrict__ w) {
    const u32 t = blockIdx.x * kSlice + threadIdx.x;
    if (t >= count) return;
    const u32* gw = wires + wire_ofs[gates[t]];
    const Fe inp = fe_load(w + 2 * (u64)gw[0]);
    fe_store(w + 2 * (u64)gw[1], fe_is_zero<F>(inp) ? fe_zero() : fe_inv_divsteps<F>(inp));
}

// A Split gate on the kEvalLanes lanes of its group (k_eval_level_lanes): lane `sub` writes output bits [32 c, 32 c + 32) for
// c = sub, sub + kEvalLanes, ... -- one word of the packed canonical value each.  On one lane the 256 stores (and their wire
// lookups) were ~50 us of the level's latency; bits past 255 are zero (a canonical value is below 2^256).
template <class F>
__device__ __forceinline__ void eval_split_lanes(const EvalGates& G, uint4* __restrict__ w, u32 g, u32 sub) {
    const u32* gw = G.wires + G.wire_ofs[g];
    const u32 n_out = G.wire_ofs[g + 1] - G.wire_ofs[g] - 1;
    u32 words[8], one[8];
    fe_pack(fe_from_mont<F>(fe_load(w + 2 * (u64)gw[0])), words);
    fe_pack(fe_one_mont<F>(), one);
#pragma unroll 1
    for (u32 base = 32u * sub; base < n_out; base += 32u * kEvalLanes) {
        u32 wd = 0;
#pragma unroll
        for (u32 q = 0; q < 8; ++q) wd = (base >> 5) == q ? words[q] : wd;
        const u32 end = min(32u, n_out - base);
#pragma unroll 4
        for (u32 i = 0; i < end; ++i) {
            const u32 m = 0u - ((wd >> i) & 1u);
            v4u32 lo, hi;
            lo.x = one[0] & m; lo.y = one[1] & m; lo.z = one[2] & m; lo.w = one[3] & m;
            hi.x = one[4] & m; hi.y = one[5] & m; hi.z = one[6] & m; hi.w = one[7] & m;
            uint4* p = w + 2 * (u64)gw[1 + base + i];
            *(g_v4u32_t*)p = lo;
            *(g_v4u32_t*)(p + 1) = hi;
        }
    }
}

// One lane per gate: the form for WIDE levels (throughput: deferred reduction, 81 multiplier instructions per entry).
template <class F>
__global__ __launch_bounds__(kBlock) void k_eval_level(EvalGates G, CsrDev A, CsrDev B, uint4* __restrict__ w) {
    const u32 t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= G.count) return;
    // Mul gates (nearly all of a circuit) carry everything in ONE level-ordered record: item -> gate -> row -> row
    // pointers -> entries -> witness becomes record -> entries -> witness
    const uint4 it = gload(G.mul + t);
    if (it.w != 0xffffffffu) {
        const Fe a = csr_range_dot<F, false>(A, w, it.y, it.y + (it.w & 0xffffu));
        const Fe b = csr_range_dot<F, false>(B, w, it.z, it.z + (it.w >> 16));
        fe_store(w + 2 * (u64)it.x, fe_mul<F>(a, b));
        return;
    }
    eval_gate_generic<F>(G, A, B, w, G.items[t]);
}

// A NARROW level (the usual case of a deep circuit: ~800 gates per level in mulgraph(2^20, window 4096)) costs its
// latency, and with one lane per gate that is a chain of ~10 dependent round trips -- record, then column -> witness for
// each of the ~4.7 entries of the gate's two rows in turn -- plus ~1400 dependent VALU instructions.  Here a Mul gate
// takes EIGHT lanes: lanes 0-3 one entry each of its A row, lanes 4-7 of its B row (more entries: strided), every lane one
// full Montgomery product, the partial sums folded by xor-shuffles, lane 0 multiplies and stores.  The chain is record ->
// {column, value} -> witness, and ~650 instructions.  Other gate kinds run on lane 0 of their group as before.
// The chain is record -> {column, value} -> witness; the level-ordered column copy (G.cols, 32 bytes per item) takes the
// column out of it: a lane's column address depends on nothing but its index, so it is record -> value beside column -> witness.
__global__ __launch_bounds__(kBlock) void k_eval_fill_cols(const uint4* __restrict__ mul, u32 count, const u32* __restrict__ col_a,
                                                          const u32* __restrict__ col_b, u32* __restrict__ cols) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    const u64 t = i / kEvalLanes;
    const u32 sub = (u32)(i % kEvalLanes);
    if (t >= count) return;
    const uint4 it = mul[t];
    u32 c = 0;
    if (it.w != 0xffffffffu) {
        const bool right = sub >= kEvalLanes / 2;
        const u32 k = sub % (kEvalLanes / 2);
        const u32 first = right ? it.z : it.y, cnt = right ? (it.w >> 16) : (it.w & 0xffffu);
        if (k < cnt) c = (right ? col_b : col_a)[first + k];
    }
    cols[i] = c;
}

// one level's worth of work of one lane: group t of the level, lane `sub` of the group; `it` / `my_col` are the group's record and the
// lane's first column (loaded by the caller: the fused kernel fetches the next level's while this level computes)
template <class F>
__device__ __forceinline__ void eval_lanes_body(const EvalGates& G, const CsrDev& A, const CsrDev& B, uint4* __restrict__ w, u32 t, u32 sub,
                                                bool live, const uint4 it, u32 my_col) {
    const bool is_mul = it.w != 0xffffffffu;
    Fe part = fe_zero();
    if (is_mul) {
        const bool right = sub >= kEvalLanes / 2;
        const u32 k = sub % (kEvalLanes / 2);
        const u32 first = right ? it.z : it.y, cnt = right ? (it.w >> 16) : (it.w & 0xffffu);
        const u32* col = right ? B.col : A.col;
        const uint4* val = right ? B.val : A.val;
#pragma unroll 1
        for (u32 j = k; j < cnt; j += kEvalLanes / 2) {
            const u32 c = (j == k) ? my_col : gload(col + first + j);
            const Fe v = fe_gload(val + 2 * (u64)(first + j));
            const Fe p = fe_mul<F>(v, fe_gload(w + 2 * (u64)c));
            part = (j == k) ? p : fe_add<F>(part, p);
        }
    }
    // every lane of the wave takes part in the shuffles (groups of other kinds and groups past the end carry zeros)
#pragma unroll 1
    for (int off = 1; off < (int)kEvalLanes / 2; off <<= 1) {
        Fe o;
#pragma unroll
        for (int i = 0; i < kLimbs; ++i) o.l[i] = (u32)__shfl_xor((int)part.l[i], off, kSlice);
        part = fe_add<F>(part, o);
    }
    Fe other;
#pragma unroll
    for (int i = 0; i < kLimbs; ++i) other.l[i] = (u32)__shfl_xor((int)part.l[i], (int)kEvalLanes / 2, kSlice);
    if (!live) return;
    if (is_mul) {
        if (sub == 0) fe_store(w + 2 * (u64)it.x, fe_mul<F>(part, other));
        return;
    }
    const u32 g = G.items[t];
    if (G.kind[g] == 2) eval_split_lanes<F>(G, w, g, sub);
    else if (sub == 0) eval_gate_generic<F>(G, A, B, w, g);
}

template <class F>
__global__ __launch_bounds__(kBlock) void k_eval_level_lanes(EvalGates G, CsrDev A, CsrDev B, uint4* __restrict__ w) {
    const u32 t = (blockIdx.x * kBlock + threadIdx.x) / kEvalLanes, sub = threadIdx.x % kEvalLanes;
    const bool live = t < G.count;
    uint4 it = make_uint4(0u, 0u, 0u, 0xffffffffu);
    u32 my_col = 0;
    if (live) {
        it = gload(G.mul + t);
        my_col = gload(G.cols + (u64)t * kEvalLanes + sub);
    }
    eval_lanes_body<F>(G, A, B, w, t, sub, live, it, my_col);
}

// A RUN of consecutive levels of at most kEvalFusedGates (128) gates each (a small or narrow circuit: the reference's own
// benchmark circuit, a 2^10-gate chain) in ONE launch of ONE workgroup: a level boundary is a workgroup barrier (~0.1 us)
// instead of a kernel boundary (~3 us of launch and first-touch latency), and the next level's records are in flight while this
// level computes.  The waves of a workgroup share their CU's write-through L1, so a wire stored before the barrier is what a
// load after it returns (workgroup-scope release / acquire = __syncthreads).  G.items / G.mul / G.cols are the WHOLE plan's arrays here.
constexpr u32 kEvalFusedBlock = 1024;
constexpr u32 kEvalFusedGates = kEvalFusedBlock / kEvalLanes;
template <class F>
__global__ __launch_bounds__(kEvalFusedBlock) void k_eval_levels_fused(EvalGates G, const u32* __restrict__ level_ofs, u32 l0, u32 l1, CsrDev A, CsrDev B,
                                                              uint4* __restrict__ w) {
    const u32 t = threadIdx.x / kEvalLanes, sub = threadIdx.x % kEvalLanes;
    u32 lo = sload(level_ofs + l0), hi = sload(level_ofs + l0 + 1);
    uint4 it = make_uint4(0u, 0u, 0u, 0xffffffffu);
    u32 my_col = 0;
    if (t < hi - lo) {
        it = gload(G.mul + lo + t);
        my_col = gload(G.cols + (u64)(lo + t) * kEvalLanes + sub);
    }
#pragma unroll 1
    for (u32 l = l0; l < l1; ++l) {
        const bool live = t < hi - lo;
        EvalGates L = G;
        L.items = G.items + lo;
        L.count = hi - lo;
        // the next level's record and column do not depend on this level's results
        const u32 nlo = hi, nhi = l + 1 < l1 ? sload(level_ofs + l + 2) : hi;
        uint4 nit = make_uint4(0u, 0u, 0u, 0xffffffffu);
        u32 ncol = 0;
        if (t < nhi - nlo) {
            nit = gload(G.mul + nlo + t);
            ncol = gload(G.cols + (u64)(nlo + t) * kEvalLanes + sub);
        }
        eval_lanes_body<F>(L, A, B, w, t, sub, live, it, my_col);
        __syncthreads();
        it = nit; my_col = ncol; lo = nlo; hi = nhi;
    }
}

// ---------------------------------------------------------------------------------------------
// K6: `createPolynomialsFFT` (src/QAP.hs:512-525) for a batch of wires: the column view (CSC) of a matrix is built
// on the device once, a batch of columns is densified into zeroed length-N buffers -- the per-wire `Map root value`
// of the GenQAP (src/QAP.hs:94-99) after `addMissingZeroes` (src/QAP.hs:566-576), for these wires only -- and the
// batched inverse NTT interpolates them.

// count[c] += 1 for every stored entry of column c
__global__ __launch_bounds__(kBlock) void k_col_histogram(const u32* __restrict__ col, u64 nnz, u32* __restrict__ count) {
    for (u64 e = (u64)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (u64)gridDim.x * kBlock) atomicAdd(&count[col[e]], 1u);
}

// out[i] = in[0] + ... + in[i-1] for i <= n (one workgroup: a one-off pass over m counters)
__global__ __launch_bounds__(1024) void k_exclusive_scan(const u32* __restrict__ in, u32* __restrict__ out, u64 n) {
    __shared__ u32 buf[1024];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u64 base = 0; base < n; base += 1024) {
        const u64 i = base + threadIdx.x;
        const u32 v = i < n ? in[i] : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (u32 off = 1; off < 1024; off <<= 1) {             // Hillis-Steele inclusive scan
            const u32 t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) out[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// CSR -> CSC: entry e of row i goes to slot colptr[c] + (a ticket of column c).  The order inside a column is
// whatever the atomics give; nothing downstream depends on it (rows of a column are distinct after normalisation).
__global__ __launch_bounds__(kBlock) void k_csc_fill(CsrDev M, u64 n_rows, const u32* __restrict__ colptr, u32* __restrict__ cursor,
                                                    u32* __restrict__ rowidx, u32* __restrict__ colid, uint4* __restrict__ tval) {
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n_rows; i += (u64)gridDim.x * kBlock) {
        for (u32 e = M.rowptr[i]; e < M.rowptr[i + 1]; ++e) {
            const u32 c = M.col[e];
            const u32 dst = colptr[c] + atomicAdd(&cursor[c], 1u);
            rowidx[dst] = (u32)i;
            colid[dst] = c;
            tval[2 * (u64)dst] = M.val[2 * (u64)e];
            tval[2 * (u64)dst + 1] = M.val[2 * (u64)e + 1];
        }
    }
}

// densify columns [wire_begin, wire_begin + wire_count) into out[w][0..N) (zero filled beforehand); every entry
// carries its column id, so there is no search
__global__ __launch_bounds__(kBlock) void k_scatter_columns(const u32* __restrict__ colptr, const u32* __restrict__ rowidx,
                                                           const u32* __restrict__ colid, const uint4* __restrict__ val,
                                                           u64 wire_begin, u64 wire_count, u32 log_n, uint4* __restrict__ out) {
    const u64 e_begin = colptr[wire_begin], e_end = colptr[wire_begin + wire_count];
    for (u64 e = e_begin + (u64)blockIdx.x * kBlock + threadIdx.x; e < e_end; e += (u64)gridDim.x * kBlock) {
        uint4* dst = out + 2 * (((u64)(colid[e] - wire_begin) << log_n) + rowidx[e]);
        dst[0] = val[2 * e];
        dst[1] = val[2 * e + 1];
    }
}

// `FFT.interpolate` of a SPARSE column without a transform.  A QAP column is the interpolant of a wire's few appearances
// (an intermediate wire of a Mul-gate circuit has one entry in C and one or two in A / B): with k nonzero values v_t at roots
// omega^(i_t) the coefficients are c_j = (1/N) sum_t v_t omega^(-i_t j) -- k geometric progressions -- against the ~10
// Montgomery products per coefficient of the radix-2 transform, with no zero fill and no scatter.
// Coefficient j = 256 (blk + s) + l of lane l at step s factors as
//     c_j = sum_t  A_t(l) * B_t(blk + s),   A_t(l) = (v_t / N) omega^(-i_t l),   B_t(x) = omega^(-256 i_t x):
// A_t lives in 9 VGPRs per entry for the whole block, B_t is UNIFORM over the block -- one s_load_dwordx8 per entry and
// step from the stride-256 power table, unpacked on the scalar unit, and it enters v_mad_u64_u32 as the SGPR operand --
// and the k products of a coefficient share ONE Montgomery reduction (fe_dot): 81 k + 90 multiplier instructions per
// coefficient, no running powers, no step factors.
// Columns with more than kDirectMax entries keep the batched inverse NTT (the host splits a batch into dense runs and the
// rest).  blockIdx.y = column of the batch; a block produces kBlock * L consecutive coefficients (every store of the block is
// one contiguous 8 KiB run).
constexpr u32 kDirectMax = 4;       // k_col_direct: one shared reduction per coefficient
constexpr u32 kDirectMid = 12;      // k_col_direct_mid: 5 .. 12 entries, a reduction per group of four (a kernel of its own: its
                                    // lane factors would cost the common case its fifth wave)
struct ColDirect {
    const u32* colptr;
    const u32* rowidx;
    const uint4* val;
    u64 wire_begin;
    u32 log_n;
    u32 steps;              // L
    const uint4* tw_lo;     // omega_N^-j, j < min(N, 1024)
    const uint4* tw_hi;     // omega_N^-(1024 j), j < N / 1024 (null for N <= 1024)
    const uint4* tw_blk;    // omega_N^-(256 j), j < max(1, N / 256)
    FeArg inv_n;            // 1/N (Montgomery)
};

template <class F>
__device__ __forceinline__ Fe omega_inv_pow(const ColDirect& P, u64 e) {     // omega_N^-e, e < N
    if (P.tw_hi == nullptr) return fe_gload(P.tw_lo + 2 * e);
    return fe_mul<F>(fe_gload(P.tw_lo + 2 * (e & 1023u)), fe_gload(P.tw_hi + 2 * (e >> 10)));
}

// entry t of the column: its row and the lane's factor A_t(l)
template <class F>
__device__ __forceinline__ void col_direct_entry(const ColDirect& P, u32 e, u32 l, u64 mask, const Fe& inv_n, Fe& lane, u32& row) {
    row = sload(P.rowidx + e);
    const Fe v = fe_mul<F>(fe_sload(P.val + 2 * (u64)e), inv_n);
    lane = fe_mul<F>(omega_inv_pow<F>(P, ((u64)row * l) & mask), v);
    __builtin_amdgcn_sched_barrier(0);      // entry by entry: the set-up of eight entries scheduled together peaks at 183 registers
}
// (a fold over the entries, not a loop: three products per entry are more than `#pragma unroll` will unroll, and a rolled
// loop would index lane[] dynamically, i.e. keep it in scratch memory)
template <class F, int... T>
__device__ __forceinline__ void col_direct_setup(const ColDirect& P, u32 e0, u32 l, u64 mask, const Fe& inv_n, Fe* lane, u32* row,
                                                 std::integer_sequence<int, T...>) {
    (col_direct_entry<F>(P, e0 + T, l, mask, inv_n, lane[T], row[T]), ...);
}

template <class F, int K>
__device__ __forceinline__ void col_direct_body(const ColDirect& P, uint4* __restrict__ out, u32 e0) {
    const u64 N = 1ull << P.log_n, mask = N - 1;
    const u32 l = threadIdx.x;
    if (l >= N) return;                                           // N < kBlock
    const u64 blk = (u64)blockIdx.x * P.steps;                    // in units of kBlock coefficients
    uint4* dst = out + 2 * (((u64)blockIdx.y << P.log_n) + blk * kBlock + l);
    const u64 bmask = (N >> 8) ? (N >> 8) - 1 : 0;
    Fe lane[K];
    u32 row[K];
    col_direct_setup<F>(P, e0, l, mask, fe_from_arg(P.inv_n), lane, row, std::make_integer_sequence<int, K>{});
#pragma unroll 1
    for (u32 s = 0; s < P.steps; ++s) {
        Fe b[K];
#pragma unroll
        for (int t = 0; t < K; ++t) b[t] = fe_sload(P.tw_blk + 2 * (((u64)row[t] * (blk + s)) & bmask));
        fe_store(dst + 2 * (u64)s * kBlock, fe_dot<F, K>(lane, b));
    }
}

// 5 .. kDirectMid entries: the same factorisation, the entries in groups of four with a reduction each (a column accumulator
// holds six terms), 81 k + 90 ceil(k / 4) multiplier instructions per coefficient -- against ten 171-instruction products per
// coefficient for the transform such a column took before (tools/kbench.py colsk: 33 - 56 us per 2^20-point column for
// k = 5 .. 12 against 107).  The lane factors alone are 9 k registers: 2 waves per SIMD, which this rare class can afford.
template <class F, int G>
__device__ __forceinline__ Fe col_direct_group(const ColDirect& P, const Fe (&lane)[G], const u32 (&row)[G], u64 x, u64 bmask) {
    Fe b[G];
#pragma unroll
    for (int t = 0; t < G; ++t) b[t] = fe_sload(P.tw_blk + 2 * (((u64)row[t] * x) & bmask));
    const Fe r = fe_dot<F, G>(lane, b);
    // one group after the other, a group's step factors loaded behind the previous group's products: all of them at once are
    // more scalar registers of limbs than the scalar file leaves
    __builtin_amdgcn_sched_barrier(0);
    return r;
}

template <class F, int K>
__device__ __forceinline__ void col_direct_mid_body(const ColDirect& P, uint4* __restrict__ out, u32 e0) {
    constexpr int K2 = K > 8 ? 4 : K - 4, K3 = K > 8 ? K - 8 : 0;
    const u64 N = 1ull << P.log_n, mask = N - 1;
    const u32 l = threadIdx.x;
    if (l >= N) return;
    const u64 blk = (u64)blockIdx.x * P.steps;
    uint4* dst = out + 2 * (((u64)blockIdx.y << P.log_n) + blk * kBlock + l);
    const u64 bmask = (N >> 8) ? (N >> 8) - 1 : 0;
    Fe la[4], lb[K2], lc[K3 ? K3 : 1];
    u32 ra[4], rb[K2], rc[K3 ? K3 : 1];
    const Fe inv_n = fe_from_arg(P.inv_n);
    col_direct_setup<F>(P, e0, l, mask, inv_n, la, ra, std::make_integer_sequence<int, 4>{});
    col_direct_setup<F>(P, e0 + 4, l, mask, inv_n, lb, rb, std::make_integer_sequence<int, K2>{});
    if constexpr (K3 > 0) col_direct_setup<F>(P, e0 + 8, l, mask, inv_n, lc, rc, std::make_integer_sequence<int, K3>{});
#pragma unroll 1
    for (u32 s = 0; s < P.steps; ++s) {
        Fe sum = fe_add<F>(col_direct_group<F, 4>(P, la, ra, blk + s, bmask), col_direct_group<F, K2>(P, lb, rb, blk + s, bmask));
        if constexpr (K3 > 0) sum = fe_add<F>(sum, col_direct_group<F, (K3 ? K3 : 1)>(P, lc, rc, blk + s, bmask));
        fe_store(dst + 2 * (u64)s * kBlock, sum);
    }
}

template <class F>
__global__ __launch_bounds__(kBlock) void k_col_direct_mid(ColDirect P, uint4* __restrict__ out) {
    const u64 wire = P.wire_begin + blockIdx.y;
    const u32 e0 = sload(P.colptr + wire), k = sload(P.colptr + wire + 1) - e0;       // uniform over the block
    switch (k) {
        case 5: col_direct_mid_body<F, 5>(P, out, e0); break;
        case 6: col_direct_mid_body<F, 6>(P, out, e0); break;
        case 7: col_direct_mid_body<F, 7>(P, out, e0); break;
        case 8: col_direct_mid_body<F, 8>(P, out, e0); break;
        case 9: col_direct_mid_body<F, 9>(P, out, e0); break;
        case 10: col_direct_mid_body<F, 10>(P, out, e0); break;
        case 11: col_direct_mid_body<F, 11>(P, out, e0); break;
        case 12: col_direct_mid_body<F, 12>(P, out, e0); break;
        default: break;                                           // k_col_direct's or the transform's
    }
}

template <class F>
__global__ __launch_bounds__(kBlock) void k_col_direct(ColDirect P, uint4* __restrict__ out) {
    const u64 wire = P.wire_begin + blockIdx.y;
    const u32 e0 = sload(P.colptr + wire), k = sload(P.colptr + wire + 1) - e0;       // uniform over the block
    switch (k) {
        case 0: {                                                 // a wire the matrix never mentions: the zero polynomial
            const u64 N = 1ull << P.log_n;
            if (threadIdx.x >= N) break;
            uint4* dst = out + 2 * (((u64)blockIdx.y << P.log_n) + (u64)blockIdx.x * P.steps * kBlock + threadIdx.x);
            for (u32 s = 0; s < P.steps; ++s) fe_store(dst + 2 * (u64)s * kBlock, fe_zero());
            break;
        }
        case 1: col_direct_body<F, 1>(P, out, e0); break;
        case 2: col_direct_body<F, 2>(P, out, e0); break;
        case 3: col_direct_body<F, 3>(P, out, e0); break;
        case 4: col_direct_body<F, 4>(P, out, e0); break;
        default: break;                                           // a dense column: the transform's
    }
}

// len[w] = 1 + index of the last nonzero coefficient of polynomial w (0 for the zero polynomial): poly's `toPoly`
// stripping, computed where the data is.  One workgroup per polynomial, scanning down from the top.
template <class F>
__global__ __launch_bounds__(kBlock) void k_poly_len(const uint4* __restrict__ data, u32 log_n, unsigned long long* __restrict__ len,
                                                    const u32* __restrict__ colptr) {
    const u64 N = 1ull << log_n;
    const uint4* p = data + 2 * ((u64)blockIdx.x << log_n);
    // a column without entries is the zero polynomial: no scan (one workgroup walking 2^20 zeros takes milliseconds, and
    // 37 % of the A / B columns of a k = 2 Mul-gate circuit are empty); colptr points at the batch's first column
    if (colptr != nullptr && colptr[blockIdx.x] == colptr[blockIdx.x + 1]) {
        if (threadIdx.x == 0) len[blockIdx.x] = 0;
        return;
    }
    __shared__ u32 best;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    for (u64 top = N; top > 0;) {
        const u64 base = top > kBlock ? top - kBlock : 0;
        const u64 i = base + threadIdx.x;
        if (i < top && !fe_is_zero<F>(fe_load(p + 2 * i))) atomicMax(&best, (u32)(i + 1));
        __syncthreads();
        const u32 found = best;
        __syncthreads();                                        // nobody updates `best` again before everyone has read it
        if (found != 0) break;                                  // uniform
        top = base;
    }
    if (threadIdx.x == 0) len[blockIdx.x] = best;
}

}  // namespace acx
