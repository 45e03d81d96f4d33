// fr.hip.h -- 256-bit prime-field arithmetic for gfx950 (CDNA4) device code.
//
// Replaces the arithmetic of galois-field-1.0.2 `Prime p` (third party; call sites
// /root/reference/src/QAP.hs:52,87-90,450 and src/Circuit/Arithmetic.hs:131,143) on the device.
//
// Representation, chosen from measured gfx950 issue rates (tools/microbench/valu_rates.hip,
// profiles/r01_valu_rates.txt): v_mad_u64_u32 issues at the SAME rate as v_addc_co_u32
// (4 cycles per wave64 instruction), so carry handling -- not multiplies -- dominates a
// 8 x 32-bit-limb Montgomery product (136 mads + >=128 carry ops).  With 9 limbs of 29 bits a
// column of 18 limb products (< 2^58 each) fits a 64-bit accumulator with no carry tracking:
// one Montgomery product is 162 v_mad_u64_u32 + 9 v_mul_lo_u32 + 17 v_lshrrev_b64 + 17 v_and
// and hipcc emits exactly that from plain C++ (no inline asm needed).
//
//   * In registers: Fe = 9 limbs, radix 2^29, Montgomery radix R = 2^261.
//     "Lazy" invariant: limbs < 2^29 (top limb unbounded by the mask), value in [0, 2p).
//     mul() accepts values < 4p with limbs < 2^30 and returns a lazy value without any
//     conditional subtraction (a*b/R + p < 2p because 16p/R <= 1/4).
//   * In memory ("dev" format): the lazy value packed into 8 x u32 little-endian (2p < 2^256).
//   * At the ABI: canonical [0,p), same packing, plain (non-Montgomery).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "field_consts.h"

namespace acx {

using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;

constexpr int kLimbs = 9;
constexpr int kLimbBits = 29;
constexpr u32 kLimbMask = (1u << kLimbBits) - 1;

struct Fe {
    u32 l[kLimbs];
};

// Host-prepared element passed by value as a kernel argument (already in limb form).
struct FeArg {
    u32 l[kLimbs];
};

__device__ __forceinline__ Fe fe_from_arg(const FeArg& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < kLimbs; ++i) r.l[i] = a.l[i];
    return r;
}

__device__ __forceinline__ Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < kLimbs; ++i) r.l[i] = 0;
    return r;
}

template <class F>
__device__ __forceinline__ Fe fe_one_mont() {
    Fe r;
#pragma unroll
    for (int i = 0; i < kLimbs; ++i) r.l[i] = F::R1[i];
    return r;
}

// ---- packing: 8 x u32 words <-> 9 x 29-bit limbs -------------------------------------------
__device__ __forceinline__ Fe fe_unpack(const u32 w[8]) {
    Fe r;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        const int bit = kLimbBits * k, wi = bit >> 5, off = bit & 31;
        u64 pair = w[wi];
        if (wi + 1 < 8) pair |= (u64)w[wi + 1] << 32;
        r.l[k] = (u32)(pair >> off) & kLimbMask;
    }
    return r;
}

// requires normalized limbs (each < 2^29) and value < 2^256
__device__ __forceinline__ void fe_pack(const Fe& a, u32 w[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, k0 = bit / kLimbBits, o = bit - kLimbBits * k0;
        u32 v = a.l[k0] >> o;
        if (k0 + 1 < kLimbs) v |= a.l[k0 + 1] << (kLimbBits - o);
        if (k0 + 2 < kLimbs && 2 * kLimbBits - o < 32) v |= a.l[k0 + 2] << (2 * kLimbBits - o);
        w[j] = v;
    }
}

// Every element buffer lives in global memory.  Pinning the address space and the vector type gives
// exactly two global_load/store_dwordx4 per element; with generic pointers hipcc emitted flat
// accesses (for pointers reached through a descriptor) and re-split the 32 bytes into 4+16+12.
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4u32 g_v4u32_t;

__device__ __forceinline__ Fe fe_load(const uint4* __restrict__ p) {
    const v4u32 lo = *(const g_v4u32_t*)p, hi = *(const g_v4u32_t*)(p + 1);
    const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return fe_unpack(w);
}

__device__ __forceinline__ void fe_store(uint4* __restrict__ p, const Fe& a) {
    u32 w[8];
    fe_pack(a, w);
    v4u32 lo, hi;
    lo.x = w[0]; lo.y = w[1]; lo.z = w[2]; lo.w = w[3];
    hi.x = w[4]; hi.y = w[5]; hi.z = w[6]; hi.w = w[7];
    *(g_v4u32_t*)p = lo;
    *(g_v4u32_t*)(p + 1) = hi;
}

// ---- carries --------------------------------------------------------------------------------
// Sequential carry propagation; limbs may be up to 2^32-1 on entry.
__device__ __forceinline__ void fe_carry(Fe& a) {
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < kLimbs - 1; ++k) {
        const u32 t = a.l[k] + c;
        a.l[k] = t & kLimbMask;
        c = t >> kLimbBits;
    }
    a.l[kLimbs - 1] += c;
}

// a - M if a >= M else a;   a normalized, M a compile-time limb constant.
template <const u32 (&M)[kLimbs]>
__device__ __forceinline__ Fe fe_cond_sub(const Fe& a) {
    Fe d;
    i32 c = 0;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        const i32 t = (i32)a.l[k] - (i32)M[k] + c;
        d.l[k] = (k < kLimbs - 1) ? ((u32)t & kLimbMask) : (u32)t;
        c = t >> kLimbBits;  // arithmetic shift: borrow = -1
    }
    const bool neg = c < 0;  // a < M
    Fe r;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) r.l[k] = neg ? a.l[k] : d.l[k];
    return r;
}

// ---- ring operations on lazy values -----------------------------------------------------------
template <class F>
__device__ __forceinline__ Fe fe_add(const Fe& a, const Fe& b) {
    Fe s;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) s.l[k] = a.l[k] + b.l[k];
    fe_carry(s);
    return fe_cond_sub<F::P2>(s);
}

template <class F>
__device__ __forceinline__ Fe fe_sub(const Fe& a, const Fe& b) {
    Fe s;
    i32 c = 0;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        const i32 t = (i32)a.l[k] + (i32)F::P2[k] - (i32)b.l[k] + c;
        s.l[k] = (k < kLimbs - 1) ? ((u32)t & kLimbMask) : (u32)t;
        c = t >> kLimbBits;
    }
    return fe_cond_sub<F::P2>(s);
}

// P[0] as the multiplier of the quotient digits.  For BLS12-381 Fr the limb is 1 and the compiler turns m * 1 + acc into a
// 64-bit add of the ZERO-EXTENDED digit -- which pins every digit to an aligned register pair with a zero upper half: nine
// extra VGPRs in kernels that live at the 128-register ceiling (k_ntt_r4: four BLS12-381 instances at 129-133 registers =
// 3 waves per SIMD, two spilling; profiles/r04_ntt.txt).  The limb therefore reaches the product through a scalar register
// the optimiser cannot see through: the same instruction count (one v_mad_u64_u32 instead of one v_lshl_add_u64), no pairs.
template <class F>
__device__ __forceinline__ u32 fe_p0() {
    if (F::P[0] == 1u) {
        u32 one;
        asm("s_mov_b32 %0, 1" : "=s"(one));
        return one;
    }
    return F::P[0];
}

// Montgomery product a*b/R mod p, finely integrated product scanning.  a, b: limbs < 2^30,
// values < 4p.  Result lazy (normalized, < 2p).
template <class F>
__device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b) {
    u64 acc = 0;
    u32 m[kLimbs];
    Fe r;
    const u32 p0 = fe_p0<F>();
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (u64)m[i] * F::P[k - i];
        m[k] = ((u32)acc * F::N0) & kLimbMask;
        acc += (u64)m[k] * p0;
        acc >>= kLimbBits;
    }
#pragma unroll
    for (int k = kLimbs; k < 2 * kLimbs - 1; ++k) {
#pragma unroll
        for (int i = k - kLimbs + 1; i < kLimbs; ++i) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - kLimbs + 1; i < kLimbs; ++i) acc += (u64)m[i] * F::P[k - i];
        r.l[k - kLimbs] = (u32)acc & kLimbMask;
        acc >>= kLimbBits;
    }
    r.l[kLimbs - 1] = (u32)acc;
    return r;
}

// sum_t a[t] * b[t] / R mod p with ONE Montgomery reduction, every limb product of every term chained into the running
// column accumulator (fe_mul is the case K = 1): 81 K + 90 multiplier instructions and no separate column additions.
// Operands as fe_mul's; K <= kWideTerms ((9 K + 9) 2^58 < 2^64).  Result lazy (< 2p for operands < 2p: wide_reduce's bound).
template <class F, int K>
__device__ __forceinline__ Fe fe_dot(const Fe (&a)[K], const Fe (&b)[K]) {
    static_assert(K >= 1 && K <= 6, "column accumulators overflow beyond six terms");
    u64 acc = 0;
    u32 m[kLimbs];
    Fe r;
    const u32 p0 = fe_p0<F>();
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (u64)a[t].l[i] * b[t].l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (u64)m[i] * F::P[k - i];
        m[k] = ((u32)acc * F::N0) & kLimbMask;
        acc += (u64)m[k] * p0;
        acc >>= kLimbBits;
    }
#pragma unroll
    for (int k = kLimbs; k < 2 * kLimbs - 1; ++k) {
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int i = k - kLimbs + 1; i < kLimbs; ++i) acc += (u64)a[t].l[i] * b[t].l[k - i];
#pragma unroll
        for (int i = k - kLimbs + 1; i < kLimbs; ++i) acc += (u64)m[i] * F::P[k - i];
        r.l[k - kLimbs] = (u32)acc & kLimbMask;
        acc >>= kLimbBits;
    }
    r.l[kLimbs - 1] = (u32)acc;
    return r;
}

// ---- lazy butterfly arithmetic (NTT) ---------------------------------------------------------------
// "Loose" values: limbs < 2^29 + 8 (top limb free), value < 64p (fits 261 bits for both fields).
// fe_mul accepts a loose left operand when the right operand is strictly normalised and < 2p
// (a stored twiddle or constant): a*b/R + p < (64 p/R + 1) p <= 2p, columns stay < 2^64.
// One parallel (non-rippling) carry pass keeps the loose invariant: 27 independent VALU ops
// instead of the 24-deep ripple + conditional subtraction of fe_add/fe_sub (225-230 cycles each,
// tools/microbench/fe_rates.hip).
__device__ __forceinline__ void fe_carry_loose(Fe& a) {
    u32 c[kLimbs - 1];
#pragma unroll
    for (int k = 0; k < kLimbs - 1; ++k) { c[k] = a.l[k] >> kLimbBits; a.l[k] &= kLimbMask; }
#pragma unroll
    for (int k = 1; k < kLimbs; ++k) a.l[k] += c[k - 1];
}

// a + b, loose in, loose out; value grows to a + b.  CARRY = false skips the carry pass: limbs may
// then reach 2^30 + 16, which the next (carrying) add/sub and fe_mul's left operand tolerate
// (9 * 2^31.4 * 2^29 + 9 * 2^58 < 2^64); used on every other butterfly stage.
template <bool CARRY = true>
__device__ __forceinline__ Fe fe_add_lazy(const Fe& a, const Fe& b) {
    Fe s;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) s.l[k] = a.l[k] + b.l[k];
    if (CARRY) fe_carry_loose(s);
    return s;
}

// a - b + 4p for b strictly normalised and < 2p (a product); borrow-free thanks to the "fat"
// limb form of 4p; value grows by at most 4p
template <class F, bool CARRY = true>
__device__ __forceinline__ Fe fe_sub_lazy(const Fe& a, const Fe& b) {
    Fe s;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) s.l[k] = a.l[k] + F::P4FAT[k] - b.l[k];
    if (CARRY) fe_carry_loose(s);
    return s;
}

// a - b + FAT for a "fat" limb form of a multiple of p (P4FAT, P8FAT): borrow-free as long as every limb of b is at most
// the matching limb of FAT.  P8FAT takes an UNCARRIED sum of two strict values as b (limbs <= 2^30 - 2, value < 4p).
template <const u32 (&FAT)[kLimbs], bool CARRY = true>
__device__ __forceinline__ Fe fe_sub_fat(const Fe& a, const Fe& b) {
    Fe s;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) s.l[k] = a.l[k] + FAT[k] - b.l[k];
    if (CARRY) fe_carry_loose(s);
    return s;
}

// ---- deferred reduction: sum of raw limb products, one Montgomery reduction per sum --------------
// Column accumulators of the schoolbook product: c[k] = sum over terms of sum_{i+j=k} a_i b_j.
// Each v_mad_u64_u32 adds straight into its column (no carry handling at all).  With limbs
// < 2^29 a column grows by < 9 * 2^58 per term; wide_reduce() adds < 9 * 2^58 + 2^36 more, so at
// most kWideTerms = 6 terms may be accumulated before reducing ((9*6 + 9) * 2^58 < 2^64).
constexpr int kWideTerms = 6;

struct Wide {
    u64 c[2 * kLimbs - 1];
};

__device__ __forceinline__ void wide_zero(Wide& w) {
#pragma unroll
    for (int k = 0; k < 2 * kLimbs - 1; ++k) w.c[k] = 0;
}

// first term of a sum: plain products, no accumulator to initialise
__device__ __forceinline__ void wide_mul(Wide& w, const Fe& a, const Fe& b) {
#pragma unroll
    for (int k = 0; k < 2 * kLimbs - 1; ++k) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < kLimbs; ++i) {
            const int j = k - i;
            if (j >= 0 && j < kLimbs) acc += (u64)a.l[i] * b.l[j];
        }
        w.c[k] = acc;
    }
}

__device__ __forceinline__ void wide_mac(Wide& w, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < kLimbs; ++i)
#pragma unroll
        for (int j = 0; j < kLimbs; ++j) w.c[i + j] += (u64)a.l[i] * b.l[j];
}

// Montgomery reduction of the accumulated sum T (< 2^64 per column as above): T/R mod p, lazy.
// For operands < 2p each and <= 6 terms: T/R + p < (24 p/R + 1) p < 2p for both fields.
template <class F>
__device__ __forceinline__ Fe wide_reduce(const Wide& w) {
    u64 t = 0;
    u32 m[kLimbs];
    Fe r;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        t += w.c[k];
#pragma unroll
        for (int i = 0; i < k; ++i) t += (u64)m[i] * F::P[k - i];
        m[k] = ((u32)t * F::N0) & kLimbMask;
        t += (u64)m[k] * F::P[0];
        t >>= kLimbBits;
    }
#pragma unroll
    for (int k = kLimbs; k < 2 * kLimbs - 1; ++k) {
        t += w.c[k];
#pragma unroll
        for (int i = k - kLimbs + 1; i < kLimbs; ++i) t += (u64)m[i] * F::P[k - i];
        r.l[k - kLimbs] = (u32)t & kLimbMask;
        t >>= kLimbBits;
    }
    r.l[kLimbs - 1] = (u32)t;
    return r;
}

// ---- small-coefficient dot products ---------------------------------------------------------------
// A constraint matrix compiled from a program (src/Circuit/Expr.hs:256-305) carries coefficients +-1, +-2 and small
// program constants.  For |c| <= kSmallCoeffMax the term c * x needs no Montgomery product: x is already a Montgomery
// residue and c * (w R) = (c w) R, so a row's dot product is nine signed columns s[k] += c * x_k (one v_mad_i64_i32
// each) followed by ONE exact reduction of the signed 9-column sum S (|S| < 2^59 * 2^232):
//   q  = floor(S / p - 1/2) estimated in f64 from columns 8 and 7 (the rest is < 2^-20 p; f64 rounding < 2^-12), so that
//        S - q p lies in (0.49 p, 1.51 p);
//   r  = S - q p column by column with q = qh * 2^29 + ql (18 multiplier instructions), one signed carry chain.
// Result lazy (normalised limbs, value in [0, 2p)).  At most kWideTerms terms per sum.
constexpr i32 kSmallCoeffMax = 1 << 27;
using i64 = int64_t;

template <class F>
__device__ __forceinline__ Fe small_reduce(const i64 (&s)[kLimbs]) {
    const double top = __builtin_fma((double)s[kLimbs - 1], 536870912.0, (double)s[kLimbs - 2]);   // S ~ top * 2^203
    const double qd = __builtin_floor(__builtin_fma(top, F::PINV203, -0.5));
    const double qhd = __builtin_floor(qd * (1.0 / 536870912.0));
    const i32 qh = (i32)qhd;                                        // |q| < 2^41: qh in (-2^12, 2^12)
    const i32 ql = (i32)__builtin_fma(qhd, -536870912.0, qd);       // [0, 2^29)
    Fe r;
    i64 t = 0;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        t += s[k] - (i64)ql * (i64)(i32)F::P[k];
        if (k > 0) t -= (i64)qh * (i64)(i32)F::P[k - 1];
        if (k < kLimbs - 1) {
            r.l[k] = (u32)t & kLimbMask;
            t >>= kLimbBits;                                        // arithmetic: borrows propagate as -1
        }
    }
    t -= ((i64)qh * (i64)(i32)F::P[kLimbs - 1]) << kLimbBits;       // column 9 of q p; the sum is < 2p, so it cancels
    r.l[kLimbs - 1] = (u32)t;
    return r;
}

// lazy [0,2p) -> canonical residue [0,p) of the same Montgomery value
template <class F>
__device__ __forceinline__ Fe fe_reduce(const Fe& a) {
    return fe_cond_sub<F::P>(a);
}

// value == 0 mod p, for a lazy value (0 or p)
template <class F>
__device__ __forceinline__ bool fe_is_zero(const Fe& a) {
    u32 z = 0, e = 0;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        z |= a.l[k];
        e |= a.l[k] ^ F::P[k];
    }
    return z == 0 || e == 0;
}

template <class F>
__device__ __forceinline__ Fe fe_to_mont(const Fe& canonical) {
    Fe r2;
#pragma unroll
    for (int i = 0; i < kLimbs; ++i) r2.l[i] = F::R2[i];
    return fe_mul<F>(canonical, r2);
}

// Montgomery lazy value -> canonical plain integer in [0,p)
template <class F>
__device__ __forceinline__ Fe fe_from_mont(const Fe& a) {
    Fe one = fe_zero();
    one.l[0] = 1;
    return fe_reduce<F>(fe_mul<F>(a, one));
}

// ---- inversion by divsteps ---------------------------------------------------------------------------
// 1/a for a Montgomery residue a (lazy in, lazy out; a != 0), by the Bernstein-Yang division steps in the half-delta
// formulation (zeta = -(delta + 1/2); 590 steps suffice for any modulus and input below 2^256), in batches of 29 -- the limb
// width, so that dividing by 2^29 is dropping a limb: 21 rounds of {29 branch-free steps on the low 32 bits of f and g that
// yield a 2x2 transition matrix with entries below 2^29 in magnitude; the matrix applied to (f, g) exactly and to (d, e)
// modulo p} with signed 9 x 29-bit limbs.  About 20 000 instructions against 78 000 for a^(p-2) by square-and-multiply
// (254 squarings + ~127 products), and it is the LATENCY of one inversion that an Equal gate adds to its level of the
// GPU witness generation (src/Circuit/Arithmetic.hs:117-131).  Checked against big-integer arithmetic by a bit-accurate
// Python model of this routine before it was written (tests: every Equal gate of the -m gpu suite).
template <class F>
__device__ __forceinline__ Fe fe_inv_divsteps(const Fe& a_mont) {
    using i64 = int64_t;
    constexpr i32 kM = (i32)kLimbMask;
    constexpr u32 kPinv = (0u - F::N0) & kLimbMask;                 // p^-1 mod 2^29 (N0 = -p^-1)
    const Fe c = fe_reduce<F>(a_mont);                              // canonical residue of a R
    i32 f[kLimbs], g[kLimbs], d[kLimbs], e[kLimbs];
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) { f[k] = (i32)F::P[k]; g[k] = (i32)c.l[k]; d[k] = 0; e[k] = k == 0 ? 1 : 0; }
    i32 zeta = -1;
#pragma unroll 1
    for (int round = 0; round < 21; ++round) {
        // 29 division steps on the low bits
        u32 u = 1, v = 0, q = 0, r = 1;
        u32 fl = (u32)f[0] | ((u32)f[1] << kLimbBits), gl = (u32)g[0] | ((u32)g[1] << kLimbBits);
#pragma unroll 1
        for (int i = 0; i < kLimbBits; ++i) {
            u32 c1 = (u32)(zeta >> 31);                             // zeta < 0
            const u32 m2 = 0u - (gl & 1u);                          // g odd
            const u32 x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
            gl += x & m2; q += y & m2; r += z & m2;
            c1 &= m2;
            zeta = (i32)(((u32)zeta ^ c1) - 1u);
            fl += gl & c1; u += q & c1; v += r & c1;
            gl >>= 1; u <<= 1; v <<= 1;
        }
        const i64 U = (i32)u, V = (i32)v, Q = (i32)q, R = (i32)r;
        {   // (d, e) <- (U d + V e, Q d + R e) / 2^29 mod p, both kept in (-2p, p)
            const i32 sd = d[kLimbs - 1] >> 31, se = e[kLimbs - 1] >> 31;
            i32 md = ((i32)U & sd) + ((i32)V & se), me = ((i32)Q & sd) + ((i32)R & se);
            i64 cd = U * d[0] + V * e[0], ce = Q * d[0] + R * e[0];
            md -= (i32)((kPinv * (u32)cd + (u32)md) & kLimbMask);
            me -= (i32)((kPinv * (u32)ce + (u32)me) & kLimbMask);
            cd += (i64)(i32)F::P[0] * md; ce += (i64)(i32)F::P[0] * me;
            cd >>= kLimbBits; ce >>= kLimbBits;
#pragma unroll
            for (int k = 1; k < kLimbs; ++k) {
                cd += U * d[k] + V * e[k] + (i64)(i32)F::P[k] * md;
                ce += Q * d[k] + R * e[k] + (i64)(i32)F::P[k] * me;
                d[k - 1] = (i32)cd & kM; e[k - 1] = (i32)ce & kM;
                cd >>= kLimbBits; ce >>= kLimbBits;
            }
            d[kLimbs - 1] = (i32)cd; e[kLimbs - 1] = (i32)ce;
        }
        {   // (f, g) <- (U f + V g, Q f + R g) / 2^29, exactly
            i64 cf = U * f[0] + V * g[0], cg = Q * f[0] + R * g[0];
            cf >>= kLimbBits; cg >>= kLimbBits;
#pragma unroll
            for (int k = 1; k < kLimbs; ++k) {
                cf += U * f[k] + V * g[k];
                cg += Q * f[k] + R * g[k];
                f[k - 1] = (i32)cf & kM; g[k - 1] = (i32)cg & kM;
                cf >>= kLimbBits; cg >>= kLimbBits;
            }
            f[kLimbs - 1] = (i32)cf; g[kLimbs - 1] = (i32)cg;
        }
    }
    // g = 0, f = +-1, d = +-1/c in (-2p, p): y = sign(f) * d reduced to [0, p)
    const i32 sf = f[kLimbs - 1] >> 31;                             // f = -1: all limbs 2^29 - 1 and a negative top
    i32 y[kLimbs];
    {
        const i32 neg = d[kLimbs - 1] >> 31;
        i32 carry = 0;
#pragma unroll
        for (int k = 0; k < kLimbs; ++k) {                          // d + (p if d < 0), then the sign of f
            i32 t = d[k] + ((i32)F::P[k] & neg);
            t = (t ^ sf) - sf;
            t += carry;
            y[k] = k < kLimbs - 1 ? (t & kM) : t;
            carry = k < kLimbs - 1 ? (t >> kLimbBits) : 0;
        }
    }
    {
        const i32 neg = y[kLimbs - 1] >> 31;
        i32 carry = 0;
#pragma unroll
        for (int k = 0; k < kLimbs; ++k) {                          // + p once more if the negation made it negative
            i32 t = y[k] + ((i32)F::P[k] & neg) + carry;
            y[k] = k < kLimbs - 1 ? (t & kM) : t;
            carry = k < kLimbs - 1 ? (t >> kLimbBits) : 0;
        }
    }
    Fe yv;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) yv.l[k] = (u32)y[k];
    Fe r2;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) r2.l[k] = F::R2[k];
    // y = (a R)^-1 = a^-1 R^-1 (canonical, possibly in [p, 2p) before the last step: lazy is enough); a^-1 R = y R^2:
    return fe_mul<F>(fe_mul<F>(yv, r2), r2);
}

// canonical check on a plain unpacked value: a < p
template <class F>
__device__ __forceinline__ bool fe_lt_p(const Fe& a) {
    i32 c = 0;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        const i32 t = (i32)a.l[k] - (i32)F::P[k] + c;
        c = t >> kLimbBits;
    }
    return c < 0;
}

// a^e for a 64-bit exponent (square and multiply, LSB first)
template <class F>
__device__ __forceinline__ Fe fe_pow(Fe base, u64 e) {
    Fe acc = fe_one_mont<F>();
    while (e) {
        if (e & 1) acc = fe_mul<F>(acc, base);
        base = fe_mul<F>(base, base);
        e >>= 1;
    }
    return acc;
}

}  // namespace acx
