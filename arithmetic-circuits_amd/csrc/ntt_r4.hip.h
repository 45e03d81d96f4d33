// ntt_r4.hip.h -- register-resident NTT pass for gfx950: four elements per lane, two radix-2 DIT
// stages per round in registers, digit exchanges between rounds.
//
// Replaces galois-fft `FFT.fft` / `FFT.interpolate` (third party; call sites
// /root/reference/src/QAP.hs:521-524) for the large transforms; k_ntt_tile (kernels.hip.h) keeps
// the small ones.  Same pass descriptor (NttPass) and the same mathematics per pass -- bit-reversed
// placement, log2(S) radix-2 DIT stages with lazy add/sub, one closing multiplication -- but:
//
//   * a thread group of U = 2^(LP-2) lanes owns one column (S = 2^LP points; two columns of
//     2^(LP-1) points when the pass digit is odd: the spare top "position" bit is the column bit
//     and its stage is skipped).  Lane u keeps FOUR elements x[0..3] in VGPRs (36 registers).
//   * Round r works on extended-position bits (2r+1, 2r), which sit in the slot index: stage A
//     pairs slots (0,1),(2,3), stage B pairs (0,2),(1,3) -- both without leaving the registers.
//     S = 1024: 5 rounds = 4 exchanges instead of 10 LDS round trips with 10 workgroup barriers.
//   * Between rounds the slot digit is swapped with a two-bit field of the lane index.  Lane
//     order is chosen (logical index v = bitrev(u)) so that the FIRST exchange is the one that
//     crosses wavefronts (LDS + s_barrier) and every later one stays inside a wavefront: those
//     go through a wave-private LDS window with no barrier at all (DS operations of one wave
//     execute in order): the wavefront shuffle of the butterflies.  (A flavour on v_permlane32_swap /
//     v_permlane16_swap / DPP row rotations with no LDS at all was built and measured in round 2:
//     bit-exact, 0-3 % slower at every size because it spends VALU issue slots in a VALU-bound kernel;
//     profiles/r02_ntt.txt sections 3-4, code in git history before round 3.)  With v = bitrev(u) a
//     contiguous input row is also read in lane order.
//   * XOR-swizzled LDS addresses make both sides of every exchange bank-conflict free.
#pragma once
#include "ntt_pass.hip.h"

namespace acx {

__device__ __forceinline__ u32 rev2(u32 x) { return ((x & 1u) << 1) | (x >> 1); }

// loose (limbs < 2^29 + 8 after a carry pass, value < 64p) -> strictly normalised, < 2p, without a
// Montgomery multiplication: ripple the carries, estimate q = floor(top / (P[8] + 1)) <= x / p
// (q < 64), subtract q * p limb-wise (signed 64-bit column arithmetic), then one conditional
// subtraction of 2p covers the estimate's slack (remainder < 2p + p/2^20).
template <class F>
__device__ __forceinline__ Fe fe_reduce_loose(Fe a) {
    fe_carry(a);
    const u32 q = (u32)(((u64)a.l[kLimbs - 1] * (u64)((1ull << 32) / (F::P[kLimbs - 1] + 1))) >> 32);
    Fe r;
    long long c = 0;
#pragma unroll
    for (int k = 0; k < kLimbs; ++k) {
        const long long t = (long long)a.l[k] - (long long)((u64)q * F::P[k]) + c;
        r.l[k] = (k < kLimbs - 1) ? ((u32)t & kLimbMask) : (u32)t;
        c = t >> kLimbBits;
    }
    // r in [0, 3p) here (q may be one short of the exact quotient): bring it below 2p
    return fe_cond_sub<F::P2>(r);
}

// sub-transform twiddles in limb form: entry j = 3 x uint4 = the nine 29-bit limbs of w_S^j (strict) + padding.
// (Round 3 tried a second table operand -- w'' = w * (-p^-1) mod R, which turns the product into 151 multiplier
// instructions instead of 171: bit-exact, never faster, because the pass kernels sit at the 128-register ceiling of a
// 1024-thread workgroup; profiles/r03_ntt.txt, code at commit 8562ec3, model tools/model_mul_pre.py.)
__device__ __forceinline__ Fe fe_load_limbs(const uint4* __restrict__ tab, u64 idx) {
    const uint4 a = gload(tab + kLimbEntryQuads * idx), b = gload(tab + kLimbEntryQuads * idx + 1), c = gload(tab + kLimbEntryQuads * idx + 2);
    Fe r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
    return r;
}

// One round on the four slots: stage A pairs (0,1),(2,3) with twiddle wA, stage B pairs (0,2),(1,3) with
// twiddles wB0 / wB1.  Stage A leaves its sums uncarried (limbs < 2^31: fe_mul's left operand and the
// carrying add/sub of stage B take them); stage B carries.  TRIV (wave-uniform): wA = wB0 = 1, so those
// three products are replaced by the multiplication-free reduction.
template <class F, bool TRIV>
__device__ __forceinline__ void r4_round(Fe (&x)[4], const uint4* __restrict__ tw, u64 iA, u64 iB0, u64 iB1, bool stage_b) {
    {
        Fe t1, t3;
        if (TRIV) {
            t1 = fe_reduce_loose<F>(x[1]); t3 = fe_reduce_loose<F>(x[3]);
        } else {
            const Fe w = fe_load_limbs(tw, iA);
            t1 = fe_mul<F>(x[1], w); t3 = fe_mul<F>(x[3], w);
        }
        const Fe a0 = fe_add_lazy<false>(x[0], t1), s0 = fe_sub_lazy<F, false>(x[0], t1);
        const Fe a1 = fe_add_lazy<false>(x[2], t3), s1 = fe_sub_lazy<F, false>(x[2], t3);
        x[0] = a0; x[1] = s0; x[2] = a1; x[3] = s1;
    }
    // stage B's twiddle loads and products stay behind stage A: hoisted above it they cost the registers that sent
    // k_ntt_r4<*,10,2> to scratch memory (profiles/r04_ntt.txt: same speed, no spill)
    __builtin_amdgcn_sched_barrier(0);
    if (stage_b) {
        const Fe w1 = fe_load_limbs(tw, iB1);
        Fe t2;
        if (TRIV) t2 = fe_reduce_loose<F>(x[2]);
        else t2 = fe_mul<F>(x[2], fe_load_limbs(tw, iB0));
        const Fe t3 = fe_mul<F>(x[3], w1);
        const Fe a0 = fe_add_lazy(x[0], t2), s0 = fe_sub_lazy<F>(x[0], t2);
        const Fe a1 = fe_add_lazy(x[1], t3), s1 = fe_sub_lazy<F>(x[1], t3);
        x[0] = a0; x[2] = s0; x[1] = a1; x[3] = s1;
    }
}

// ---- the pass kernel ---------------------------------------------------------------------------------
// LP: even number of extended position bits of a thread group (sub-transform digit rounded up to
// even); LG: log2(thread groups per workgroup).  blockDim.x = 2^(LP-2+LG), tile = 2^(LP+LG) elements.
// The rounds are a real loop (one body in the instruction cache, ~1.5k instructions, instead of a
// 12k-instruction straight line), and so are the four closing multiplications (slots rotate through x[0]).
template <class F, int LP, int LG>
__global__ __launch_bounds__(1 << (LP - 2 + LG)) void k_ntt_r4(NttPass P) {
    constexpr int LU = LP - 2;
    constexpr u32 U = 1u << LU;
    constexpr int R = LP / 2;
    constexpr u32 ELEMS = 4u << (LU + LG);
    constexpr bool kUsesLds = R > 1;          // a single round exchanges nothing
    __shared__ u32 lds[kUsesLds ? kLimbs : 1][kUsesLds ? ELEMS : 1];

    const u32 ls = P.log_s;                  // LP == ls (even digit) or ls + 1 (odd digit: two columns per group)
    const u32 odd = (u32)LP - ls;
    const u32 S = 1u << ls;
    const u32 t = threadIdx.x, g = t >> LU, u = t & (U - 1);
    // logical lane index v = bitrev(u): holds the already-processed position bits.  Two instructions from u, and recomputed where
    // it is used (behind an optimisation barrier) instead of being held across the rounds: the instances of 1024 threads sit
    // at the 128-register ceiling, and this was the register that went to scratch memory.
    auto lane_v = [&]() { u32 uu = u; asm("" : "+v"(uu)); return __brev(uu) >> (32 - LU); };
    const u32 gbase = g * (4u * U);

    u64 base_in = 0, base_out = 0, K0 = 0, I0 = 0;
    {
        u64 tt = blockIdx.x;
        for (u32 d = 0; d < P.n_outer; ++d) {
            const u64 idx = tt % P.outer[d].count;
            tt /= P.outer[d].count;
            base_in += idx * P.outer[d].stride_in;
            base_out += idx * P.outer[d].stride_out;
            K0 += idx * P.outer[d].k_w;
            I0 += idx * P.outer[d].i_w;
        }
    }

    // ---- load: slot e of lane u = input point d = (rev2(e) << (ls-2)) | (u >> odd) of column 2g+(u&odd) | g
    Fe x[4];
    {
        const u32 col = (g << odd) | (u & odd);
        const u64 cbase = base_in + (u64)col * P.stride_c_in;
        const u32 dlow = u >> odd, smask = (1u << P.split_in) - 1u;
        uint4 raw[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32 d = (rev2(e) << (ls - 2)) | dlow;
            const uint4* p = P.src + 2 * (cbase + (u64)(d & smask) * P.stride_t_in + (u64)(d >> P.split_in) * P.stride_t_in_hi);
            raw[2 * e] = gload(p);
            raw[2 * e + 1] = gload(p + 1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32 w[8] = {raw[2 * e].x, raw[2 * e].y, raw[2 * e].z, raw[2 * e].w,
                              raw[2 * e + 1].x, raw[2 * e + 1].y, raw[2 * e + 1].z, raw[2 * e + 1].w};
            x[e] = fe_unpack(w);
        }
        if (P.scale_on_load) {               // coset pre-multiplication of a forward transform
#pragma unroll 1
            for (u32 e = 0; e < 4; ++e) {
                const u32 d = (rev2(e) << (ls - 2)) | dlow;
                Fe f;
                if (P.scale_on_load == 2) {  // by the transform digit alone, from a direct table of S entries (distributed steps)
                    f = fe_load(P.sc_lo + 2 * (u64)d);
                } else if (P.scale_on_load == 3) {  // pointwise product of two vectors on the way in (h(x): L * R on the coset)
                    f = fe_load(P.mul_src + 2 * (cbase + (u64)(d & smask) * P.stride_t_in + (u64)(d >> P.split_in) * P.stride_t_in_hi));
                } else {
                    const u64 off = cbase + (u64)(d & smask) * P.stride_t_in + (u64)(d >> P.split_in) * P.stride_t_in_hi;
                    const u64 ex = P.e_mode ? (P.e_base + (u64)d * P.e_t + (P.i_base + I0 + (u64)col * P.c_iw) * P.e_c) : (off & P.idx_mask);
                    f = two_level_pow<F>(P.sc_lo, P.sc_hi, ex);
                }
                const Fe y = fe_mul<F>(x[0], f);
                x[0] = x[1]; x[1] = x[2]; x[2] = x[3]; x[3] = y;
            }
        }
    }

    // ---- round 0: stage A has w = 1 on strict inputs (no multiplication), stage B has w = 1 and w_4
    {
        const Fe a0 = fe_add_lazy<false>(x[0], x[1]), s0 = fe_sub_lazy<F, false>(x[0], x[1]);
        const Fe a1 = fe_add_lazy<false>(x[2], x[3]), s1 = fe_sub_lazy<F, false>(x[2], x[3]);
        x[0] = a0; x[1] = s0; x[2] = a1; x[3] = s1;
        if (!(R == 1 && odd)) {
            // pair (0,2): w = 1 and x[2] is the uncarried sum of two strict values: it is subtracted as it is, against
            // the fat form of 8p (value grows by 8p once per pass: 12p after this round, < 64p after 12 stages)
            const Fe t3 = fe_mul<F>(x[3], fe_load_limbs(P.sub_tw, (u64)(S >> 2)));
            const Fe b0 = fe_add_lazy(x[0], x[2]), d0 = fe_sub_fat<F::P8FAT>(x[0], x[2]);
            const Fe b1 = fe_add_lazy(x[1], t3), d1 = fe_sub_lazy<F>(x[1], t3);
            x[0] = b0; x[2] = d0; x[1] = b1; x[3] = d1;
        }
    }

    // ---- exchange + round r, r = 1 .. R-1
#pragma unroll 1
    for (int r = 1; r < R; ++r) {
        // exchange: slot digit <-> lane field at physical bits (phi+1, phi)
        const int phi = LP - 2 - 2 * r;
        const bool cross = phi > 4;
        {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 a = gbase + (u32)e * U + (u ^ (rev2(e) << phi));
#pragma unroll
                for (int k = 0; k < kLimbs; ++k) lds[kUsesLds ? k : 0][kUsesLds ? a : 0] = x[e].l[k];
            }
            if (cross) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            const u32 pf = (u >> phi) & 3u;
            const u32 rbase = gbase + rev2(pf) * U + (u & ~(3u << phi));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 a = rbase + ((rev2(e) ^ pf) << phi);
#pragma unroll
                for (int k = 0; k < kLimbs; ++k) x[e].l[k] = lds[kUsesLds ? k : 0][kUsesLds ? a : 0];
            }
            // An in-wave exchange only touches its wave's own window and DS operations of a wave execute
            // in order; after a cross-wave exchange other waves may still be reading this wave's window
            // when it writes again, so that case needs the second barrier.
            if (cross) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        }
        // round r: twiddle exponents from the processed position bits jlow = v mod 4^r
        const u32 jlow = lane_v() & ((1u << (2 * r)) - 1u);
        const bool stage_b = !(r == R - 1 && odd);
        const u64 iA = (u64)jlow << (ls - 1 - 2 * r);
        const u64 iB0 = stage_b ? ((u64)jlow << (ls - 2 - 2 * r)) : 0;
        if (__ballot(jlow != 0) == 0) r4_round<F, true>(x, P.sub_tw, iA, iB0, iB0 + (S >> 2), stage_b);
        else r4_round<F, false>(x, P.sub_tw, iA, iB0, iB0 + (S >> 2), stage_b);
    }

    // ---- closing: inter-pass twiddle / scale / coset factor (one multiplication) or plain reduction; store.
    // Slot e = output digit (e << LU) | v; the slots rotate through x[0] so that the loop body exists once.
    const Fe scale = fe_from_arg(P.scale);
    const u32 v = lane_v();
#pragma unroll 1
    for (u32 e = 0; e < 4; ++e) {
        const Fe cur = x[0];
        x[0] = x[1]; x[1] = x[2]; x[2] = x[3];
        const u32 col = odd ? ((g << 1) | (e >> 1)) : g;
        const u32 kd = odd ? (((e & 1u) << LU) | v) : ((e << LU) | v);
        const u64 off = base_out + (u64)(kd & ((1u << P.split_out) - 1u)) * P.stride_t_out +
                        (u64)(kd >> P.split_out) * P.stride_t_out_hi + (u64)col * P.stride_c_out;
        Fe f = scale;
        bool mul = true;
        if (P.tw_mode == 3) {                                // the closing factor of every element from a table in STORE order
            f = fe_load(P.tw_lo + 2 * off);
        } else if (P.tw_mode == 1) {
            const u64 K = P.k_base + K0 + (u64)kd * P.t_kw + (u64)col * P.c_kw, I = P.i_base + I0 + (u64)col * P.c_iw;
            f = fe_load(P.tw_lo + 2 * ((I * K) >> P.tw_shift));
        } else if (P.tw_mode == 2 || P.scale_mode == 2) {
            const u64 K = P.k_base + K0 + (u64)kd * P.t_kw + (u64)col * P.c_kw, I = P.i_base + I0 + (u64)col * P.c_iw;
            const bool tw = P.tw_mode == 2;
            const u64 ex = P.e_mode ? (P.e_base + (u64)kd * P.e_t + I * P.e_c) : (off & P.idx_mask);
            f = two_level_pow<F>(tw ? P.tw_lo : P.sc_lo, tw ? P.tw_hi : P.sc_hi, tw ? ((I * K) & P.tw_mask) : ex);
        } else if (P.scale_mode == 3 && P.scale_off_end != 0 && off >= P.scale_off_end) {
            mul = false;                                     // a vector of the batch that does not take the coset factor
        } else if (P.scale_mode == 3) {                      // direct coset table: g^(element index), 1/N included
            const u64 I = P.i_base + I0 + (u64)col * P.c_iw;
            f = fe_load(P.sc_lo + 2 * (P.e_mode ? (P.e_base + (u64)kd * P.e_t + I * P.e_c) : (off & P.idx_mask)));
        } else if (P.scale_mode == 0) {
            mul = false;
        }
        Fe y;
        if (mul) y = fe_mul<F>(cur, f); else y = fe_reduce_loose<F>(cur);
        if (P.add_src != nullptr) y = fe_add<F>(y, fe_load(P.add_src + 2 * off));     // uniform
        fe_store(P.dst + 2 * off, y);
    }
}

}  // namespace acx
