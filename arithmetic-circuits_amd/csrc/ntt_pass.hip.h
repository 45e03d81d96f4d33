// ntt_pass.hip.h -- the pass descriptor shared by the NTT kernels (k_ntt_tile in kernels.hip.h, k_ntt_r4 in ntt_r4.hip.h /
// ntt_r4.hip) and the planner in engine.hip.
#pragma once
#include "mem.hip.h"

namespace acx {

constexpr int kLimbEntryQuads = 3;     // uint4 words per entry of a limb-form twiddle table (k_pow_table_limbs): 9 limbs + padding

// ---- K3/K4: tiled multi-pass NTT ---------------------------------------------------------------
// A length-N transform is factored N = N_1 * ... * N_P (P <= 4, every N_p <= 256).  Pass p runs
// all the length-N_p sub-transforms over digit p of the index; a workgroup owns a tile of
// S = N_p points x T columns (S*T = 1024 elements, 36 KiB of LDS in limb-plane form, so four
// workgroups share a CU and one tile's global load/store overlaps the others' butterflies).
// T consecutive elements of the fastest-varying remaining digit form a 32*T-byte segment: every
// global access is a full 128/256-byte line.  Inside the tile: bit-reversed placement on load,
// log2(S) radix-2 DIT stages out of LDS with lazy (carry-only) add/sub, then ONE multiplication
// per element that both applies the inter-pass twiddle w_N^(I*K) (or the final 1/N, coset factor)
// and brings the lazily grown value back below 2p.  The last pass stores in natural order, so
// there is no separate transpose or bit-reversal kernel and no barrier between workgroups.
constexpr int kTileElems = 1024;
constexpr int kMaxOuter = 4;

struct NttOuter {          // one outer loop dimension of the tile enumeration
    u32 count;             // number of values
    u32 pad;
    u64 stride_in, stride_out;   // element strides
    u64 k_w, i_w;          // contribution of this index to the twiddle factors K and I
};

struct NttPass {
    const uint4* src;
    uint4* dst;
    const uint4* sub_tw;   // w_S^j, j < S/2 (dev format, strictly normalised)
    const uint4* tw_lo;    // twiddle table: direct (w_M^e, e < M) or low level of a two-level table
    const uint4* tw_hi;    // high level (w^(1024 j)) or null
    const uint4* sc_lo;    // coset powers g^j (j < 1024) or null
    const uint4* sc_hi;    // g^(1024 j) or null
    const uint4* mul_src;  // (k_ntt_r4, scale_on_load 3) second input vector: the pass transforms src[i] * mul_src[i]
    const uint4* add_src;  // (k_ntt_r4, last pass) null, or a vector added to the output after the closing step: dst[k] = closing(X[k]) + add_src[k]
    u32 log_s, log_t;      // S points, T columns
    u32 n_outer;
    u32 tw_mode;           // 0 none, 1 direct table index (I*K) >> tw_shift, 2 two-level on (I*K) & tw_mask, 3 (k_ntt_r4) table in store order: tw_lo[output offset]
    u32 tw_shift;
    u32 scale_mode;        // 0 none, 1 multiply by `scale`, 2 scale * g^(element index) via sc_lo/sc_hi, 3 (k_ntt_r4) g^(index) from the direct table sc_lo
    u32 scale_on_load;     // coset pre-multiplication of a forward transform (first pass): 1 two-level by element index, 2 (k_ntt_r4) sc_lo[transform digit], 3 (k_ntt_r4) the element of mul_src at the same offset
    u32 pad;
    u64 scale_off_end;     // scale_mode 3 (k_ntt_r4): outputs at offsets >= this take the plain reduction (0 = no bound): the
                           // leading vectors of a batch get the coset factor, the rest do not (h(x): L and R, not O)
    u64 tw_mask;
    u64 stride_t_in, stride_t_out;   // transform direction
    u64 stride_c_in, stride_c_out;   // column direction
    u64 t_kw;              // K contribution of the output digit k_p
    u64 c_kw, c_iw;        // K / I contribution of the column index
    u64 idx_mask;          // element index within its transform = offset & idx_mask (coset exponent)
    // --- used by k_ntt_r4 only (local steps of the distributed four-step transform, acx_ntt_dist_step_dev) ---
    // transform-direction offset of digit d = (d & (2^split - 1)) * stride_t + (d >> split) * stride_t_hi;
    // the planner's default split = 0, stride_t_hi = stride_t is the plain single stride.
    u32 split_in, split_out;
    u64 stride_t_in_hi, stride_t_out_hi;
    u64 k_base, i_base;    // constants added to the twiddle factors K and I (this rank's block offset)
    u32 e_mode;            // coset exponent: 0 = offset & idx_mask, 1 = e_base + digit * e_t + I * e_c (I = the column's global index)
    u32 pad2;
    u64 e_base, e_t, e_c;
    NttOuter outer[kMaxOuter];
    FeArg scale;
};

template <class F>
__device__ __forceinline__ Fe two_level_pow(const uint4* __restrict__ lo, const uint4* __restrict__ hi, u64 e) {
    const Fe a = fe_load(lo + 2 * (e & 1023));
    if (hi == nullptr) return a;
    return fe_mul<F>(a, fe_load(hi + 2 * (e >> 10)));
}

}  // namespace acx
