// ntt_r4.hip -- the instances of k_ntt_r4 (ntt_r4.hip.h) and their launcher, in a translation unit of their own:
//   * it is compiled with `-mllvm -misched=gcn-iterative-minreg` (build.py): with the default scheduler the pass kernels sit
//     at the 128-VGPR ceiling of a 1024-thread workgroup and the BLS12-381 instances spill; the register-minimising
//     scheduler holds every instance spill-free at 120-126 VGPRs WITH fe_mul_pre's second table operand
//     (tools/kres.sh, profiles/r03_ntt.txt);
//   * the two translation units of libacx.so compile in parallel.
#include <hip/hip_runtime.h>

#include "field_consts.h"
#include "ntt_r4.hip.h"

namespace acx {

template <class F>
static bool launch_r4(int lp, int lg, unsigned tiles, hipStream_t st, const NttPass& Q) {
#define ACX_R4_CASE(LP_, LG_)                                                                              \
    if (lp == LP_ && lg == LG_) {                                                                         \
        hipLaunchKernelGGL((k_ntt_r4<F, LP_, LG_>), dim3(tiles), dim3(1u << (LP_ - 2 + LG_)), 0, st, Q);   \
        return true;                                                                                      \
    }
    ACX_R4_CASE(6, 0) ACX_R4_CASE(6, 2) ACX_R4_CASE(6, 4)
    ACX_R4_CASE(8, 0) ACX_R4_CASE(8, 2)
    ACX_R4_CASE(10, 0) ACX_R4_CASE(10, 1) ACX_R4_CASE(10, 2)
    ACX_R4_CASE(12, 0)
#undef ACX_R4_CASE
    return false;
}

// (LP, LG) = (even number of position bits of a thread group, log2 of the thread groups per workgroup); false: no such instance
bool launch_ntt_r4(bool bls12_381, int lp, int lg, unsigned tiles, hipStream_t st, const NttPass& Q) {
    return bls12_381 ? launch_r4<Bls12381Fr>(lp, lg, tiles, st, Q) : launch_r4<Bn254Fr>(lp, lg, tiles, st, Q);
}

}  // namespace acx
