// ntt_r4.hip -- the instances of k_ntt_r4 (ntt_r4.hip.h) and their launcher, in a translation unit of their own: the two
// units of libacx.so compile in parallel (27 s instead of 74 s), and the pass kernels can be given their own device-side
// LLVM options (build.py, ACX_NTT_MISCHED: how the scheduler experiments of profiles/r03_ntt.txt were built).
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
