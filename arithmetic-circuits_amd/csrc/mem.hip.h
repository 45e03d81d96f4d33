// mem.hip.h -- address-space-pinned memory accesses shared by every kernel header.
#pragma once
#include "fr.hip.h"

namespace acx {

// Pointers that reach a kernel through a descriptor in memory (SellSystem) have no known address
// space and hipcc emits flat_load for them (counted against lgkmcnt as well as vmcnt, and split into
// odd 4/16/12-byte pieces for the 32-byte gathers).  These helpers pin the global address space
// and the access width: one global_load_dwordx4 / dwordx2 per call.
typedef __attribute__((address_space(1))) const uint4 g_uint4;
typedef __attribute__((address_space(1))) const uint2 g_uint2;
typedef __attribute__((address_space(1))) const u32 g_u32;
typedef u32 v2u32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const v4u32 g_v4u32;
typedef __attribute__((address_space(1))) const v2u32 g_v2u32;

__device__ __forceinline__ uint4 gload(const uint4* p) {
    const v4u32 r = *(g_v4u32*)p;
    return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ uint2 gload(const uint2* p) {
    const v2u32 r = *(g_v2u32*)p;
    return make_uint2(r.x, r.y);
}
__device__ __forceinline__ u32 gload(const u32* p) { return *(g_u32*)p; }
// a wave-uniform word through the scalar cache (constant address space: s_load_dword)
typedef __attribute__((address_space(4))) const u32 c_u32;
__device__ __forceinline__ u32 sload(const u32* p) { return *(c_u32*)(unsigned long long)p; }
// The constraint stream is read exactly once per verification: non-temporal loads keep it from
// evicting the witness window (re-read by every row) out of the XCD's L2.
__device__ __forceinline__ uint4 nt_load(const uint4* p) {
    const v4u32 r = __builtin_nontemporal_load((g_v4u32*)p);
    return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ uint2 nt_load(const uint2* p) {
    const v2u32 r = __builtin_nontemporal_load((g_v2u32*)p);
    return make_uint2(r.x, r.y);
}
// a wave-uniform element through the scalar cache: one s_load_dwordx8, unpacked on the scalar unit (the limbs stay in SGPRs)
typedef u32 v8u32 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const v8u32 c_v8u32;
__device__ __forceinline__ Fe fe_sload(const uint4* p) {
    const v8u32 r = *(c_v8u32*)(unsigned long long)p;
    const u32 w[8] = {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]};
    return fe_unpack(w);
}
__device__ __forceinline__ Fe fe_gload(const uint4* p) {
    const uint4 lo = gload(p), hi = gload(p + 1);
    const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return fe_unpack(w);
}

}  // namespace acx
