// host_field.h -- host-side prime-field arithmetic (4 x 64-bit limbs, Montgomery radix 2^256).
//
// Used by the HOST logic of libacx only: marshalling (`affineCircuitToAffineMap`,
// /root/reference/src/Circuit/Affine.hs:90-105), the sequential witness generator
// (`evalGate`, src/Circuit/Arithmetic.hs:106-145), root ordering and the handful of scalar
// constants (omega_k, N^-1, coset factors) handed to kernels.  It is not a fallback for any
// device kernel: every bulk operation of the hot path runs on the GPU.
#pragma once
#include <cstdint>
#include <cstring>
#include "field_consts.h"

namespace acx {

struct H256 {
    uint64_t l[4];
    bool operator==(const H256& o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const H256& o) const { return !(*this == o); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
};

inline int h256_cmp(const H256& a, const H256& b) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
    }
    return 0;
}

// Runtime-parametrised field (the field is chosen by enum at context creation).
class HostField {
public:
    template <class F>
    static HostField make() {
        HostField f;
        std::memcpy(f.p_.l, F::P64, 32);
        std::memcpy(f.one_.l, F::R1_64, 32);
        std::memcpy(f.r2_.l, F::R2_64, 32);
        f.n0_ = F::N064;
        f.two_adicity_ = F::kTwoAdicity;
        H256 w;
        std::memcpy(w.l, F::OMEGA64, 32);
        f.omega_max_ = f.to_mont(w);
        H256 g = {{F::kGenerator, 0, 0, 0}};
        f.gen_ = f.to_mont(g);
        return f;
    }

    const H256& modulus() const { return p_; }
    int two_adicity() const { return two_adicity_; }
    H256 zero() const { return H256{{0, 0, 0, 0}}; }
    H256 one() const { return one_; }                 // Montgomery 1
    const H256& omega_max() const { return omega_max_; }  // Montgomery
    const H256& generator() const { return gen_; }        // Montgomery
    void set_omega_max(const H256& mont, int two_adicity) { omega_max_ = mont; two_adicity_ = two_adicity; }

    bool is_canonical(const H256& a) const { return h256_cmp(a, p_) < 0; }

    H256 add(const H256& a, const H256& b) const {
        H256 t;
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; t.l[i] = (uint64_t)c; c >>= 64; }
        if (c || h256_cmp(t, p_) >= 0) sub_raw(t, t, p_);
        return t;
    }
    H256 sub(const H256& a, const H256& b) const {
        H256 t;
        if (sub_raw(t, a, b)) add_raw(t, t, p_);
        return t;
    }
    H256 neg(const H256& a) const { return sub(zero(), a); }
    // Montgomery product (operand scanning, coarsely integrated)
    H256 mul(const H256& a, const H256& b) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c; c >>= 64;
            }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * n0_;
            c = (unsigned __int128)m * p_.l[0] + t[0]; c >>= 64;
            for (int j = 1; j < 4; ++j) {
                c += (unsigned __int128)m * p_.l[j] + t[j];
                t[j - 1] = (uint64_t)c; c >>= 64;
            }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        H256 r = {{t[0], t[1], t[2], t[3]}};
        if (t[4] || h256_cmp(r, p_) >= 0) sub_raw(r, r, p_);
        return r;
    }
    H256 to_mont(const H256& canonical) const { return mul(canonical, r2_); }
    H256 from_mont(const H256& mont) const { return mul(mont, H256{{1, 0, 0, 0}}); }
    H256 from_u64(uint64_t v) const { return to_mont(H256{{v, 0, 0, 0}}); }
    H256 pow(H256 base, const H256& e) const {
        H256 acc = one_;
        for (int i = 0; i < 256; ++i) {
            if ((e.l[i / 64] >> (i % 64)) & 1) acc = mul(acc, base);
            base = mul(base, base);
        }
        return acc;
    }
    H256 pow_u64(H256 base, uint64_t e) const { return pow(base, H256{{e, 0, 0, 0}}); }
    H256 inv(const H256& a) const {  // a^(p-2); inv(0) = 0
        H256 e = p_;
        H256 two = {{2, 0, 0, 0}};
        sub_raw(e, e, two);
        return pow(a, e);
    }
    // Montgomery form of the primitive 2^k-th root of unity (pairing `getRootOfUnity k`).
    H256 root_of_unity(int k) const {
        H256 w = omega_max_;
        for (int i = k; i < two_adicity_; ++i) w = mul(w, w);
        return w;
    }
    // Device element of a host Montgomery value as it lies in device MEMORY: x * 2^261 mod p, 32 bytes little endian.
    H256 to_dev_word(const H256& mont) const {
        H256 y = mont;
        for (int i = 0; i < 5; ++i) y = add(y, y);
        return y;
    }
    // Device element (lazy Montgomery radix 2^261) of a host Montgomery value, as 9 x 29-bit limbs.
    void to_dev_limbs(const H256& mont, uint32_t out[9]) const {
        // stored integer of mont = x*2^256 mod p; x*2^261 mod p = 32 * that (mod p)
        H256 y = mont;
        for (int i = 0; i < 5; ++i) y = add(y, y);
        for (int k = 0; k < 9; ++k) {
            const int bit = 29 * k, wi = bit >> 6, off = bit & 63;
            uint64_t v = y.l[wi] >> off;
            if (off + 29 > 64 && wi + 1 < 4) v |= y.l[wi + 1] << (64 - off);
            out[k] = (uint32_t)v & 0x1fffffffu;
        }
    }

private:
    static uint64_t add_raw(H256& o, const H256& a, const H256& b) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; o.l[i] = (uint64_t)c; c >>= 64; }
        return (uint64_t)c;
    }
    static uint64_t sub_raw(H256& o, const H256& a, const H256& b) {
        uint64_t borrow = 0;
        for (int i = 0; i < 4; ++i) {
            const unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - borrow;
            o.l[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        return borrow;
    }
    H256 p_, one_, r2_, omega_max_, gen_;
    uint64_t n0_ = 0;
    int two_adicity_ = 0;
};

}  // namespace acx
