// oracle/c/ff.hpp — TEST ORACLE ONLY (CPU restatement; never linked into libzkhip).
//
// Montgomery prime fields with 64-bit limbs + Fq2, restating [UPSTREAM] ark-ff 0.3.0 `Fp256`/`Fp384`
// (generic, non-asm backend: `ark-ff-asm` is not enabled, /root/reference/zokrates_ark/Cargo.toml:24-36)
// as summarised in SURVEY.md App. A.1.  Elements are little-endian u64 limbs in Montgomery form,
// R = 2^(64*N).  Only the modulus is given; R, R^2 and -p^{-1} mod 2^64 are derived at start-up.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace orc {
typedef uint64_t u64;
typedef unsigned __int128 u128;

template <int N_>
struct FieldParams {
    static constexpr int N = N_;
    u64 mod[N_];
    u64 r[N_];    // R mod p
    u64 r2[N_];   // R^2 mod p
    u64 inv;      // -p^{-1} mod 2^64
    int bits;

    static bool geq(const u64* a, const u64* b) {
        for (int i = N_ - 1; i >= 0; --i) {
            if (a[i] != b[i]) return a[i] > b[i];
        }
        return true;
    }
    static u64 sub_n(u64* a, const u64* b) {  // a -= b, returns borrow
        u64 br = 0;
        for (int i = 0; i < N_; ++i) {
            u128 d = (u128)a[i] - b[i] - br;
            a[i] = (u64)d;
            br = (u64)(d >> 64) & 1;
        }
        return br;
    }
    void dbl_mod(u64* a) const {  // a = 2a mod p (a < p)
        u64 c = 0;
        for (int i = 0; i < N_; ++i) {
            u64 n = (a[i] << 1) | c;
            c = a[i] >> 63;
            a[i] = n;
        }
        if (c || geq(a, mod)) sub_n(a, mod);
    }
    explicit FieldParams(const u64 (&m)[N_]) {
        memcpy(mod, m, sizeof(mod));
        u64 x = 1;  // Newton: x = p^{-1} mod 2^64
        for (int i = 0; i < 6; ++i) x *= 2 - mod[0] * x;
        inv = (u64)0 - x;
        u64 t[N_] = {1};
        for (int i = 0; i < 64 * N_; ++i) dbl_mod(t);
        memcpy(r, t, sizeof(r));
        for (int i = 0; i < 64 * N_; ++i) dbl_mod(t);
        memcpy(r2, t, sizeof(r2));
        bits = 64 * N_;
        while (!((mod[(bits - 1) / 64] >> ((bits - 1) % 64)) & 1)) --bits;
    }
};

// Tag types select the parameter instance.
template <class Tag>
struct Fp {
    static constexpr int N = Tag::N;
    static constexpr int BYTES = 8 * Tag::N;
    u64 v[Tag::N];

    static const FieldParams<Tag::N>& P() { return Tag::params(); }

    static Fp zero() { Fp x; memset(x.v, 0, sizeof(x.v)); return x; }
    static Fp one() { Fp x; memcpy(x.v, P().r, sizeof(x.v)); return x; }
    bool is_zero() const { u64 a = 0; for (int i = 0; i < N; ++i) a |= v[i]; return a == 0; }
    bool operator==(const Fp& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }

    Fp operator+(const Fp& o) const {
        Fp x; u64 c = 0;
        for (int i = 0; i < N; ++i) { u128 s = (u128)v[i] + o.v[i] + c; x.v[i] = (u64)s; c = (u64)(s >> 64); }
        if (c || FieldParams<N>::geq(x.v, P().mod)) FieldParams<N>::sub_n(x.v, P().mod);
        return x;
    }
    Fp operator-(const Fp& o) const {
        Fp x = *this;
        if (FieldParams<N>::sub_n(x.v, o.v)) {
            u64 c = 0;
            for (int i = 0; i < N; ++i) { u128 s = (u128)x.v[i] + P().mod[i] + c; x.v[i] = (u64)s; c = (u64)(s >> 64); }
        }
        return x;
    }
    Fp neg() const { return zero() - *this; }
    Fp dbl() const { return *this + *this; }

    Fp operator*(const Fp& o) const {  // CIOS Montgomery product
        const auto& p = P();
        u64 t[N + 2];
        memset(t, 0, sizeof(t));
        for (int i = 0; i < N; ++i) {
            u128 c = 0;
            for (int j = 0; j < N; ++j) { c += (u128)v[j] * o.v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
            c += t[N]; t[N] = (u64)c; t[N + 1] = (u64)(c >> 64);
            u64 m = t[0] * p.inv;
            c = (u128)m * p.mod[0] + t[0]; c >>= 64;
            for (int j = 1; j < N; ++j) { c += (u128)m * p.mod[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
            c += t[N]; t[N - 1] = (u64)c; t[N] = t[N + 1] + (u64)(c >> 64);
        }
        Fp x; memcpy(x.v, t, sizeof(x.v));
        if (t[N] || FieldParams<N>::geq(x.v, p.mod)) FieldParams<N>::sub_n(x.v, p.mod);
        return x;
    }
    Fp sqr() const { return *this * *this; }

    Fp pow_limbs(const u64* e, int n) const {
        Fp r = one();
        for (int i = n * 64 - 1; i >= 0; --i) {
            r = r.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) r = r * *this;
        }
        return r;
    }
    Fp pow_u64(u64 e) const { return pow_limbs(&e, 1); }
    Fp inverse() const {  // Fermat; 0 -> 0
        u64 e[N]; memcpy(e, P().mod, sizeof(e));
        u64 two[N] = {2};
        FieldParams<N>::sub_n(e, two);
        return pow_limbs(e, N);
    }

    // canonical little-endian bytes <-> Montgomery (same bytes as ark `ToBytes`/`FromBytes`)
    static Fp from_canonical_limbs(const u64* c) {
        Fp x, r2; memcpy(x.v, c, sizeof(x.v)); memcpy(r2.v, P().r2, sizeof(r2.v));
        return x * r2;
    }
    void to_canonical_limbs(u64* out) const {
        Fp o; memset(o.v, 0, sizeof(o.v)); o.v[0] = 1;
        Fp c = *this * o;
        memcpy(out, c.v, sizeof(c.v));
    }
    static Fp from_bytes(const uint8_t* b) { u64 c[N]; memcpy(c, b, sizeof(c)); return from_canonical_limbs(c); }
    void to_bytes(uint8_t* b) const { u64 c[N]; to_canonical_limbs(c); memcpy(b, c, sizeof(c)); }
    static Fp from_u64(u64 x) { u64 c[N] = {x}; return from_canonical_limbs(c); }
    static bool canonical_in_range(const uint8_t* b) {
        u64 c[N]; memcpy(c, b, sizeof(c));
        return !FieldParams<N>::geq(c, P().mod);
    }
};

// Fq2 = Fq[u]/(u^2 + 1)  (non-residue -1 for both BN254 and BLS12-381)
template <class F>
struct Fp2 {
    static constexpr int BYTES = 2 * F::BYTES;
    F c0, c1;
    static Fp2 zero() { return {F::zero(), F::zero()}; }
    static Fp2 one() { return {F::one(), F::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fp2& o) const { return !(*this == o); }
    Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2 operator*(const Fp2& o) const {
        F a = c0 * o.c0, b = c1 * o.c1;
        F s = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, s - a - b};
    }
    Fp2 sqr() const {
        F a = (c0 + c1) * (c0 - c1);
        F b = c0 * c1;
        return {a, b.dbl()};
    }
    Fp2 inverse() const {
        F n = (c0.sqr() + c1.sqr()).inverse();
        return {c0 * n, (c1 * n).neg()};
    }
    static Fp2 from_bytes(const uint8_t* b) { return {F::from_bytes(b), F::from_bytes(b + F::BYTES)}; }
    void to_bytes(uint8_t* b) const { c0.to_bytes(b); c1.to_bytes(b + F::BYTES); }
};

}  // namespace orc
