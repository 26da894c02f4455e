// field.cuh — Montgomery prime-field arithmetic for the Groth16 hot path (product code).
//
// Replaces, on the device, what the reference gets from [UPSTREAM] ark-ff 0.3.0 `Fp256`/`Fp384`
// through `zokrates_field` (/root/reference/zokrates_field/src/lib.rs:56-62 `ArkFieldExtensions`,
// curve modules /root/reference/zokrates_field/src/bn128.rs:1-13, bls12_381.rs:1-13).
//
// CDNA4 has no 64x64 multiplier: a wide product is a chain of `v_mad_u64_u32`
// (32x32+64 -> 64).  Elements are therefore 8 (or 12) 32-bit limbs, little-endian, Montgomery
// form with R = 2^(32*N) — the same bytes as ark's 4x/6x u64 Montgomery limbs, so one 32-byte HBM
// load (2 x dwordx4) moves one element.  All loops are fully unrolled so limbs live in VGPRs.
// The same code compiles for the host (final proof assembly, key preparation, CPU-side tests).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
// big, cold routines (general add, doubling, inversion, scalar ladders) are real calls on the device:
// inlining them everywhere makes kernels of 10^5 instructions that take minutes to register-allocate.
#define ZK_HD_CALL __host__ __device__ __noinline__
#define ZK_UNROLL _Pragma("unroll")
#else
#define ZK_HD inline
#define ZK_HD_CALL inline
#define ZK_UNROLL
#endif

namespace zk {
typedef uint32_t u32;
typedef uint64_t u64;

// The order a transform leaves its output in ("sigma order"; kernels_ntt.cuh).  Two passes, N = n1 * n2 (n3 = 1): position
// k1 * n2 + k2 holds the element of natural index k1 + n1 * k2.  Three passes, N = n1 * n2 * n3 (domains above 2^22): position
// (k1 * n2 + k2) * n3 + k3 holds natural index k1 + n1 * (k2 + n2 * k3) — the same rule applied again inside the n2 * n3 block
// of every k1.  The h bases of a key are stored in this order (k_sigma_gather_points), so the prover never permutes.
ZK_HD u64 sigma_nat(u64 p, u32 n1, u32 n2, u32 n3) {
    const u64 m = (u64)n2 * n3;
    const u64 k1 = p / m, rem = p % m;
    const u64 k2 = rem / n3, k3 = rem % n3;
    return k1 + (u64)n1 * (k2 + (u64)n2 * k3);
}

// ---- field parameter packs (constexpr tables readable from host and device code) ----
#define ZK_TABLE(name, n, ...) \
    ZK_HD static constexpr u32 name(int i) { constexpr u32 t[n] = {__VA_ARGS__}; return t[i]; }

struct Bn254Fr {
    static constexpr int N = 8;
    static constexpr int BITS = 254;
    static constexpr u32 INV = 0xefffffffu;
    ZK_TABLE(mod, 8, 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u)
    ZK_TABLE(r1, 8, 0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u)
    ZK_TABLE(r2, 8, 0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u)
};
struct Bn254Fq {
    static constexpr int N = 8;
    static constexpr int BITS = 254;
    static constexpr u32 INV = 0xe4866389u;
    ZK_TABLE(mod, 8, 0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u)
    ZK_TABLE(r1, 8, 0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u)
    ZK_TABLE(r2, 8, 0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u)
};
struct Bls381Fr {
    static constexpr int N = 8;
    static constexpr int BITS = 255;
    static constexpr u32 INV = 0xffffffffu;
    ZK_TABLE(mod, 8, 0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u)
    ZK_TABLE(r1, 8, 0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u)
    ZK_TABLE(r2, 8, 0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u)
};
struct Bls381Fq {
    static constexpr int N = 12;
    static constexpr int BITS = 381;
    static constexpr u32 INV = 0xfffcfffdu;
    ZK_TABLE(mod, 12, 0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
             0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau)
    ZK_TABLE(r1, 12, 0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
             0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u)
    ZK_TABLE(r2, 12, 0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
             0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u)
};

// ---- element type ----
template <class P>
struct Fe {
    static constexpr int N = P::N;
    static constexpr int BYTES = 4 * P::N;
    typedef P Params;
    u32 v[P::N];

    ZK_HD static Fe zero() {
        Fe r;
        ZK_UNROLL for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fe one() {  // Montgomery one = R mod p
        Fe r;
        ZK_UNROLL for (int i = 0; i < N; ++i) r.v[i] = P::r1(i);
        return r;
    }
    ZK_HD static Fe r2() {
        Fe r;
        ZK_UNROLL for (int i = 0; i < N; ++i) r.v[i] = P::r2(i);
        return r;
    }
    ZK_HD bool is_zero() const {
        u32 a = 0;
        ZK_UNROLL for (int i = 0; i < N; ++i) a |= v[i];
        return a == 0;
    }
    ZK_HD bool equals(const Fe& o) const {
        u32 a = 0;
        ZK_UNROLL for (int i = 0; i < N; ++i) a |= v[i] ^ o.v[i];
        return a == 0;
    }
};

// r = (a >= p) ? a - p : a      (a < 2p)
template <class P>
ZK_HD void fe_reduce_once(Fe<P>& a) {
    u32 d[P::N];
    u64 bw = 0;
    ZK_UNROLL for (int i = 0; i < P::N; ++i) {
        u64 t = (u64)a.v[i] - P::mod(i) - bw;
        d[i] = (u32)t;
        bw = t >> 63;
    }
    ZK_UNROLL for (int i = 0; i < P::N; ++i) a.v[i] = bw ? a.v[i] : d[i];
}

template <class P>
ZK_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    u64 c = 0;
    ZK_UNROLL for (int i = 0; i < P::N; ++i) {  // every modulus here leaves >= 1 spare top bit: no carry out
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    fe_reduce_once(r);
    return r;
}

template <class P>
ZK_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    u64 bw = 0;
    ZK_UNROLL for (int i = 0; i < P::N; ++i) {
        u64 t = (u64)a.v[i] - b.v[i] - bw;
        r.v[i] = (u32)t;
        bw = t >> 63;
    }
    u32 mask = (u32)0 - (u32)bw;  // add p back when a < b
    u64 c = 0;
    ZK_UNROLL for (int i = 0; i < P::N; ++i) {
        c += (u64)r.v[i] + (P::mod(i) & mask);
        r.v[i] = (u32)c;
        c >>= 32;
    }
    return r;
}

template <class P>
ZK_HD Fe<P> fe_neg(const Fe<P>& a) {
    return fe_sub(Fe<P>::zero(), a);
}
template <class P>
ZK_HD Fe<P> fe_dbl(const Fe<P>& a) {
    return fe_add(a, a);
}

// Montgomery product a*b*R^{-1} mod p.  CIOS with the "spare top bit" simplification
// (valid because 2p < 2^(32N) for all four moduli): per outer iteration 2N v_mad_u64_u32.
#if defined(__HIP_DEVICE_COMPILE__)
// Device form: product scanning with a 96-bit column accumulator; the carry out of every v_mad_u64_u32 goes straight
// into the third word (v_addc_co_u32) instead of through the compiler's 64-bit add sequences.  Same result as the
// portable CIOS below (tools/femul_bench.hip: bit-identical, 125 vs 87 G mul/s on MI355X).
__device__ __forceinline__ void fe_mac3(u64& acc, u32& ext, u32 a, u32 b) {
    u64 carry;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %0\n\tv_addc_co_u32_e64 %2, %1, 0, %2, %1" : "+v"(acc), "=&s"(carry), "+v"(ext) : "v"(a), "v"(b));
}
template <class P>
__device__ __forceinline__ Fe<P> fe_mul_device(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    u32 m[N];
    Fe<P> r;
    u64 acc = 0;
    u32 ext = 0;
    ZK_UNROLL for (int k = 0; k < N; ++k) {
        ZK_UNROLL for (int i = 0; i < k; ++i) {
            fe_mac3(acc, ext, a.v[i], b.v[k - i]);
            fe_mac3(acc, ext, m[i], P::mod(k - i));
        }
        fe_mac3(acc, ext, a.v[k], b.v[0]);
        m[k] = (u32)acc * P::INV;
        fe_mac3(acc, ext, m[k], P::mod(0));
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
    ZK_UNROLL for (int k = N; k < 2 * N - 1; ++k) {
        ZK_UNROLL for (int i = k - N + 1; i < N; ++i) {
            fe_mac3(acc, ext, a.v[i], b.v[k - i]);
            fe_mac3(acc, ext, m[i], P::mod(k - i));
        }
        r.v[k - N] = (u32)acc;
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
    r.v[N - 1] = (u32)acc;
    fe_reduce_once(r);
    return r;
}
#endif
template <class P>
ZK_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_device(a, b);
#else
    constexpr int N = P::N;
    u32 t[N];
    ZK_UNROLL for (int i = 0; i < N; ++i) t[i] = 0;
    ZK_UNROLL for (int i = 0; i < N; ++i) {
        const u32 bi = b.v[i];
        u64 x = (u64)a.v[0] * bi + t[0];
        u32 A = (u32)(x >> 32);
        const u32 m = (u32)x * P::INV;
        u64 y = (u64)m * P::mod(0) + (u32)x;
        u32 C = (u32)(y >> 32);
        ZK_UNROLL for (int j = 1; j < N; ++j) {
            x = (u64)a.v[j] * bi + t[j] + A;
            A = (u32)(x >> 32);
            y = (u64)m * P::mod(j) + (u32)x + C;
            C = (u32)(y >> 32);
            t[j - 1] = (u32)y;
        }
        t[N - 1] = C + A;
    }
    Fe<P> r;
    ZK_UNROLL for (int i = 0; i < N; ++i) r.v[i] = t[i];
    fe_reduce_once(r);
    return r;
#endif
}
template <class P>
ZK_HD Fe<P> fe_sqr(const Fe<P>& a) {
    return fe_mul(a, a);
}
// The same product as a real call on the device, operands and result in VGPRs (by value).  The elliptic-curve
// kernels use this form: a mixed addition inlines to 60-170 KB of code, far beyond the 64 KB instruction cache
// (measured: G2 accumulation 21 ms inlined vs 8 ms with calls, tools/accum_bench.hip), and it compiles ~50x faster.
template <class P>
ZK_HD_CALL Fe<P> fe_mul_nc(const Fe<P> a, const Fe<P> b) {
    return fe_mul(a, b);
}

// canonical integer (limbs) <-> Montgomery
template <class P>
ZK_HD Fe<P> fe_to_mont(const Fe<P>& canon) {
    return fe_mul(canon, Fe<P>::r2());
}
template <class P>
ZK_HD Fe<P> fe_from_mont(const Fe<P>& a) {
    Fe<P> o = Fe<P>::zero();
    o.v[0] = 1;
    return fe_mul(a, o);
}

// a^e for a little-endian limb exponent (square-and-multiply, MSB first)
template <class P>
ZK_HD_CALL Fe<P> fe_pow(const Fe<P>& a, const u32* e, int nlimbs) {
    Fe<P> r = Fe<P>::one();
    for (int i = 32 * nlimbs - 1; i >= 0; --i) {
        r = fe_sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul(r, a);
    }
    return r;
}
template <class P>
ZK_HD Fe<P> fe_pow_u64(const Fe<P>& a, u64 e) {
    u32 l[2] = {(u32)e, (u32)(e >> 32)};
    return fe_pow(a, l, 2);
}
// Fermat inverse (0 -> 0)
template <class P>
ZK_HD_CALL Fe<P> fe_inv(const Fe<P>& a) {
    u32 e[P::N];
    u64 bw = 2;
    for (int i = 0; i < P::N; ++i) {
        u64 t = (u64)P::mod(i) - bw;
        e[i] = (u32)t;
        bw = t >> 63;
    }
    return fe_pow(a, e, P::N);
}
template <class P>
ZK_HD Fe<P> fe_from_u64(u64 x) {
    Fe<P> c = Fe<P>::zero();
    c.v[0] = (u32)x;
    c.v[1] = (u32)(x >> 32);
    return fe_to_mont(c);
}

// multiplication as the curve code wants it: an out-of-line call for either field
template <class P> ZK_HD Fe<P> ec_mul(const Fe<P>& a, const Fe<P>& b) { return fe_mul_nc(a, b); }
template <class P> ZK_HD Fe<P> ec_sqr(const Fe<P>& a) { return fe_mul_nc(a, a); }

// ---- quadratic extension Fq2 = Fq[u]/(u^2+1) (both supported curves) ----
template <class P>
struct Fe2 {
    typedef P Params;
    static constexpr int BYTES = 8 * P::N;
    Fe<P> c0, c1;
    ZK_HD static Fe2 zero() { return {Fe<P>::zero(), Fe<P>::zero()}; }
    ZK_HD static Fe2 one() { return {Fe<P>::one(), Fe<P>::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool equals(const Fe2& o) const { return c0.equals(o.c0) && c1.equals(o.c1); }
};
template <class P> ZK_HD Fe2<P> fe_add(const Fe2<P>& a, const Fe2<P>& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
template <class P> ZK_HD Fe2<P> fe_sub(const Fe2<P>& a, const Fe2<P>& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
template <class P> ZK_HD Fe2<P> fe_neg(const Fe2<P>& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
template <class P> ZK_HD Fe2<P> fe_dbl(const Fe2<P>& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
// Fq2 products are out-of-line calls as well (operands by value), built from fe_mul_nc
template <class P>
ZK_HD_CALL Fe2<P> fe_mul(const Fe2<P> a, const Fe2<P> b) {  // Karatsuba: 3 base-field products
    Fe<P> v0 = fe_mul_nc(a.c0, b.c0), v1 = fe_mul_nc(a.c1, b.c1);
    Fe<P> s = fe_mul_nc(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return {fe_sub(v0, v1), fe_sub(fe_sub(s, v0), v1)};
}
template <class P>
ZK_HD_CALL Fe2<P> fe_sqr(const Fe2<P> a) {  // complex squaring: 2 base-field products
    Fe<P> t = fe_mul_nc(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1));
    Fe<P> u = fe_mul_nc(a.c0, a.c1);
    return {t, fe_dbl(u)};
}
template <class P>
ZK_HD_CALL Fe2<P> fe_inv(const Fe2<P>& a) {
    Fe<P> n = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return {fe_mul(a.c0, n), fe_neg(fe_mul(a.c1, n))};
}
template <class P> ZK_HD Fe2<P> ec_mul(const Fe2<P>& a, const Fe2<P>& b) { return fe_mul(a, b); }
template <class P> ZK_HD Fe2<P> ec_sqr(const Fe2<P>& a) { return fe_sqr(a); }
template <class P> ZK_HD Fe2<P> fe_to_mont(const Fe2<P>& a) { return {fe_to_mont(a.c0), fe_to_mont(a.c1)}; }
template <class P> ZK_HD Fe2<P> fe_from_mont(const Fe2<P>& a) { return {fe_from_mont(a.c0), fe_from_mont(a.c1)}; }

}  // namespace zk
