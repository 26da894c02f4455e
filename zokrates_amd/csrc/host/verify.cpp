// verify.cpp — `Backend::verify` of the compiled host layer: the pairing check of a Groth16 / GM17 proof, on the host CPU.
//
// The reference verifies on the CPU too: `impl Backend<T, G16> for Ark::verify` (/root/reference/zokrates_ark/src/groth16.rs:55-87)
// and `impl Backend<T, GM17> for Ark::verify` (gm17.rs:69-110) turn the hex strings of `verification.key` / `proof.json` into
// ark points (`serialization::to_g1 / to_g2`, zokrates_ark/src/lib.rs:229-271: "0x" + big-endian hex of exactly the base-field
// width, no curve check) and call [UPSTREAM] ark-groth16 0.3.0 `verify_proof`:
//     e(A, B) * e(g_ic, -gamma) * e(C, -delta) == e(alpha, beta),   g_ic = gamma_abc[0] + sum_i input_i * gamma_abc[i + 1],
// the equation of the Solidity verifier (zokrates_proof_systems/src/scheme/groth16.rs:156-172), or ark-gm17 0.3.0's two checks
//     e(A + g_alpha, B + h_beta) == e(g_alpha, h_beta) * e(g_psi, h_gamma) * e(C, h)   and   e(A, h_gamma) == e(g_gamma, B)
// (scheme/gm17.rs:170-195).  This is SURVEY.md §8 row N4: it stays on the CPU; it exists here so that the compiled host layer
// holds the whole trait (generate_proof, setup, verify) and `zkhip-cli verify` can say PASSED / FAILED as
// zokrates_cli/src/ops/verify.rs:188-195 does.  Product code: it shares nothing with oracle/pairing.py (the tests compare them).
// Known answers: the reference's own GM17 proofs (made by its ark backend over BLS12-377, tests/golden/gm17_bls12_377_*.json) verify here.
//
// How: the Miller loop runs on the TWIST in affine Fq2 coordinates and multiplies sparse line values into one Fq12
// accumulator for all pairs (one squaring per loop bit whatever the number of pairs).  Fq12 = Fq2[w] / (w^6 - xi) as six
// Fq2 coefficients.  The loop count is T = t - 1 (BN254: p - r = 6x^2; BLS12-381: |x|), the plain ate pairing: no Frobenius
// end-steps to get wrong, and for a "product of pairings == 1" test any power of a non-degenerate bilinear pairing serves.
// Everything else — the Frobenius constant xi^((p-1)/6), the exponent (p^4 - p^2 + 1) / r of the final exponentiation, the
// twist coefficient — is derived at start-up from p, r and xi with a few lines of big-integer code: no magic tables.
// Strictness: coordinates must be canonical (< p) and inputs < r, else zokrates_hip::Error (the reference panics through
// `unwrap`); a point off its curve or outside the r-torsion makes the answer `false` (ark 0.3.0 does not look and computes
// a meaningless product — false in practice as well).
#include <cstring>
#include <map>
#include <memory>

#include "../ec.cuh"
#include "../../../include/zkhip_backend.hpp"

namespace zk {
// BLS12-377 is not a curve libzkhip proves over; the verifier knows it so that it can be held against the reference's own GM17
// artefacts, all of which are over this curve (zokrates_stdlib/tests/tests/snark/gm17.json and zokrates_core_test/tests/tests/
// snark/snark_verify_bls12_377_{1,2,5}.json: proofs made by `zokrates generate-proof -b ark -s gm17`).  [UPSTREAM] ark-bls12-377
// 0.3.0: q (377 bits), r (253 bits), y^2 = x^3 + 1, Fq2 = Fq[u]/(u^2 + 5), Fq6 = Fq2[v]/(v^3 - u), twist y^2 = x^3 + 1/u (type D),
// x = 0x8508c00000000001.  Montgomery constants as field.cuh defines them (R = 2^384).
struct Bls377Fr {
    static constexpr int N = 8;
    static constexpr int BITS = 253;
    ZK_TABLE(mod, 8, 0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu)
};
struct Bls377Fq {
    static constexpr int N = 12;
    static constexpr int BITS = 377;
    static constexpr u32 INV = 0xffffffffu;
    ZK_TABLE(mod, 12, 0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu, 0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu,
             0xc63b05c0u, 0x17c510eau, 0x01ae3a46u)
    ZK_TABLE(r1, 12, 0xffffff68u, 0x02cdffffu, 0x7fffffb1u, 0x51409f83u, 0x8a7d3ff2u, 0x9f7db3a9u, 0x6e7c6305u, 0x7b4e97b7u, 0x803c84e8u,
             0x4cf495bfu, 0xe2fdf49au, 0x008d6661u)
    ZK_TABLE(r2, 12, 0x9400cd22u, 0xb786686cu, 0xb00431b1u, 0x0329fcaau, 0x62d6b46du, 0x22a5f111u, 0x827dc3acu, 0xbfdf7d03u, 0x41790bf9u,
             0x837e92f0u, 0x1e914b88u, 0x006dfccbu)
};

// Fq2 = Fq[u] / (u^2 + BETA) for the verifier: BETA = 1 for BN254 and BLS12-381 (field.cuh's Fe2), 5 for BLS12-377.  With the
// overloads below the group law of ec.cuh (Xyzz<F>) works over it unchanged.
template <class P> struct QBeta { static constexpr u32 value = 1; };
template <> struct QBeta<Bls377Fq> { static constexpr u32 value = 5; };
template <class P>
struct Q2 {
    typedef P Params;
    Fe<P> c0, c1;
    static Q2 zero() { return {Fe<P>::zero(), Fe<P>::zero()}; }
    static Q2 one() { return {Fe<P>::one(), Fe<P>::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool equals(const Q2& o) const { return c0.equals(o.c0) && c1.equals(o.c1); }
};
template <class P> inline Fe<P> q2_times_beta(const Fe<P>& x) {
    if (QBeta<P>::value == 1) return x;
    Fe<P> r = Fe<P>::zero(), t = x;
    for (u32 k = QBeta<P>::value; k; k >>= 1, t = fe_dbl(t))
        if (k & 1) r = fe_add(r, t);
    return r;
}
template <class P> inline Q2<P> fe_add(const Q2<P>& a, const Q2<P>& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
template <class P> inline Q2<P> fe_sub(const Q2<P>& a, const Q2<P>& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
template <class P> inline Q2<P> fe_neg(const Q2<P>& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
template <class P> inline Q2<P> fe_dbl(const Q2<P>& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
template <class P> inline Q2<P> fe_mul(const Q2<P>& a, const Q2<P>& b) {      // Karatsuba: (a0 b0 - beta a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u
    const Fe<P> v0 = fe_mul(a.c0, b.c0), v1 = fe_mul(a.c1, b.c1), s = fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return {fe_sub(v0, q2_times_beta(v1)), fe_sub(fe_sub(s, v0), v1)};
}
template <class P> inline Q2<P> fe_sqr(const Q2<P>& a) { return fe_mul(a, a); }
template <class P> inline Q2<P> fe_inv(const Q2<P>& a) {                        // (a0 - a1 u) / (a0^2 + beta a1^2)
    const Fe<P> n = fe_inv(fe_add(fe_sqr(a.c0), q2_times_beta(fe_sqr(a.c1))));
    return {fe_mul(a.c0, n), fe_neg(fe_mul(a.c1, n))};
}
template <class P> inline Q2<P> ec_mul(const Q2<P>& a, const Q2<P>& b) { return fe_mul(a, b); }
template <class P> inline Q2<P> ec_sqr(const Q2<P>& a) { return fe_sqr(a); }
template <int K, class P> inline Q2<P> fe_sub_k(const Q2<P>& a, const Q2<P>& b) { return fe_sub(a, b); }
template <class P> inline Q2<P> fe_relax(const Q2<P>& a) { return a; }
template <class P> inline bool fe_is_zero_modp(const Q2<P>& a) { return a.is_zero(); }
}  // namespace zk

namespace zokrates_hip {
namespace {
using namespace zk;

// ---------------- a little big-integer arithmetic (start-up only) ----------------
typedef std::vector<u32> Big;   // little-endian words, no leading zeros except for the value 0 = {}
void big_trim(Big& a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
Big big_from(const u32* w, int n) { Big a(w, w + n); big_trim(a); return a; }
Big big_small(u64 v) { Big a{(u32)v, (u32)(v >> 32)}; big_trim(a); return a; }
int big_cmp(const Big& a, const Big& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
Big big_add(const Big& a, const Big& b) {
    Big r(std::max(a.size(), b.size()) + 1, 0);
    u64 c = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        c += (u64)(i < a.size() ? a[i] : 0) + (i < b.size() ? b[i] : 0);
        r[i] = (u32)c;
        c >>= 32;
    }
    big_trim(r);
    return r;
}
Big big_sub(const Big& a, const Big& b) {   // a >= b
    Big r(a.size(), 0);
    u64 bw = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const u64 t = (u64)a[i] - (i < b.size() ? b[i] : 0) - bw;
        r[i] = (u32)t;
        bw = (t >> 63) & 1;
    }
    big_trim(r);
    return r;
}
Big big_mul(const Big& a, const Big& b) {
    Big r(a.size() + b.size() + 1, 0);
    for (size_t i = 0; i < a.size(); ++i) {
        u64 c = 0;
        for (size_t j = 0; j < b.size() || c; ++j) {
            c += (u64)r[i + j] + (j < b.size() ? (u64)a[i] * b[j] : 0);
            r[i + j] = (u32)c;
            c >>= 32;
        }
    }
    big_trim(r);
    return r;
}
int big_bits(const Big& a) { return a.empty() ? 0 : 32 * (int)(a.size() - 1) + (32 - __builtin_clz(a.back())); }
bool big_bit(const Big& a, int i) { return (size_t)(i >> 5) < a.size() && ((a[i >> 5] >> (i & 31)) & 1); }
// floor(a / b) and the remainder, shift-and-subtract (a few thousand word operations: start-up only)
Big big_divmod(const Big& a, const Big& b, Big* rem) {
    Big q((a.size() ? a.size() : 1), 0), r;
    for (int i = big_bits(a) - 1; i >= 0; --i) {
        u32 carry = big_bit(a, i);                  // r = 2 r + bit i of a
        for (size_t k = 0; k < r.size(); ++k) { const u32 nc = r[k] >> 31; r[k] = (r[k] << 1) | carry; carry = nc; }
        if (carry) r.push_back(carry);
        if (big_cmp(r, b) >= 0) { r = big_sub(r, b); q[i >> 5] |= 1u << (i & 31); }
    }
    big_trim(q);
    if (rem) *rem = r;
    return q;
}

// ---------------- per-curve facts ----------------
template <class P> struct PairingCfg;      // XI0: xi = XI0 + u; LOOP: the ate loop count t - 1 (0: p - r, the BN case)
template <> struct PairingCfg<Bn254Fq> {     // y^2 = x^3 + 3; twist y^2 = x^3 + 3 / xi (type D), xi = 9 + u
    typedef Bn254Fr Fr;
    static constexpr u32 XI0 = 9, B = 3;
    static constexpr bool D_TWIST = true;
    static constexpr u64 LOOP = 0;
};
template <> struct PairingCfg<Bls381Fq> {    // y^2 = x^3 + 4; twist y^2 = x^3 + 4 xi (type M), xi = 1 + u; |x| = 0xd201000000010000
    typedef Bls381Fr Fr;
    static constexpr u32 XI0 = 1, B = 4;
    static constexpr bool D_TWIST = false;
    static constexpr u64 LOOP = 0xd201000000010000ull;
};
template <> struct PairingCfg<Bls377Fq> {    // y^2 = x^3 + 1; twist y^2 = x^3 + 1 / xi (type D), xi = u (u^2 = -5); x = 0x8508c00000000001
    typedef Bls377Fr Fr;
    static constexpr u32 XI0 = 0, B = 1;
    static constexpr bool D_TWIST = true;
    static constexpr u64 LOOP = 0x8508c00000000001ull;
};

template <class P> Big modulus_of() { u32 w[P::N]; for (int i = 0; i < P::N; ++i) w[i] = P::mod(i); return big_from(w, P::N); }

template <class P>
struct Pairing {
    typedef PairingCfg<P> Cfg;
    typedef Fe<P> Fq;
    typedef Q2<P> Fq2;
    struct F12 { Fq2 c[6]; };                 // sum c[i] w^i, w^6 = xi

    Big p, r, loop, hard;                     // loop = t - 1 (ate), hard = (p^4 - p^2 + 1) / r
    Fq2 xi, b2, gamma[6];                     // gamma[i] = xi^(i (p - 1) / 6): (c w^i)^p = conj(c) gamma[i] w^i
    Fq b1;

    static Fq small(u32 v) { return fe_from_u64<P>(v); }
    static Fq2 conj(const Fq2& a) { return {a.c0, fe_neg(a.c1)}; }
    static Fq2 mul_small(const Fq2& a, u32 k) {   // k * a by doubling
        Fq2 r = Fq2::zero(), t = a;
        for (; k; k >>= 1, t = fe_dbl(t))
            if (k & 1) r = fe_add(r, t);
        return r;
    }
    // xi * a = (xi0 a0 - beta a1) + (a0 + xi0 a1) u
    static Fq2 mul_xi(const Fq2& a) {
        const Fq2 k = mul_small(a, Cfg::XI0);
        return {fe_sub(k.c0, q2_times_beta(a.c1)), fe_add(a.c0, k.c1)};
    }
    static Fq2 pow2(const Fq2& a, const Big& e) {
        Fq2 r = Fq2::one();
        for (int i = big_bits(e) - 1; i >= 0; --i) {
            r = fe_sqr(r);
            if (big_bit(e, i)) r = fe_mul(r, a);
        }
        return r;
    }

    Pairing() {
        p = modulus_of<P>();
        r = modulus_of<typename Cfg::Fr>();
        loop = Cfg::LOOP ? big_small(Cfg::LOOP) : big_sub(p, r);
        const Big p2 = big_mul(p, p), p4 = big_mul(p2, p2);
        Big rem;
        hard = big_divmod(big_add(big_sub(p4, p2), big_small(1)), r, &rem);
        if (!rem.empty()) throw Error(ZKHIP_ERR_BAD_ARG, "pairing: r does not divide p^4 - p^2 + 1");
        xi = {small(Cfg::XI0), Fq::one()};
        const Big e = big_divmod(big_sub(p, big_small(1)), big_small(6), &rem);
        if (!rem.empty()) throw Error(ZKHIP_ERR_BAD_ARG, "pairing: p != 1 mod 6");
        gamma[0] = Fq2::one();
        gamma[1] = pow2(xi, e);
        for (int i = 2; i < 6; ++i) gamma[i] = fe_mul(gamma[i - 1], gamma[1]);
        b1 = small(Cfg::B);
        const Fq2 bb = {b1, Fq::zero()};
        b2 = Cfg::D_TWIST ? fe_mul(bb, fe_inv(xi)) : mul_xi(bb);
    }

    // ---- Fq12 ----
    static F12 one() { F12 r; for (auto& c : r.c) c = Fq2::zero(); r.c[0] = Fq2::one(); return r; }
    static bool is_one(const F12& a) {
        bool ok = a.c[0].equals(Fq2::one());
        for (int i = 1; i < 6; ++i) ok = ok && a.c[i].is_zero();
        return ok;
    }
    static F12 fold(Fq2 (&t)[11]) {               // w^(6 + k) = xi w^k
        F12 r;
        for (int k = 0; k < 6; ++k) r.c[k] = k < 5 ? fe_add(t[k], mul_xi(t[k + 6])) : t[k];
        return r;
    }
    static F12 mul(const F12& a, const F12& b) {
        Fq2 t[11];
        for (auto& x : t) x = Fq2::zero();
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) t[i + j] = fe_add(t[i + j], fe_mul(a.c[i], b.c[j]));
        return fold(t);
    }
    static F12 sqr(const F12& a) {
        Fq2 t[11];
        for (auto& x : t) x = Fq2::zero();
        for (int i = 0; i < 6; ++i) {
            t[2 * i] = fe_add(t[2 * i], fe_sqr(a.c[i]));
            for (int j = i + 1; j < 6; ++j) t[i + j] = fe_add(t[i + j], fe_dbl(fe_mul(a.c[i], a.c[j])));
        }
        return fold(t);
    }
    // a * (v0 w^i0 + v1 w^i1 + v2 w^i2): the value of a line
    static F12 mul_sparse(const F12& a, const int (&idx)[3], const Fq2 (&val)[3]) {
        Fq2 t[11];
        for (auto& x : t) x = Fq2::zero();
        for (int i = 0; i < 6; ++i)
            for (int s = 0; s < 3; ++s) t[i + idx[s]] = fe_add(t[i + idx[s]], fe_mul(a.c[i], val[s]));
        return fold(t);
    }
    static F12 conj6(const F12& a) {               // a^(p^6): w -> -w
        F12 r = a;
        for (int i = 1; i < 6; i += 2) r.c[i] = fe_neg(a.c[i]);
        return r;
    }
    F12 frobenius(const F12& a) const {            // a^p
        F12 r;
        for (int i = 0; i < 6; ++i) r.c[i] = fe_mul(conj(a.c[i]), gamma[i]);
        return r;
    }
    // 1 / a = conj6(a) / (a conj6(a)); the norm lies in Fq6 = Fq2[v] / (v^3 - xi), v = w^2 (even coefficients only)
    static F12 inv(const F12& a) {
        const F12 ac = conj6(a), n = mul(a, ac);
        const Fq2 &a0 = n.c[0], &a1 = n.c[2], &a2 = n.c[4];
        const Fq2 t0 = fe_sub(fe_sqr(a0), mul_xi(fe_mul(a1, a2)));
        const Fq2 t1 = fe_sub(mul_xi(fe_sqr(a2)), fe_mul(a0, a1));
        const Fq2 t2 = fe_sub(fe_sqr(a1), fe_mul(a0, a2));
        const Fq2 d = fe_inv(fe_add(fe_mul(a0, t0), mul_xi(fe_add(fe_mul(a2, t1), fe_mul(a1, t2)))));
        F12 ni;
        for (auto& c : ni.c) c = Fq2::zero();
        ni.c[0] = fe_mul(t0, d); ni.c[2] = fe_mul(t1, d); ni.c[4] = fe_mul(t2, d);
        return mul(ac, ni);
    }
    F12 final_exponentiation(const F12& f) const {
        const F12 f1 = mul(conj6(f), inv(f));                       // ^(p^6 - 1)
        const F12 f2 = mul(frobenius(frobenius(f1)), f1);           // ^(p^2 + 1)
        F12 acc = one();                                            // ^((p^4 - p^2 + 1) / r)
        for (int i = big_bits(hard) - 1; i >= 0; --i) {
            acc = sqr(acc);
            if (big_bit(hard, i)) acc = mul(acc, f2);
        }
        return acc;
    }

    // ---- points ----
    bool on_curve_g1(const Aff<Fq>& a) const { return fe_sqr(a.y).equals(fe_add(fe_mul(fe_sqr(a.x), a.x), b1)); }
    bool on_curve_g2(const Aff<Fq2>& a) const { return fe_sqr(a.y).equals(fe_add(fe_mul(fe_sqr(a.x), a.x), b2)); }
    template <class F> bool in_subgroup(const Aff<F>& a) const {
        return xyzz_mul_limbs(Xyzz<F>::from_affine(a), r.data(), (int)r.size()).is_inf();
    }

    struct Pair { Aff<Fq> g1; Aff<Fq2> g2; };
    // prod_i e(g1_i, g2_i) == 1 ?   Points at infinity contribute 1.
    bool product_is_one(const std::vector<Pair>& in) const {
        std::vector<Pair> pairs;
        for (const Pair& q : in)
            if (!q.g1.is_inf() && !q.g2.is_inf()) pairs.push_back(q);
        std::vector<Aff<Fq2>> T;
        std::vector<bool> t_inf(pairs.size(), false);
        for (const Pair& q : pairs) T.push_back(q.g2);
        F12 f = one();
        // one round of the loop: every T <- T + Q (add) or 2T, and the lines through the old T's multiplied into f.  The slopes'
        // denominators of all pairs are inverted together (one Fq2 inversion per round: it costs as much as 380 products)
        std::vector<Fq2> num(pairs.size()), den(pairs.size()), pre(pairs.size());
        std::vector<bool> live(pairs.size());
        auto round = [&](bool add) {
            for (size_t k = 0; k < pairs.size(); ++k) {
                live[k] = false;
                const Aff<Fq2>& q = pairs[k].g2;
                if (t_inf[k]) { if (add) { T[k] = q; t_inf[k] = false; } continue; }   // the line through infinity and Q is vertical
                const Aff<Fq2>& t = T[k];
                if (add && !t.x.equals(q.x)) {
                    num[k] = fe_sub(q.y, t.y);
                    den[k] = fe_sub(q.x, t.x);
                } else if (add && !t.y.equals(q.y)) {        // T = -Q: a vertical line (no contribution), T + Q = infinity
                    t_inf[k] = true;
                    continue;
                } else {                                     // doubling (y = 0 cannot happen: no points of order two)
                    const Fq2 x2 = fe_sqr(t.x);
                    num[k] = fe_add(fe_dbl(x2), x2);
                    den[k] = fe_dbl(t.y);
                }
                live[k] = true;
            }
            Fq2 run = Fq2::one();
            for (size_t k = 0; k < pairs.size(); ++k)
                if (live[k]) { pre[k] = run; run = fe_mul(run, den[k]); }
            Fq2 inv_run = fe_inv(run);
            for (size_t k = pairs.size(); k-- > 0;) {
                if (!live[k]) continue;
                const Fq2 lambda = fe_mul(num[k], fe_mul(inv_run, pre[k]));
                inv_run = fe_mul(inv_run, den[k]);
                const Aff<Fq2> t = T[k];
                const Fq2 x3 = fe_sub(fe_sub(fe_sqr(lambda), t.x), add ? pairs[k].g2.x : t.x);
                T[k] = {x3, fe_sub(fe_mul(lambda, fe_sub(t.x, x3)), t.y)};
                const Fq2 yp = {pairs[k].g1.y, Fq::zero()};
                const Fq2 lx = fe_neg(Fq2{fe_mul(lambda.c0, pairs[k].g1.x), fe_mul(lambda.c1, pairs[k].g1.x)});
                const Fq2 lt = fe_sub(fe_mul(lambda, t.x), t.y);
                if (Cfg::D_TWIST) {       // l = y_P - lambda x_P w + (lambda x_T - y_T) w^3
                    const int idx[3] = {0, 1, 3};
                    const Fq2 val[3] = {yp, lx, lt};
                    f = mul_sparse(f, idx, val);
                } else {                  // w^3 l = (lambda x_T - y_T) - lambda x_P w^2 + y_P w^3      (w^3 lies in a proper subfield)
                    const int idx[3] = {0, 2, 3};
                    const Fq2 val[3] = {lt, lx, yp};
                    f = mul_sparse(f, idx, val);
                }
            }
        };
        for (int i = big_bits(loop) - 2; i >= 0; --i) {
            f = sqr(f);
            round(false);
            if (big_bit(loop, i)) round(true);
        }
        return is_one(final_exponentiation(f));
    }

    // ---- decoding of the JSON strings ----
    static int hexval(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
    // "0x" + exactly 8 N hex digits, big-endian, < p  ->  Montgomery form   (decode_hex + Fq::read, lib.rs:236-248)
    static Fq coord(const std::string& s) {
        if (s.size() != 2 + 8 * (size_t)P::N || s[0] != '0' || s[1] != 'x')
            throw Error(ZKHIP_ERR_PARSE, "verify: a coordinate must be \"0x\" followed by " + std::to_string(8 * P::N) + " hex digits, got \"" + s + "\"");
        Fq v = Fq::zero();
        for (size_t i = 2; i < s.size(); ++i) {
            const int h = hexval(s[i]);
            if (h < 0) throw Error(ZKHIP_ERR_PARSE, "verify: bad hex digit in \"" + s + "\"");
            const size_t nib = s.size() - 1 - i;                   // nibble index from the least significant end
            v.v[nib / 8] |= (u32)h << (4 * (nib % 8));
        }
        Fq t = v;
        fe_reduce_once(t);
        if (!t.equals(v)) throw Error(ZKHIP_ERR_PARSE, "verify: coordinate " + s + " is not below the field modulus");
        return fe_to_mont(v);
    }
    static Aff<Fq> g1(const G1Affine& a) { return {coord(a.x), coord(a.y)}; }
    static Aff<Fq2> g2(const G2Affine& a) { return {{coord(a.x[0]), coord(a.x[1])}, {coord(a.y[0]), coord(a.y[1])}}; }
    // a public input: hex of any length after an optional "0x", < r   (T::try_from_str(s.trim_start_matches("0x"), 16), groth16.rs:77-83)
    Big scalar(const std::string& s) const {
        size_t i = 0;
        while (s.compare(i, 2, "0x") == 0) i += 2;
        if (i == s.size()) throw Error(ZKHIP_ERR_PARSE, "verify: empty input \"" + s + "\"");
        Big v;
        for (; i < s.size(); ++i) {
            const int h = hexval(s[i]);
            if (h < 0) throw Error(ZKHIP_ERR_PARSE, "verify: bad hex digit in input \"" + s + "\"");
            u32 carry = (u32)h;
            for (auto& w : v) { const u32 nc = w >> 28; w = (w << 4) | carry; carry = nc; }
            if (carry) v.push_back(carry);
        }
        big_trim(v);
        if (big_cmp(v, r) >= 0) throw Error(ZKHIP_ERR_PARSE, "verify: input " + s + " is not below the scalar field modulus");
        return v;
    }
    // base[0] + sum_i inputs[i] * base[i + 1]
    Aff<Fq> input_combination(const std::vector<G1Affine>& base, const std::vector<std::string>& inputs, bool* ok) const {
        Xyzz<Fq> acc = Xyzz<Fq>::inf();
        for (size_t i = 0; i < base.size(); ++i) {
            const Aff<Fq> b = g1(base[i]);
            if (!good(b)) { *ok = false; return Aff<Fq>::inf(); }       // on the curve AND in the r-torsion, like every other vk / proof point
            if (i == 0) { acc = Xyzz<Fq>::from_affine(b); continue; }
            const Big k = scalar(inputs[i - 1]);
            if (!k.empty()) acc = xyzz_add(acc, xyzz_mul_limbs(Xyzz<Fq>::from_affine(b), k.data(), (int)k.size()));
        }
        return xyzz_to_affine(acc);
    }
    Aff<Fq> add_g1(const Aff<Fq>& a, const Aff<Fq>& b) const { return xyzz_to_affine(xyzz_add(Xyzz<Fq>::from_affine(a), Xyzz<Fq>::from_affine(b))); }
    Aff<Fq2> add_g2(const Aff<Fq2>& a, const Aff<Fq2>& b) const { return xyzz_to_affine(xyzz_add(Xyzz<Fq2>::from_affine(a), Xyzz<Fq2>::from_affine(b))); }
    bool good(const Aff<Fq>& a) const { return on_curve_g1(a) && in_subgroup(a); }
    bool good(const Aff<Fq2>& a) const { return on_curve_g2(a) && in_subgroup(a); }
};

template <class P> const Pairing<P>& pairing() {
    static const Pairing<P> instance;
    return instance;
}

template <class P>
bool verify_curve(const VerificationKey& vk, const Proof& proof) {
    const Pairing<P>& e = pairing<P>();
    typedef typename Pairing<P>::Pair Pair;
    const auto a = e.g1(proof.proof.a), c = e.g1(proof.proof.c);
    const auto b = e.g2(proof.proof.b);
    if (proof.inputs.size() + 1 != vk.query.size())      // verify_proof: MalformedVerifyingKey -> unwrap -> panic
        throw Error(ZKHIP_ERR_BAD_ARG, "verify: " + std::to_string(proof.inputs.size()) + " public inputs for a verification key made for " +
                                           std::to_string(vk.query.empty() ? 0 : vk.query.size() - 1));
    bool ok = true;
    const auto x = e.input_combination(vk.query, proof.inputs, &ok);
    if (!ok || !e.good(a) || !e.good(b) || !e.good(c)) return false;
    if (vk.scheme == "g16") {
        const auto alpha = e.g1(vk.g1.at("alpha"));
        const auto beta = e.g2(vk.g2.at("beta")), gamma = e.g2(vk.g2.at("gamma")), delta = e.g2(vk.g2.at("delta"));
        if (!e.good(alpha) || !e.good(beta) || !e.good(gamma) || !e.good(delta)) return false;
        return e.product_is_one({Pair{a, b}, Pair{aff_neg(x), gamma}, Pair{aff_neg(c), delta}, Pair{aff_neg(alpha), beta}});
    }
    const auto g_alpha = e.g1(vk.g1.at("g_alpha")), g_gamma = e.g1(vk.g1.at("g_gamma"));
    const auto h = e.g2(vk.g2.at("h")), h_beta = e.g2(vk.g2.at("h_beta")), h_gamma = e.g2(vk.g2.at("h_gamma"));
    if (!e.good(g_alpha) || !e.good(g_gamma) || !e.good(h) || !e.good(h_beta) || !e.good(h_gamma)) return false;
    const bool first = e.product_is_one({Pair{g_alpha, h_beta}, Pair{x, h_gamma}, Pair{c, h}, Pair{aff_neg(e.add_g1(a, g_alpha)), e.add_g2(b, h_beta)}});
    return first && e.product_is_one({Pair{a, h_gamma}, Pair{aff_neg(g_gamma), b}});
}

// ---------------- a JSON reader for the two small files ----------------
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    std::string text;                                    // String: the value; Number / Bool: the token
    std::vector<Json> items;
    std::vector<std::pair<std::string, Json>> fields;    // in file order
    const Json* get(const std::string& k) const {
        for (const auto& f : fields)
            if (f.first == k) return &f.second;
        return nullptr;
    }
};
struct JsonReader {
    const std::string& s;
    size_t i = 0;
    const char* what;
    [[noreturn]] void fail(const std::string& why) const {
        throw Error(ZKHIP_ERR_PARSE, std::string("Could not deserialize ") + what + ": " + why + " at byte " + std::to_string(i));
    }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    std::string string() {
        std::string out;
        for (++i; i < s.size() && s[i] != '"'; ++i) {
            if (s[i] != '\\') { out.push_back(s[i]); continue; }
            if (++i >= s.size()) break;
            switch (s[i]) {
                case 'n': out.push_back('\n'); break;
                case 't': out.push_back('\t'); break;
                case 'r': out.push_back('\r'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'u': {
                    if (i + 4 >= s.size()) fail("short \\u escape");
                    unsigned cp = 0;
                    for (int k = 1; k <= 4; ++k) {
                        const char c = s[i + k];
                        const int h = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
                        if (h < 0) fail("bad \\u escape");
                        cp = cp * 16 + h;
                    }
                    i += 4;
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xc0 | cp >> 6)); out.push_back((char)(0x80 | (cp & 0x3f))); }
                    else { out.push_back((char)(0xe0 | cp >> 12)); out.push_back((char)(0x80 | ((cp >> 6) & 0x3f))); out.push_back((char)(0x80 | (cp & 0x3f))); }
                    break;
                }
                default: out.push_back(s[i]);
            }
        }
        if (i >= s.size()) fail("unterminated string");
        ++i;
        return out;
    }
    Json value(int depth = 0) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (i >= s.size()) fail("unexpected end");
        Json v;
        const char c = s[i];
        if (c == '"') { v.kind = Json::String; v.text = string(); return v; }
        if (c == '[') {
            v.kind = Json::Array;
            ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.items.push_back(value(depth + 1));
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return v; }
                fail("expected , or ]");
            }
        }
        if (c == '{') {
            v.kind = Json::Object;
            ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') fail("expected a field name");
                std::string k = string();
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected :");
                ++i;
                v.fields.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return v; }
                fail("expected , or }");
            }
        }
        const size_t b = i;
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.')) ++i;
        if (i == b) fail("unexpected character");
        v.text = s.substr(b, i - b);
        v.kind = v.text == "null" ? Json::Null : (v.text == "true" || v.text == "false") ? Json::Bool : Json::Number;
        return v;
    }
    Json document() {
        Json v = value();
        ws();
        if (i != s.size()) fail("trailing characters");
        return v;
    }
};

struct Shape {      // turns a Json tree into the typed values, with serde-like messages
    const char* what;
    [[noreturn]] void fail(const std::string& why) const { throw Error(ZKHIP_ERR_PARSE, std::string("Could not deserialize ") + what + ": " + why); }
    const Json& field(const Json& o, const std::string& k) const {
        if (o.kind != Json::Object) fail("expected an object");
        const Json* v = o.get(k);
        if (!v) fail("missing field `" + k + "`");
        return *v;
    }
    std::string str(const Json& v, const std::string& name) const {
        if (v.kind != Json::String) fail("`" + name + "` should be a string");
        return v.text;
    }
    G1Affine g1(const Json& v, const std::string& name) const {
        if (v.kind != Json::Array || v.items.size() != 2) fail("`" + name + "` should be a pair of strings");
        return {str(v.items[0], name), str(v.items[1], name)};
    }
    G2Affine g2(const Json& v, const std::string& name) const {
        if (v.kind != Json::Array || v.items.size() != 2) fail("`" + name + "` should be a pair of pairs of strings");
        const G1Affine x = g1(v.items[0], name), y = g1(v.items[1], name);
        return {{x.x, x.y}, {y.x, y.y}};
    }
};
}  // namespace

// `curve` and `scheme` of either file, as zokrates_cli/src/ops/verify.rs:72-93 words the failures
static std::string tag_of(const Json& doc, const char* key, const char* where) {
    const Json* v = doc.kind == Json::Object ? doc.get(key) : nullptr;
    if (!v) throw Error(ZKHIP_ERR_PARSE, std::string("Field `") + key + "` not found in " + where);
    if (v->kind != Json::String) throw Error(ZKHIP_ERR_PARSE, std::string("`") + key + "` should be a string");
    return v->text;
}

VerificationKey VerificationKey::from_json(const std::string& text) {
    JsonReader rd{text, 0, "verification key"};
    const Json doc = rd.document();
    const Shape sh{"verification key"};
    VerificationKey vk;
    vk.curve = tag_of(doc, "curve", "verification key");
    vk.scheme = tag_of(doc, "scheme", "verification key");
    const bool g16 = vk.scheme == "g16";
    if (!g16 && vk.scheme != "gm17") throw Error(ZKHIP_ERR_BAD_ARG, "verify: scheme " + vk.scheme + " is not supported (g16, gm17)");
    const std::vector<std::string> g1_names = g16 ? std::vector<std::string>{"alpha"} : std::vector<std::string>{"g_alpha", "g_gamma"};
    const std::vector<std::string> g2_names = g16 ? std::vector<std::string>{"beta", "gamma", "delta"} : std::vector<std::string>{"h", "h_beta", "h_gamma"};
    for (const std::string& k : g1_names) vk.g1[k] = sh.g1(sh.field(doc, k), k);
    for (const std::string& k : g2_names) vk.g2[k] = sh.g2(sh.field(doc, k), k);
    const char* qname = g16 ? "gamma_abc" : "query";
    const Json& q = sh.field(doc, qname);
    if (q.kind != Json::Array) sh.fail(std::string("`") + qname + "` should be a sequence");
    for (const Json& item : q.items) vk.query.push_back(sh.g1(item, qname));
    return vk;
}

Proof Proof::from_json(const std::string& text) {
    JsonReader rd{text, 0, "proof"};
    const Json doc = rd.document();
    const Shape sh{"proof"};
    Proof p;
    p.curve = tag_of(doc, "curve", "proof");
    p.scheme = tag_of(doc, "scheme", "proof");
    const Json& pts = sh.field(doc, "proof");
    p.proof.a = sh.g1(sh.field(pts, "a"), "a");
    p.proof.b = sh.g2(sh.field(pts, "b"), "b");
    p.proof.c = sh.g1(sh.field(pts, "c"), "c");
    const Json& in = sh.field(doc, "inputs");
    if (in.kind != Json::Array) sh.fail("`inputs` should be a sequence");
    for (const Json& v : in.items) p.inputs.push_back(sh.str(v, "inputs"));
    return p;
}

bool verify(const VerificationKey& vk, const Proof& proof) {
    if (proof.curve != vk.curve)
        throw Error(ZKHIP_ERR_BAD_ARG, "Expected the curve of the proof and the verification key to be equal, found " + proof.curve + " != " + vk.curve);
    if (proof.scheme != vk.scheme)
        throw Error(ZKHIP_ERR_BAD_ARG, "Expected the scheme of the proof and the verification key to be equal, found " + proof.scheme + " != " + vk.scheme);
    if (vk.scheme != "g16" && vk.scheme != "gm17") throw Error(ZKHIP_ERR_BAD_ARG, "verify: scheme " + vk.scheme + " is not supported (g16, gm17)");
    if (vk.curve == "bn128") return verify_curve<Bn254Fq>(vk, proof);
    if (vk.curve == "bls12_381") return verify_curve<Bls381Fq>(vk, proof);
    if (vk.curve == "bls12_377") return verify_curve<Bls377Fq>(vk, proof);
    throw Error(ZKHIP_ERR_BAD_ARG, "verify: curve " + vk.curve + " is not supported (bn128, bls12_381, bls12_377)");
}

bool pairing_product_is_one(const std::string& curve, const std::vector<std::pair<G1Affine, G2Affine>>& pairs) {
    auto run = [&](auto tag) {
        typedef decltype(tag) P;
        const Pairing<P>& e = pairing<P>();
        std::vector<typename Pairing<P>::Pair> v;
        for (const auto& pr : pairs) {
            typename Pairing<P>::Pair q{e.g1(pr.first), e.g2(pr.second)};
            if (!e.good(q.g1) || !e.good(q.g2)) throw Error(ZKHIP_ERR_BAD_ARG, "pairing: a point is not in its group");
            v.push_back(q);
        }
        return e.product_is_one(v);
    };
    if (curve == "bn128") return run(Bn254Fq{});
    if (curve == "bls12_381") return run(Bls381Fq{});
    if (curve == "bls12_377") return run(Bls377Fq{});
    throw Error(ZKHIP_ERR_BAD_ARG, "pairing: curve " + curve + " is not supported (bn128, bls12_381, bls12_377)");
}

// `zokrates print-proof --format json|remix` (zokrates_cli/src/ops/print_proof.rs:85-114): the points as serde_json prints a
// compact value, then the inputs; bn128 only, as there (the Solidity verifier's curve)
std::string Proof::print(const std::string& format) const {
    if (curve != "bn128")
        throw Error(ZKHIP_ERR_BAD_ARG, "Could not print proof with given parameters (curve: " + curve + ", scheme: " + scheme + "): only bn128 is supported");
    auto q = [](const std::string& v) { return "\"" + v + "\""; };
    const std::string a = "[" + q(proof.a.x) + "," + q(proof.a.y) + "]", c = "[" + q(proof.c.x) + "," + q(proof.c.y) + "]";
    const std::string b = "[[" + q(proof.b.x[0]) + "," + q(proof.b.x[1]) + "],[" + q(proof.b.y[0]) + "," + q(proof.b.y[1]) + "]]";
    std::string in = "[";
    for (size_t i = 0; i < inputs.size(); ++i) in += (i ? "," : "") + q(inputs[i]);
    in += "]";
    if (format == "json") return "{\"a\":" + a + ",\"b\":" + b + ",\"c\":" + c + "}," + in + "\n";
    if (format == "remix") return "[" + a + ", " + b + ", " + c + "]" + (inputs.empty() ? std::string() : "," + in) + "\n";
    throw Error(ZKHIP_ERR_BAD_ARG, "print-proof: format must be json or remix, got " + format);
}

}  // namespace zokrates_hip
