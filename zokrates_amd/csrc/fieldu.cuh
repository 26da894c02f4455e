// fieldu.cuh — base-field arithmetic in UNSATURATED limbs for the MSM kernels (product code).
//
// gfx950 multiplies with v_mad_u64_u32 (32x32 + 64 -> 64).  With saturated 32-bit limbs every partial product needs
// carry handling that costs more issue slots than the multiply itself (the CIOS code in field.cuh runs at 87 G mul/s).
// With B-bit limbs, B < 32, the 64-bit column accumulators of a product-scanning (Comba) multiplication have enough
// headroom to take every partial product of a column with NO carry at all:
//     BN254 Fq:      9 limbs x 29 bits (R' = 2^261)  -> 166 G mul/s measured (tools/femul_bench.hip), 1.9x
//     BLS12-381 Fq: 14 limbs x 28 bits (R' = 2^392)
// Elements stay in Montgomery form w.r.t. R' and in a REDUNDANT representation: limbs may slightly exceed 2^B and
// the integer value may be a small multiple of p above the canonical one.  Invariants (checked by the bounds notes in
// ec.cuh and by the parity tests):
//     TIGHT: every limb <= 2^B + 8 (the top limb carries the value's overflow); all stored elements are TIGHT;
//     mul / mul2 accept TIGHT operands with values < 8p and return TIGHT results with value < 2p;
//     add, dbl, sub<K> finish with one parallel carry round (no ripple): TIGHT out; sub<K>(a, b) = a + K*p - b needs b < K*p.
// Only the curve kernels use this type; boundaries convert from/to the saturated Montgomery form of field.cuh
// (fu_from_fe / fu_to_fe), so nothing outside the MSM sees it.  Results are exact group elements either way.
#pragma once
#include "field.cuh"
#if defined(ZK_CHECK_OVERFLOW) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#endif

namespace zk {

template <class P> struct UCfg;
// FQ2_INLINE: whether the Fq2 product is inlined into the curve formulas (see the measurements next to ZK_FU_MUL_INLINE;
// with 14 limbs an inlined G2 mixed addition is > 100 KB of code, so BLS12-381 keeps it as a call)
#ifndef ZK_MUL_NQ
#define ZK_MUL_NQ 1
#endif
#ifndef ZK_MUL_CHAIN
#define ZK_MUL_CHAIN 0
#endif
template <> struct UCfg<Bn254Fq> { static constexpr int B = 29, N = 9, MUL_NQ = ZK_MUL_NQ; static constexpr bool FQ2_INLINE = true, MUL_CHAIN = ZK_MUL_CHAIN != 0; };
template <> struct UCfg<Bls381Fq> { static constexpr int B = 28, N = 14, MUL_NQ = 1; static constexpr bool FQ2_INLINE = false, MUL_CHAIN = false; };
// the scalar fields (NTT passes, kernels_ntt.cuh): both moduli are <= 255 bits
template <> struct UCfg<Bn254Fr> { static constexpr int B = 29, N = 9, MUL_NQ = ZK_MUL_NQ; static constexpr bool FQ2_INLINE = false, MUL_CHAIN = ZK_MUL_CHAIN != 0; };
template <> struct UCfg<Bls381Fr> { static constexpr int B = 29, N = 9, MUL_NQ = ZK_MUL_NQ; static constexpr bool FQ2_INLINE = false, MUL_CHAIN = ZK_MUL_CHAIN != 0; };

// ---- compile-time constants: p, -p^-1, powers of two mod p and bias multiples of p, all in B-bit limbs ----
template <class P>
struct UConst {
    static constexpr int B = UCfg<P>::B, N = UCfg<P>::N, W = P::N;   // W 32-bit words in the saturated form
    static constexpr u32 M = (1u << B) - 1;
    struct Words { u32 w[W + 1]; };
    struct Limbs { u32 v[N]; };

    static constexpr Words modulus() {
        Words r{};
        for (int i = 0; i < W; ++i) r.w[i] = P::mod(i);
        r.w[W] = 0;
        return r;
    }
    static constexpr bool geq(const Words& a, const Words& b) {
        for (int i = W; i >= 0; --i)
            if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
        return true;
    }
    static constexpr Words sub(const Words& a, const Words& b) {
        Words r{};
        u64 bw = 0;
        for (int i = 0; i <= W; ++i) {
            u64 t = (u64)a.w[i] - b.w[i] - bw;
            r.w[i] = (u32)t;
            bw = (t >> 63) & 1;
        }
        return r;
    }
    static constexpr Words dbl_mod(const Words& a) {   // 2a mod p for a < p
        Words r{};
        u32 c = 0;
        for (int i = 0; i <= W; ++i) {
            r.w[i] = (a.w[i] << 1) | c;
            c = a.w[i] >> 31;
        }
        const Words p = modulus();
        return geq(r, p) ? sub(r, p) : r;
    }
    static constexpr Words pow2_mod(int k) {   // 2^k mod p
        Words r{};
        r.w[0] = 1;
        for (int i = 0; i < k; ++i) r = dbl_mod(r);
        return r;
    }
    static constexpr Words times_small(const Words& a, u32 k) {   // k * a as an integer (no reduction), k small
        Words r{};
        u64 c = 0;
        for (int i = 0; i <= W; ++i) {
            c += (u64)a.w[i] * k;
            r.w[i] = (u32)c;
            c >>= 32;
        }
        return r;
    }
    static constexpr Limbs split(const Words& a) {   // integer < 2^(B*N) -> B-bit limbs
        Limbs l{};
        for (int i = 0; i < N; ++i) {
            const int bit = B * i, wi = bit >> 5, sh = bit & 31;
            u64 two = wi <= W ? a.w[wi] : 0;
            if (wi + 1 <= W) two |= (u64)a.w[wi + 1] << 32;
            l.v[i] = (u32)(two >> sh) & M;
        }
        // the top limb keeps every remaining bit
        {
            const int bit = B * (N - 1), wi = bit >> 5, sh = bit & 31;
            u64 two = wi <= W ? a.w[wi] : 0;
            if (wi + 1 <= W) two |= (u64)a.w[wi + 1] << 32;
            l.v[N - 1] = (u32)(two >> sh);
        }
        return l;
    }
    // K*p in "spread" limbs: same integer, every limb but the top one raised by 2^(B+2) (minus what the limb above
    // gives back), so that a_i + bias_i - b_i never goes negative for b_i < 2^(B+2) - 4.
    static constexpr Limbs bias(u32 k) {
        Limbs c = split(times_small(modulus(), k));
        Limbs r{};
        for (int i = 0; i < N; ++i) {
            if (i == 0) r.v[i] = c.v[i] + (1u << (B + 2));
            else if (i < N - 1) r.v[i] = c.v[i] + (1u << (B + 2)) - 4;
            else r.v[i] = c.v[i] - 4;
        }
        return r;
    }
    // K*p with every limb but the top one raised by 2^sh (the top limb gives the excess back): bias(k) is sh = B + 2; the
    // un-normalised negation of a TIGHT value (fe_neg_lazy) gets by with sh = B + 1
    static constexpr Limbs bias_spread(u32 k, int sh) {
        Limbs c = split(times_small(modulus(), k));
        Limbs r{};
        const u32 up = 1u << sh, back = 1u << (sh - B);
        for (int i = 0; i < N; ++i) {
            if (i == 0) r.v[i] = c.v[i] + up;
            else if (i < N - 1) r.v[i] = c.v[i] + up - back;
            else r.v[i] = c.v[i] - back;
        }
        return r;
    }
    static constexpr u32 inv_low() {   // p^-1 mod 2^32 (Newton)
        u32 p0 = P::mod(0), x = 1;
        for (int i = 0; i < 6; ++i) x *= 2 - p0 * x;
        return x;
    }

    ZK_HD static constexpr u32 p(int i) { constexpr Limbs t = split(modulus()); return t.v[i]; }
    ZK_HD static constexpr u32 one(int i) { constexpr Limbs t = split(pow2_mod(B * N)); return t.v[i]; }            // R' mod p
    ZK_HD static constexpr u32 from_fe(int i) { constexpr Limbs t = split(pow2_mod(2 * B * N - 32 * W)); return t.v[i]; }   // x*2^(32W) -> x*R'
    ZK_HD static constexpr u32 bias2(int i) { constexpr Limbs t = bias(2); return t.v[i]; }
    ZK_HD static constexpr u32 bias4(int i) { constexpr Limbs t = bias(4); return t.v[i]; }
    ZK_HD static constexpr u32 bias8(int i) { constexpr Limbs t = bias(8); return t.v[i]; }
    ZK_HD static constexpr u32 bias16(int i) { constexpr Limbs t = bias(16); return t.v[i]; }
    ZK_HD static constexpr u32 nbias2(int i) { constexpr Limbs t = bias_spread(2, B + 1); return t.v[i]; }   // 2p, spread 2^(B+1)
    ZK_HD static constexpr u32 nbias4(int i) { constexpr Limbs t = bias_spread(4, B + 1); return t.v[i]; }
    ZK_HD static constexpr u32 nbias8(int i) { constexpr Limbs t = bias_spread(8, B + 1); return t.v[i]; }
    static constexpr u32 PINV = inv_low() & M;               //  p^-1 mod 2^B
    static constexpr u32 NINV = (0u - inv_low()) & M;        // -p^-1 mod 2^B
    static constexpr u32 NINV32 = 0u - inv_low();            // -p^-1 mod 2^32 (the loose quotient digits of fu_dot_inl<.., LOOSE>)
    static constexpr u64 limb_sum() { u64 t = 0; for (int i = 0; i < N; ++i) t += split(modulus()).v[i]; return t; }
    // LOOSE quotient digits are 32-bit: a column then holds up to 2^32 * (sum of p's limbs) of them next to N * 2^(2B+2) of products
    // — one product of a TIGHT and a lazily added / negated operand (limbs < 2^(B+2)), the two products of an Fq2 component (TIGHT x
    // TIGHT + TIGHT x 3 * 2^B), or the four TIGHT products of the fused Y3: both must fit the 64-bit accumulator
    static constexpr bool LOOSE_OK = (limb_sum() << 30) + ((u64)N << (2 * B)) < ((u64)1 << 62) - ((u64)1 << 57);   // (divided by four; 3 % to spare)
    static constexpr u32 P_TOP = split(modulus()).v[N - 1];
    static constexpr u32 Q_MAGIC = (u32)(((u64)1 << 32) / ((u64)P_TOP + 1));   // floor(2^32 / (p_top + 1))
};

template <class P>
struct Fu {
    typedef P Params;
    typedef UConst<P> C;
    static constexpr int N = C::N, B = C::B;
    static constexpr u32 M = C::M;
    u32 v[N];

    ZK_HD static Fu zero() {
        Fu r;
        ZK_UNROLL for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fu one() {
        Fu r;
        ZK_UNROLL for (int i = 0; i < N; ++i) r.v[i] = C::one(i);
        return r;
    }
    // all limbs zero: the representation of the affine-infinity sentinel and of an empty accumulator (NOT "== 0 mod p")
    ZK_HD bool is_zero() const {
        u32 a = 0;
        ZK_UNROLL for (int i = 0; i < N; ++i) a |= v[i];
        return a == 0;
    }
};

// one parallel carry round: limb i keeps its low B bits and receives the overflow of limb i-1
template <class P>
ZK_HD Fu<P> fu_norm(const u32* t) {
    constexpr int N = Fu<P>::N, B = Fu<P>::B;
    Fu<P> r;
    r.v[0] = t[0] & Fu<P>::M;
    ZK_UNROLL for (int i = 1; i < N - 1; ++i) r.v[i] = (t[i] & Fu<P>::M) + (t[i - 1] >> B);
    r.v[N - 1] = t[N - 1] + (t[N - 2] >> B);
    return r;
}
template <class P>
ZK_HD Fu<P> fe_add(const Fu<P>& a, const Fu<P>& b) {
    u32 t[Fu<P>::N];
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) t[i] = a.v[i] + b.v[i];
    return fu_norm<P>(t);
}
template <class P>
ZK_HD Fu<P> fe_dbl(const Fu<P>& a) {
    u32 t[Fu<P>::N];
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) t[i] = a.v[i] << 1;
    return fu_norm<P>(t);
}
// a + K*p - b for K in {2, 4, 8, 16}; needs value(b) < K*p
template <int K, class P>
ZK_HD Fu<P> fe_sub_k(const Fu<P>& a, const Fu<P>& b) {
    typedef UConst<P> C;
    static_assert(K == 2 || K == 4 || K == 8 || K == 16, "fe_sub_k: K in {2, 4, 8, 16}");
    u32 t[Fu<P>::N];
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) t[i] = a.v[i] + (K == 2 ? C::bias2(i) : K == 4 ? C::bias4(i) : K == 8 ? C::bias8(i) : C::bias16(i)) - b.v[i];
    return fu_norm<P>(t);
}
template <class P> ZK_HD Fu<P> fe_sub(const Fu<P>& a, const Fu<P>& b) { return fe_sub_k<4>(a, b); }
// negation of an AFFINE coordinate (value < 2p, as produced by fu_from_fe): the result is again < 2p, so a negated base
// obeys the same bounds as any other; the all-zero sentinel of the point at infinity stays all-zero
template <class P> ZK_HD Fu<P> fe_neg(const Fu<P>& a) { return a.is_zero() ? a : fe_sub_k<2>(Fu<P>::zero(), a); }

// The helpers below subtract limb by limb from a SPREAD multiple of p without a carry round: no limb may go negative.  The lower
// limbs are safe by the spread (2^(B+1) above any TIGHT limb); the TOP limb is (K p)_top - 2 - b_top, which wraps if the subtrahend's
// top limb exceeds that — i.e. the VALUE precondition: subtrahend < about (K - 1) p (a Montgomery product or an affine coordinate
// against 2p; the column checks of ZK_CHECK_OVERFLOW do not see a wrapped limb, so the builds that carry them check this too).
#if defined(ZK_CHECK_OVERFLOW) && !defined(__HIP_DEVICE_COMPILE__)
#define ZK_LAZY_TOP_CHECK(P_, bias_top, b_top, who) do { if ((b_top) > (bias_top)) { fprintf(stderr, "%s: top limb of the subtrahend (%u) above the bias's (%u): a wrapped limb would be multiplied\n", who, (unsigned)(b_top), (unsigned)(bias_top)); abort(); } } while (0)
#else
#define ZK_LAZY_TOP_CHECK(P_, bias_top, b_top, who) ((void)0)
#endif
// 2p - a WITHOUT the carry round, for a TIGHT a with value < 2p (a product, an affine coordinate): limbs up to 2^(B+1) + 2^B, so
// the result is ONLY good as one operand of a single product (fu_mul_inl) or of a two-product sum (fu_mul2_inl) whose other
// operands are TIGHT — N * 2^(2B+1.6) + the reduction's N * 2^(2B) stay far below 2^64 — and saves the 25 instructions of
// the carry round where a negated value is multiplied at once (the base's y of a negative digit, PPP in the fused Y3).
template <class P>
ZK_HD Fu<P> fe_neg_lazy(const Fu<P>& a) {
    Fu<P> r;
    ZK_LAZY_TOP_CHECK(P, UConst<P>::nbias2(Fu<P>::N - 1), a.v[Fu<P>::N - 1], "fe_neg_lazy");
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) r.v[i] = UConst<P>::nbias2(i) - a.v[i];
    return r;
}
// a + K p - b and a + b WITHOUT the carry round, for TIGHT a and b (value(b) < K p): limbs up to 2^(B+2), so the result is ONLY
// good as the operand of a single product whose other operand is TIGHT with limbs < 2^B exactly — a twiddle factor unpacked
// from its table: N * 2^(2B+2) + the reduction's N * 2^(2B) < 2^64 for both limb widths.  The butterflies of the transforms
// multiply most of their sums and differences at once (kernels_ntt.cuh): 25 instructions less each.
template <int K, class P>
ZK_HD Fu<P> fe_sub_k_lazy(const Fu<P>& a, const Fu<P>& b) {
    typedef UConst<P> C;
    Fu<P> r;
    ZK_LAZY_TOP_CHECK(P, (K == 2 ? C::nbias2(Fu<P>::N - 1) : K == 4 ? C::nbias4(Fu<P>::N - 1) : C::nbias8(Fu<P>::N - 1)), b.v[Fu<P>::N - 1], "fe_sub_k_lazy");
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) r.v[i] = a.v[i] + ((K == 2 ? C::nbias2(i) : K == 4 ? C::nbias4(i) : C::nbias8(i)) - b.v[i]);
    return r;
}
template <class P>
ZK_HD Fu<P> fe_add_lazy(const Fu<P>& a, const Fu<P>& b) {
    Fu<P> r;
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// neg ? 2p - y : y for the y of a base that is NOT the point at infinity; the result only feeds the product S2 = ZZZ1 * y
template <class P>
ZK_HD Fu<P> fe_cneg_for_mul(const Fu<P>& y, bool neg) {
    Fu<P> r;
    if (neg) ZK_LAZY_TOP_CHECK(P, UConst<P>::nbias2(Fu<P>::N - 1), y.v[Fu<P>::N - 1], "fe_cneg_for_mul");
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) r.v[i] = neg ? UConst<P>::nbias2(i) - y.v[i] : y.v[i];
    return r;
}
// neg ? 2p - y : y, normalised (TIGHT, < 2p): the y of a base that is not the point at infinity with the sign of its digit
template <class P>
ZK_HD Fu<P> fe_cneg(const Fu<P>& y, bool neg) {
    u32 t[Fu<P>::N];
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) t[i] = neg ? UConst<P>::bias2(i) - y.v[i] : y.v[i];
    return fu_norm<P>(t);
}
// c ? a : b, limb by limb (a v_cndmask each): where two lanes of a wavefront need different values in the same registers
template <class P>
ZK_HD Fu<P> fe_select(bool c, const Fu<P>& a, const Fu<P>& b) {
    Fu<P> r;
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
// rr + 8p - ppp - 2q in ONE carry round (the numerator of X3 = R^2 - PPP - 2Q): operands TIGHT with values < 2p, result
// TIGHT with value < 10p.  Per limb ppp_i + 2 q_i < 3 * 2^B + 24 < 2^(B+2) - 4, the bound bias8's spread is made for, and
// rr_i + bias8_i < 2^32.  Three field operations with a carry round each (sub<2>, dbl, sub<4>: 120 instructions) become 52.
template <class P>
ZK_HD Fu<P> fu_x3_numerator(const Fu<P>& rr, const Fu<P>& ppp, const Fu<P>& q) {
    u32 t[Fu<P>::N];
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) t[i] = rr.v[i] + (UConst<P>::bias8(i) - ((q.v[i] << 1) + ppp.v[i]));
    return fu_norm<P>(t);
}

#ifndef ZK_LOOSE_M
#define ZK_LOOSE_M 1
#endif
// ---- the products ----
// Product scanning (Comba): column k collects a_i * b_(k-i) and m_i * p_(k-i) in 64-bit accumulators that never overflow
// for TIGHT operands, m_k makes the column's low B bits vanish, the rest carries into column k+1.
// How the multiply-adds of one column are issued can matter: written as ONE accumulator the column is a chain of up to
// 18 dependent v_mad_u64_u32 (hipcc starts the next column in a fresh register and pays a 64-bit add per column to join
// them).  UCfg::MUL_NQ > 1 feeds each column's partial products to several accumulators round-robin and MUL_CHAIN keeps
// the compiler from re-splitting the carry chain.  In isolation at 3 wavefronts per SIMD that is worth +24 %
// (tools/mul29_bench.hip: 137.7 -> 171.4 G products/s with 2 accumulators + chain); inside the curve and NTT kernels,
// whose formulas already offer several independent products, it is worth nothing (same-box A/B, profiles/r2_ab_runs.txt:
// 87.3 vs 88.8 proofs/s) and the 14-limb BLS12-381 G1 unit then takes 40 minutes to compile — so the default is the
// single accumulator (MUL_NQ = 1, MUL_CHAIN = 0), and the knobs stay for the next compiler.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_CARRY_CHAIN(x) asm volatile("" : "+v"(x))
#else
#define ZK_CARRY_CHAIN(x)
#endif
// (Round 4 measured the alternative of writing the multiply-add as the instruction — inline v_mad_u64_u32, the carry of a column as
// the addend of the next column's first multiply-add, so that the 17 joins per product the compiler makes with v_lshl_add_u64
// disappear: 2 382 instead of 2 540 instructions per mixed addition, and SLOWER, 5.79 against 5.48 ms: one dependent chain per
// wave issues a multiply-add every ~7 cycles, three waves per SIMD do not cover that (tools/mad_latency.hip: 1.97 ns per
// instruction against 1.85 with independent chains), and the compiler puts an s_nop behind every inline-asm statement whose
// result the next instruction reads.  The compiler's fresh chain per column IS the instruction-level parallelism this kernel
// runs on.  profiles/r4b_accum_variants_ab.txt.)
// one Montgomery reduction over `NT` operand pairs: r = (sum_t x[t] * y[t]) / R'.  x[t], y[t]: pointers to N limbs.
// SQR (NT = 1, x = y): the cross terms are taken once against the doubled limb.
// LOOSE: the quotient digits m_k of all columns but the last are taken as the full 32-bit product lo32(acc) * (-p^-1
// mod 2^32) instead of its low B bits — acc + m_k p_0 then vanishes mod 2^32, a fortiori mod 2^B, which is all the reduction
// needs — and lose their mask (8 of the ~220 instructions of a product).  The LAST digit keeps it: the result is
// (T + M p) / R' with M < (m_(N-1) + 8) 2^(B(N-1)) < R' (1 + 2^-25), i.e. the same "< T / R' + p" as ever.  What changes is the
// size of the columns (UConst::LOOSE_OK: operands as the accumulation kernel's hot path hands them over — see there), so only that path uses it.
template <class P, int NT, bool SQR, bool LOOSE = false>
ZK_HD Fu<P> fu_dot_inl(const u32* const (&x)[NT], const u32* const (&y)[NT]) {
    static_assert(!LOOSE || (NT <= 4 && UConst<P>::LOOSE_OK), "loose quotient digits: a field whose columns have the room");
    typedef UConst<P> C;
    constexpr int N = Fu<P>::N, B = Fu<P>::B, NQ = UCfg<P>::MUL_NQ;
    constexpr u32 M = Fu<P>::M;
    u32 m[N], x2[N];
    if (SQR) { ZK_UNROLL for (int i = 0; i < N; ++i) x2[i] = x[0][i] << 1; }
    Fu<P> r;
    u64 acc = 0;
#if defined(ZK_CHECK_OVERFLOW) && !defined(__HIP_DEVICE_COMPILE__)
    // host builds of the tests: the same column sums in 128 bits — a column that does not fit 64 bits is a broken bounds argument
    {
        unsigned __int128 wide = 0;
        u32 mm[N];
        for (int k = 0; k < 2 * N - 1; ++k) {
            for (int i = 0; i < N; ++i) {
                const int j = k - i;
                if (j < 0 || j >= N) continue;
                if (SQR) wide += (unsigned __int128)x[0][i] * x[0][j];
                else for (int t = 0; t < NT; ++t) wide += (unsigned __int128)x[t][i] * y[t][j];
                if (i < k && j >= 1) wide += (unsigned __int128)mm[i] * C::p(j);
            }
            if (k < N) { mm[k] = (LOOSE && k < N - 1) ? (u32)wide * C::NINV32 : (((u32)wide * C::NINV) & M); wide += (unsigned __int128)mm[k] * C::p(0); }
            if (wide >> 64) { fprintf(stderr, "fu_dot_inl: column %d overflows 64 bits (NT = %d)\n", k, NT); abort(); }
            wide >>= B;
        }
    }
#endif
    ZK_UNROLL for (int k = 0; k < 2 * N - 1; ++k) {
        u64 q[NQ];
        ZK_UNROLL for (int t = 0; t < NQ; ++t) q[t] = 0;
        int slot = 0;
        ZK_UNROLL for (int i = 0; i < N; ++i) {
            const int j = k - i;
            if (j < 0 || j >= N) continue;
            const bool last = k < N && i == k;          // the x_k * y_0 terms close the column (m_k follows them)
            if (SQR) {
                if (2 * i < k) { q[slot % NQ] += (u64)x2[i] * x[0][j]; ++slot; }
            } else if (!last) {
                ZK_UNROLL for (int t = 0; t < NT; ++t) { q[slot % NQ] += (u64)x[t][i] * y[t][j]; ++slot; }
            }
            if (i < k && j >= 1) { q[slot % NQ] += (u64)m[i] * C::p(j); ++slot; }
        }
        ZK_UNROLL for (int t = 0; t < NQ; ++t) acc += q[t];
        if (SQR) {
            if ((k & 1) == 0) acc += (u64)x[0][k / 2] * x[0][k / 2];
        } else if (k < N) {
            ZK_UNROLL for (int t = 0; t < NT; ++t) acc += (u64)x[t][k] * y[t][0];
        }
        if (k < N) {
            m[k] = (LOOSE && k < N - 1) ? (u32)acc * C::NINV32 : (((u32)acc * C::NINV) & M);
            acc += (u64)m[k] * C::p(0);
        } else {
            r.v[k - N] = (u32)acc & M;
        }
        acc >>= B;
        if (UCfg<P>::MUL_CHAIN) { ZK_CARRY_CHAIN(acc); }
    }
    r.v[N - 1] = (u32)acc;
    return r;
}
// Montgomery product a*b/R'
template <class P>
ZK_HD Fu<P> fu_mul_inl(const Fu<P>& a, const Fu<P>& b) {
    const u32* const x[1] = {a.v};
    const u32* const y[1] = {b.v};
    return fu_dot_inl<P, 1, false>(x, y);
}
// a*a/R': N(N+1)/2 products instead of N^2
template <class P>
ZK_HD Fu<P> fu_sqr_inl(const Fu<P>& a) {
    const u32* const x[1] = {a.v};
    return fu_dot_inl<P, 1, true>(x, x);
}
// the same two with loose quotient digits (fu_dot_inl): operands TIGHT, at most one of them lazily negated / added (limbs < 2^(B+2))
template <class P>
ZK_HD Fu<P> fu_mul_loose(const Fu<P>& a, const Fu<P>& b) {
    const u32* const x[1] = {a.v};
    const u32* const y[1] = {b.v};
    return fu_dot_inl<P, 1, false, ZK_LOOSE_M && UConst<P>::LOOSE_OK>(x, y);
}
template <class P>
ZK_HD Fu<P> fu_sqr_loose(const Fu<P>& a) {
    const u32* const x[1] = {a.v};
    return fu_dot_inl<P, 1, true, ZK_LOOSE_M && UConst<P>::LOOSE_OK>(x, x);
}
// (a*b + c*d)/R' with one reduction — the building block of the Fq2 product
template <class P, bool LOOSE = false>
ZK_HD Fu<P> fu_mul2_inl(const Fu<P>& a, const Fu<P>& b, const Fu<P>& c, const Fu<P>& d) {
    const u32* const x[2] = {a.v, c.v};
    const u32* const y[2] = {b.v, d.v};
    return fu_dot_inl<P, 2, false, LOOSE && ZK_LOOSE_M && UConst<P>::LOOSE_OK>(x, y);
}
// (a*b + c*d + e*f + g*h)/R' with one reduction: four products of TIGHT operands still fit the 64-bit column accumulators
// (4 * 9 * 2^58 + 9 * 2^58 < 2^63.4).  Operand values < 8p: result < (4 * 64 p^2) / R' + p < 3p for both base fields' R' >= 2^7 p.
template <class P, bool LOOSE = false>
ZK_HD Fu<P> fu_mul4_inl(const Fu<P>& a, const Fu<P>& b, const Fu<P>& c, const Fu<P>& d, const Fu<P>& e, const Fu<P>& f, const Fu<P>& g,
                        const Fu<P>& h) {
    const u32* const x[4] = {a.v, c.v, e.v, g.v};
    const u32* const y[4] = {b.v, d.v, f.v, h.v};
    return fu_dot_inl<P, 4, false, LOOSE && ZK_LOOSE_M && UConst<P>::LOOSE_OK>(x, y);
}
// out-of-line forms (operands by value in VGPRs) — what the curve code calls; see fe_mul_nc in field.cuh
template <class P> ZK_HD_CALL Fu<P> fu_mul(const Fu<P> a, const Fu<P> b) { return fu_mul_inl(a, b); }
template <class P> ZK_HD_CALL Fu<P> fu_mul2(const Fu<P> a, const Fu<P> b, const Fu<P> c, const Fu<P> d) { return fu_mul2_inl(a, b, c, d); }
// Inline or out-of-line?  Measured on MI355X (tools/accum_bench.hip, 2^24 mixed additions, ms):
//                                   G1     G2
//   saturated CIOS, calls          2.05   8.05
//   unsaturated, calls             1.65  12.9     (a 36-byte Fu / 72-byte Fu2 argument is passed through scratch memory)
//   unsaturated, everything inline 1.22   3.17    <- default: the Comba body is 240 instructions, a whole mixed
//                                                    addition stays inside the 64 KB instruction cache
#ifndef ZK_FU_MUL_INLINE
#define ZK_FU_MUL_INLINE 1
#endif
template <class P> ZK_HD Fu<P> ec_mul(const Fu<P>& a, const Fu<P>& b) { return ZK_FU_MUL_INLINE ? fu_mul_inl(a, b) : fu_mul(a, b); }
template <class P> ZK_HD Fu<P> ec_sqr(const Fu<P>& a) { return ZK_FU_MUL_INLINE ? fu_sqr_inl(a) : fu_mul(a, a); }

// weak reduction: TIGHT x with value < 32p -> TIGHT, value < 3p.  The quotient estimate q = floor(x_top / (p_top+1))
// never overshoots, so x - q*p >= 0; the subtraction runs through a signed ripple (it is off the multiplier's path:
// one call per point operation).
template <class P>
ZK_HD Fu<P> fe_relax(const Fu<P>& x) {
    typedef UConst<P> C;
    constexpr int N = Fu<P>::N, B = Fu<P>::B;
    const u32 q = (u32)(((u64)x.v[N - 1] * C::Q_MAGIC) >> 32);
    Fu<P> r;
    long long carry = 0;
    ZK_UNROLL for (int i = 0; i < N; ++i) {
        long long t = (long long)x.v[i] - (long long)((u64)q * C::p(i)) + carry;
        if (i < N - 1) {
            r.v[i] = (u32)t & Fu<P>::M;
            carry = t >> B;
        } else {
            r.v[i] = (u32)t;
        }
    }
    return r;
}

// x == 0 (mod p) for a TIGHT x with value < 32p.  The low limb of a multiple j*p is j*p0 mod 2^B, so
// j = x0 * p0^-1 mod 2^B must be tiny: everything else (all but 2^-25 of the calls) is rejected by one multiply.
template <class P>
ZK_HD_CALL bool fu_is_zero_modp_slow(const Fu<P> x, u32 j) {
    typedef UConst<P> C;
    constexpr int N = Fu<P>::N, B = Fu<P>::B;
    u32 a[N], b[N];
    u32 c = 0;
    for (int i = 0; i < N; ++i) {          // full carry propagation of x
        u32 t = x.v[i] + c;
        if (i < N - 1) { a[i] = t & Fu<P>::M; c = t >> B; } else a[i] = t;
    }
    u64 cc = 0;
    for (int i = 0; i < N; ++i) {          // j * p, normalised
        cc += (u64)j * C::p(i);
        if (i < N - 1) { b[i] = (u32)cc & Fu<P>::M; cc >>= B; } else b[i] = (u32)cc;
    }
    u32 d = 0;
    for (int i = 0; i < N; ++i) d |= a[i] ^ b[i];
    return d == 0;
}
template <class P>
ZK_HD bool fe_is_zero_modp(const Fu<P>& x) {
    const u32 j = (x.v[0] * UConst<P>::PINV) & Fu<P>::M;   // x.v[0] < 2^B exactly: the low limb never receives a carry
    if (j > 32) return false;      // (values up to 32p: the G1 accumulation's Pp reaches 18p since its X lost the weak reduction)
    return fu_is_zero_modp_slow(x, j);
}
// the saturated field answers the same questions trivially (canonical representation)
template <int K, class P> ZK_HD Fe<P> fe_sub_k(const Fe<P>& a, const Fe<P>& b) { return fe_sub(a, b); }
template <class P> ZK_HD Fe<P> fe_relax(const Fe<P>& x) { return x; }
template <class P> ZK_HD bool fe_is_zero_modp(const Fe<P>& x) { return x.is_zero(); }
template <int K, class P> ZK_HD Fe2<P> fe_sub_k(const Fe2<P>& a, const Fe2<P>& b) { return fe_sub(a, b); }
template <class P> ZK_HD Fe2<P> fe_relax(const Fe2<P>& x) { return x; }
template <class P> ZK_HD bool fe_is_zero_modp(const Fe2<P>& x) { return x.is_zero(); }

// ---- Fq2 = Fq[u]/(u^2+1) over unsaturated limbs ----
template <class P>
struct Fu2 {
    typedef P Params;
    Fu<P> c0, c1;
    ZK_HD static Fu2 zero() { return {Fu<P>::zero(), Fu<P>::zero()}; }
    ZK_HD static Fu2 one() { return {Fu<P>::one(), Fu<P>::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
};
template <class P> ZK_HD Fu2<P> fe_add(const Fu2<P>& a, const Fu2<P>& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
template <class P> ZK_HD Fu2<P> fe_dbl(const Fu2<P>& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
template <int K, class P> ZK_HD Fu2<P> fe_sub_k(const Fu2<P>& a, const Fu2<P>& b) { return {fe_sub_k<K>(a.c0, b.c0), fe_sub_k<K>(a.c1, b.c1)}; }
template <class P> ZK_HD Fu2<P> fe_sub(const Fu2<P>& a, const Fu2<P>& b) { return fe_sub_k<4>(a, b); }
template <class P> ZK_HD Fu2<P> fe_neg(const Fu2<P>& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
template <class P> ZK_HD Fu2<P> fe_relax(const Fu2<P>& a) { return {fe_relax(a.c0), fe_relax(a.c1)}; }
template <class P> ZK_HD bool fe_is_zero_modp(const Fu2<P>& a) { return fe_is_zero_modp(a.c0) && fe_is_zero_modp(a.c1); }
// Fq2 keeps the carry rounds: its products are sums of two (fused Y3: four) limb products whose columns have no room for
// un-normalised operands
template <class P> ZK_HD Fu2<P> fe_cneg_for_mul(const Fu2<P>& y, bool neg) { return neg ? Fu2<P>{fe_sub_k<2>(Fu<P>::zero(), y.c0), fe_sub_k<2>(Fu<P>::zero(), y.c1)} : y; }
template <class P> ZK_HD Fu2<P> fu_x3_numerator(const Fu2<P>& rr, const Fu2<P>& ppp, const Fu2<P>& q) {
    return {fu_x3_numerator(rr.c0, ppp.c0, q.c0), fu_x3_numerator(rr.c1, ppp.c1, q.c1)};
}
template <class P> ZK_HD Fu2<P> fe_cneg(const Fu2<P>& y, bool neg) { return {fe_cneg(y.c0, neg), fe_cneg(y.c1, neg)}; }
template <class P> ZK_HD Fu2<P> fe_select(bool c, const Fu2<P>& a, const Fu2<P>& b) { return {fe_select(c, a.c0, b.c0), fe_select(c, a.c1, b.c1)}; }
template <class P> ZK_HD Fe<P> fe_cneg(const Fe<P>& y, bool neg) { return neg ? fe_neg(y) : y; }
template <class P> ZK_HD Fe2<P> fe_cneg(const Fe2<P>& y, bool neg) { return neg ? fe_neg(y) : y; }
template <class P> ZK_HD Fe<P> fe_select(bool c, const Fe<P>& a, const Fe<P>& b) { return c ? a : b; }
template <class P> ZK_HD Fe2<P> fe_select(bool c, const Fe2<P>& a, const Fe2<P>& b) { return c ? a : b; }
template <class P> ZK_HD Fe<P> fe_cneg_for_mul(const Fe<P>& y, bool neg) { return neg ? fe_neg(y) : y; }
template <class P> ZK_HD Fe2<P> fe_cneg_for_mul(const Fe2<P>& y, bool neg) { return neg ? fe_neg(y) : y; }
template <class P> ZK_HD Fe<P> fu_x3_numerator(const Fe<P>& rr, const Fe<P>& ppp, const Fe<P>& q) { return fe_sub(fe_sub(rr, ppp), fe_dbl(q)); }
template <class P> ZK_HD Fe2<P> fu_x3_numerator(const Fe2<P>& rr, const Fe2<P>& ppp, const Fe2<P>& q) { return fe_sub(fe_sub(rr, ppp), fe_dbl(q)); }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u: two sums of two products, each reduced once
// (the same 4 x N^2 + 2 x N^2 multiply-adds as Karatsuba's three full products, but one negation instead of five
// additions, and results that stay below 2p whatever the operands)
template <class P, bool LOOSE = false>
ZK_HD Fu2<P> fu2_mul_inl(const Fu2<P>& a, const Fu2<P>& b) {
    // 8p - b1 without its carry round (limbs < 2^(B+1) + 2^B): it is multiplied at once, in a column of two products whose other
    // operands are TIGHT — N (2^(2B) + 2^(2B+1.6)) + N 2^(2B) stays below 2^64 for both limb widths (ZK_CHECK_OVERFLOW builds check)
    Fu<P> nb1;
    ZK_LAZY_TOP_CHECK(P, UConst<P>::nbias8(Fu<P>::N - 1), b.c1.v[Fu<P>::N - 1], "fu2_mul_inl");
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) nb1.v[i] = UConst<P>::nbias8(i) - b.c1.v[i];
    return {fu_mul2_inl<P, LOOSE>(a.c0, b.c0, a.c1, nb1), fu_mul2_inl<P, LOOSE>(a.c0, b.c1, a.c1, b.c0)};
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u: two single products.  Operands < 6p keep (a0 + a1) < 12p and
// (a0 + 8p - a1) < 14p, so the result stays below 12*14/169 + 1 < 2p.
template <class P, bool LOOSE = false>
ZK_HD Fu2<P> fu2_sqr_inl(const Fu2<P>& a) {
    // (the sum and the doubled limb are multiplied at once by a TIGHT operand: no carry round for them)
    Fu<P> d0;
    ZK_UNROLL for (int i = 0; i < Fu<P>::N; ++i) d0.v[i] = a.c0.v[i] << 1;
    if (LOOSE) return {fu_mul_loose(fe_add_lazy(a.c0, a.c1), fe_sub_k<8>(a.c0, a.c1)), fu_mul_loose(d0, a.c1)};
    return {fu_mul_inl(fe_add_lazy(a.c0, a.c1), fe_sub_k<8>(a.c0, a.c1)), fu_mul_inl(d0, a.c1)};
}
// The same product with THREE limb products instead of four (Karatsuba on the columns, before any reduction):
//   c0 = a0 b0 + a1 (8p - b1),   c1 = (a0 + a1)(b0 + b1) - a0 b0 + a1 (8p - b1)      [a1 (8p - b1) = -a1 b1 mod p]
// Column k of (a0 + a1)(b0 + b1) is the sum of the columns of a0 b0, a0 b1, a1 b0, a1 b1, so subtracting column k of a0 b0
// never goes negative; both reductions run side by side.  5 N^2 multiply-adds instead of 6 N^2.
// Operands: TIGHT, a < 4p per component, b < 2p per component.
template <class P>
ZK_HD Fu2<P> fu2_mul_kara(const Fu2<P>& a, const Fu2<P>& b) {
    typedef UConst<P> C;
    constexpr int N = Fu<P>::N, B = Fu<P>::B;
    constexpr u32 M = Fu<P>::M;
    const Fu<P> nb1 = fe_sub_k<8>(Fu<P>::zero(), b.c1);
    // limb-wise sums WITHOUT a carry round: only then is every column of sa * sb at least the same column of a0 * b0
    // (limbs <= 2^30 + 16: nine products of 2^60 plus the other terms stay below 2^64)
    Fu<P> sa, sb;
    ZK_UNROLL for (int i = 0; i < N; ++i) { sa.v[i] = a.c0.v[i] + a.c1.v[i]; sb.v[i] = b.c0.v[i] + b.c1.v[i]; }
    u32 m0[N], m1[N];
    Fu2<P> r;
    u64 acc0 = 0, acc1 = 0;
    ZK_UNROLL for (int k = 0; k < 2 * N - 1; ++k) {
        u64 p0 = 0, p1 = 0, ps = 0;
        ZK_UNROLL for (int i = 0; i < N; ++i) {
            if (k - i >= 0 && k - i < N) {
                p0 += (u64)a.c0.v[i] * b.c0.v[k - i];
                p1 += (u64)a.c1.v[i] * nb1.v[k - i];
                ps += (u64)sa.v[i] * sb.v[k - i];
            }
        }
        acc0 += p0 + p1;
        acc1 += ps - p0 + p1;
        ZK_UNROLL for (int i = 0; i < N; ++i) {
            if (i < k && k - i < N && i < N) {
                acc0 += (u64)m0[i] * C::p(k - i);
                acc1 += (u64)m1[i] * C::p(k - i);
            }
        }
        if (k < N) {
            m0[k] = ((u32)acc0 * C::NINV) & M;
            m1[k] = ((u32)acc1 * C::NINV) & M;
            acc0 += (u64)m0[k] * C::p(0);
            acc1 += (u64)m1[k] * C::p(0);
        } else {
            r.c0.v[k - N] = (u32)acc0 & M;
            r.c1.v[k - N] = (u32)acc1 & M;
        }
        acc0 >>= B;
        acc1 >>= B;
    }
    r.c0.v[N - 1] = (u32)acc0;
    r.c1.v[N - 1] = (u32)acc1;
    return r;
}
// a*b - c*d in Fq2 with one reduction per component (four limb products each): the tail of the mixed addition's Y3
template <class P, bool LOOSE = false>
ZK_HD Fu2<P> fu2_mulsub_inl(const Fu2<P>& a, const Fu2<P>& b, const Fu2<P>& c, const Fu2<P>& d) {
    // c0 = a0 b0 - a1 b1 - c0 d0 + c1 d1;  c1 = a0 b1 + a1 b0 - c0 d1 - c1 d0      (negations as 8p - x)
    const Fu<P> nb1 = fe_sub_k<8>(Fu<P>::zero(), b.c1), nd0 = fe_sub_k<8>(Fu<P>::zero(), d.c0), nd1 = fe_sub_k<8>(Fu<P>::zero(), d.c1);
    return {fu_mul4_inl<P, LOOSE>(a.c0, b.c0, a.c1, nb1, c.c0, nd0, c.c1, d.c1), fu_mul4_inl<P, LOOSE>(a.c0, b.c1, a.c1, b.c0, c.c0, nd1, c.c1, nd0)};
}
template <class P> ZK_HD_CALL Fu2<P> fu2_mul_call(const Fu2<P> a, const Fu2<P> b) { return fu2_mul_inl<P>(a, b); }
template <class P> ZK_HD_CALL Fu2<P> fu2_sqr_call(const Fu2<P> a) { return fu2_sqr_inl<P>(a); }
template <class P> ZK_HD Fu2<P> ec_mul(const Fu2<P>& a, const Fu2<P>& b) { return UCfg<P>::FQ2_INLINE ? fu2_mul_inl<P>(a, b) : fu2_mul_call(a, b); }
template <class P> ZK_HD Fu2<P> ec_sqr(const Fu2<P>& a) { return UCfg<P>::FQ2_INLINE ? fu2_sqr_inl<P>(a) : fu2_sqr_call(a); }

// ---- inversion (Fermat) on the unsaturated form: x^(p-2), one squaring per exponent bit and a product per set bit ----
// For the kernels that invert once per work-item (the window-multiple tables at key load): 254 + ~127 Comba products
// instead of the same count of saturated CIOS products behind an out-of-line call (field.cuh's fe_inv: half the rate and a
// scratch frame).  x: TIGHT, value < 8p, non-zero mod p; the result is TIGHT, < 2p.
template <class P>
ZK_HD Fu<P> fu_inv(const Fu<P>& x) {
    Fu<P> r = Fu<P>::one();
    u32 borrow = 2;                                  // the exponent p - 2, word by word
    u32 e[P::N];
    ZK_UNROLL for (int i = 0; i < P::N; ++i) {
        const u64 t = (u64)P::mod(i) - borrow;
        e[i] = (u32)t;
        borrow = (u32)(t >> 63);
    }
    for (int w = P::N - 1; w >= 0; --w) {
        const u32 word = e[w];
        for (int b = 31; b >= 0; --b) {
            r = fu_sqr_inl(r);
            if ((word >> b) & 1) r = fu_mul_inl(r, x);
        }
    }
    return r;
}
template <class P> ZK_HD Fu<P> ec_inv(const Fu<P>& x) { return fu_inv(x); }

// ---- conversions at the MSM boundary ----
// saturated Montgomery (x * 2^(32W) mod p, canonical) -> unsaturated Montgomery (x * R'), TIGHT, value < 2p
template <class P>
ZK_HD Fu<P> fu_from_fe(const Fe<P>& a) {
    typedef UConst<P> C;
    constexpr int N = Fu<P>::N, B = Fu<P>::B, W = P::N;
    Fu<P> s, k;
    ZK_UNROLL for (int i = 0; i < N; ++i) {
        const int bit = B * i, wi = bit >> 5, sh = bit & 31;
        u64 two = wi < W ? a.v[wi] : 0;
        if (wi + 1 < W) two |= (u64)a.v[wi + 1] << 32;
        s.v[i] = (u32)(two >> sh) & Fu<P>::M;
        k.v[i] = C::from_fe(i);
    }
    return fu_mul_inl(s, k);
}
// unsaturated Montgomery, TIGHT, value < 8p -> saturated Montgomery, canonical
template <class P>
ZK_HD Fe<P> fu_to_fe(const Fu<P>& a) {
    constexpr int N = Fu<P>::N, B = Fu<P>::B, W = P::N;
    Fu<P> unit = Fu<P>::zero();
    unit.v[0] = 1;
    const Fu<P> x = fu_mul_inl(a, unit);          // x = value / R' = the plain integer (mod p), < 2p
    u32 l[N];
    u32 c = 0;
    ZK_UNROLL for (int i = 0; i < N; ++i) {       // full carry propagation
        u32 t = x.v[i] + c;
        if (i < N - 1) { l[i] = t & Fu<P>::M; c = t >> B; } else l[i] = t;
    }
    Fe<P> r;
    ZK_UNROLL for (int w = 0; w < W; ++w) {       // pack into 32-bit words
        const int bit = 32 * w, li = bit / B, sh = bit % B;
        u64 acc = (u64)l[li] >> sh;
        int have = B - sh;
        ZK_UNROLL for (int q = 1; q < 3; ++q)
            if (li + q < N && have < 32) { acc |= (u64)l[li + q] << have; have += B; }
        r.v[w] = (u32)acc;
    }
    fe_reduce_once(r);                            // < 2p -> canonical
    return fe_to_mont(r);
}
// ---- packed form: the integer value of a TIGHT element, < 2^(32W), in W 32-bit words ----
// What the resident MSM base tables hold (one 16-byte-aligned run of words per coordinate: a BN254 G1 point is one 64-byte
// line instead of 72 bytes straddling two) and what the NTT vectors hold between passes.  Packing propagates every carry;
// unpacking is shifts and masks, the top limb keeps whatever is left.
template <class P>
ZK_HD void fu_pack(const Fu<P>& a, u32* w) {
    constexpr int N = Fu<P>::N, B = Fu<P>::B, W = P::N;
    u32 l[N];
    u32 c = 0;
    ZK_UNROLL for (int i = 0; i < N; ++i) {
        u32 t = a.v[i] + c;
        if (i < N - 1) { l[i] = t & Fu<P>::M; c = t >> B; } else l[i] = t;
    }
    ZK_UNROLL for (int k = 0; k < W; ++k) {
        const int bit = 32 * k, li = bit / B, sh = bit % B;
        u64 acc = (u64)l[li] >> sh;
        int have = B - sh;
        ZK_UNROLL for (int q = 1; q < 3; ++q)
            if (li + q < N && have < 32) { acc |= (u64)l[li + q] << have; have += B; }
        w[k] = (u32)acc;
    }
}
template <class P>
ZK_HD Fu<P> fu_unpack(const u32* w) {
    constexpr int N = Fu<P>::N, B = Fu<P>::B, W = P::N;
    Fu<P> s;
    ZK_UNROLL for (int i = 0; i < N; ++i) {
        const int bit = B * i, wi = bit >> 5, sh = bit & 31;
        u64 two = wi < W ? w[wi] : 0;
        if (wi + 1 < W) two |= (u64)w[wi + 1] << 32;
        s.v[i] = i < N - 1 ? ((u32)(two >> sh) & Fu<P>::M) : (u32)(two >> sh);
    }
    return s;
}
template <class P> ZK_HD void fu_pack(const Fu2<P>& a, u32* w) { fu_pack(a.c0, w); fu_pack(a.c1, w + P::N); }
template <class F> struct PackedWords;   // 32-bit words of one packed coordinate
template <class P> struct PackedWords<Fu<P>> { static constexpr int N = P::N; };
template <class P> struct PackedWords<Fu2<P>> { static constexpr int N = 2 * P::N; };
template <class F> struct FuUnpack;
template <class P> struct FuUnpack<Fu<P>> { ZK_HD static Fu<P> get(const u32* w) { return fu_unpack<P>(w); } };
template <class P> struct FuUnpack<Fu2<P>> { ZK_HD static Fu2<P> get(const u32* w) { return {fu_unpack<P>(w), fu_unpack<P>(w + P::N)}; } };

template <class P> ZK_HD Fu2<P> fu_from_fe(const Fe2<P>& a) { return {fu_from_fe(a.c0), fu_from_fe(a.c1)}; }
template <class P> ZK_HD Fe2<P> fu_to_fe(const Fu2<P>& a) { return {fu_to_fe(a.c0), fu_to_fe(a.c1)}; }
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)
template <class P>
ZK_HD Fu2<P> ec_inv(const Fu2<P>& a) {
    const Fu<P> n = fu_inv(fe_add(fu_sqr_inl(a.c0), fu_sqr_inl(a.c1)));
    return {fu_mul_inl(a.c0, n), fu_mul_inl(fe_sub_k<8>(Fu<P>::zero(), a.c1), n)};
}

// type map: saturated field of a group -> its unsaturated working type
template <class F> struct Unsat;
template <class P> struct Unsat<Fe<P>> { typedef Fu<P> type; };
template <class P> struct Unsat<Fe2<P>> { typedef Fu2<P> type; };

}  // namespace zk
