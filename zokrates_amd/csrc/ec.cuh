// ec.cuh — short-Weierstrass (a = 0) group arithmetic for the MSM kernels (product code).
//
// Replaces, on the device, the group law the reference reaches through
// `VariableBaseMSM::multi_scalar_mul` inside `Groth16::prove`
// (/root/reference/zokrates_ark/src/groth16.rs:44; [UPSTREAM] ark-ec 0.3.0).
// Bucket accumulators use XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is
// 8M + 2S with no inversion and no Z-doubling, the cheapest complete-enough form for affine bases.
// Bases are affine in Montgomery form; the point at infinity is the sentinel (0, 0), which is on
// neither curve family handled here (b != 0).
#pragma once
#include "fieldu.cuh"

namespace zk {

template <class F>
struct Aff {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    ZK_HD static Aff inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct Xyzz {
    F x, y, zz, zzz;
    ZK_HD bool is_inf() const { return zz.is_zero(); }
    ZK_HD static Xyzz inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_HD static Xyzz from_affine(const Aff<F>& p) {
        if (p.is_inf()) return inf();
        return {p.x, p.y, F::one(), F::one()};
    }
};

template <class F>
ZK_HD Aff<F> aff_neg(const Aff<F>& p) {
    return {p.x, fe_neg(p.y)};   // (0,0) stays (0,0)
}
template <class F>
ZK_HD Xyzz<F> xyzz_neg(const Xyzz<F>& p) {
    return {p.x, fe_neg(p.y), p.zz, p.zzz};
}

// 2*(x, y) for an affine point (mdbl-2008-s-1)
// Product / square with a per-call-site inlining policy: HOT call sites (the accumulation kernel) always inline the Fq2
// product; elsewhere the curve's default applies (UCfg::FQ2_INLINE — BLS12-381 keeps it out of line in the cold kernels,
// whose fully inlined form takes minutes to compile).
template <bool HOT, class F> ZK_HD F ecm(const F& a, const F& b) { return ec_mul(a, b); }
template <bool HOT, class F> ZK_HD F ecs(const F& a) { return ec_sqr(a); }
// (the accumulation kernel's single products over the base field: loose quotient digits, fieldu.cuh fu_mul_loose)
template <bool HOT, class P> ZK_HD Fu<P> ecm(const Fu<P>& a, const Fu<P>& b) { return HOT ? fu_mul_loose(a, b) : ec_mul(a, b); }
template <bool HOT, class P> ZK_HD Fu<P> ecs(const Fu<P>& a) { return HOT ? fu_sqr_loose(a) : ec_sqr(a); }
template <bool HOT, class P> ZK_HD Fu2<P> ecm(const Fu2<P>& a, const Fu2<P>& b) { return (HOT || UCfg<P>::FQ2_INLINE) ? fu2_mul_inl<P, HOT>(a, b) : fu2_mul_call(a, b); }
template <bool HOT, class P> ZK_HD Fu2<P> ecs(const Fu2<P>& a) { return (HOT || UCfg<P>::FQ2_INLINE) ? fu2_sqr_inl<P, HOT>(a) : fu2_sqr_call(a); }

// a*b - c*d, the tail of Y3 = R (Q - X3) - Y1 PPP: ONE reduction (per component) instead of two products and a subtraction
// on the hot path of the unsaturated fields; plain arithmetic elsewhere.  Result < 3p.
// Measured on MI355X (same box, 2^20 BN254, serial kernel time, profiles/r2f_fused_arithmetic_ab.txt): the fused Y3 makes
// the G2 accumulation 33 % faster (6.18 -> 4.12 ms: fewer live Fq2 temporaries, far fewer spills) and leaves G1 unchanged
// at twice the registers (84 -> 168), so it is on for Fq2 only; the three-product Fq2 form (fu2_mul_kara) saves a sixth
// of the multiply-adds but costs registers the G2 kernel does not have: 4.12 -> 4.4 ms with the fused Y3, so it is off.
#ifndef ZK_LAZY_Y3_G1
#define ZK_LAZY_Y3_G1 1
#endif
#ifndef ZK_LAZY_Y3_G2
#define ZK_LAZY_Y3_G2 1
#endif
template <bool HOT, class F> ZK_HD F ec_mulsub(const F& a, const F& b, const F& c, const F& d) { return fe_sub_k<2>(ecm<HOT>(a, b), ecm<HOT>(c, d)); }
template <bool HOT, class P> ZK_HD Fu<P> ec_mulsub(const Fu<P>& a, const Fu<P>& b, const Fu<P>& c, const Fu<P>& d) {
    if (!ZK_LAZY_Y3_G1) return fe_sub_k<2>(ecm<HOT>(a, b), ecm<HOT>(c, d));
    return fu_mul2_inl<P, HOT>(a, b, c, fe_neg_lazy(d));  // d = PPP < 2p, TIGHT: its negation needs no carry round here; result < 2p
}
template <bool HOT, class P> ZK_HD Fu2<P> ec_mulsub(const Fu2<P>& a, const Fu2<P>& b, const Fu2<P>& c, const Fu2<P>& d) {
    if (ZK_LAZY_Y3_G2 && (HOT || UCfg<P>::FQ2_INLINE)) return fu2_mulsub_inl<P, HOT>(a, b, c, d);
    return fe_sub_k<2>(ecm<HOT>(a, b), ecm<HOT>(c, d));
}
#ifndef ZK_FQ2_KARATSUBA
#define ZK_FQ2_KARATSUBA 0
#endif
// the hot Fq2 product: three limb products (fieldu.cuh fu2_mul_kara) when the first operand is < 4p per component
template <bool HOT, class P> ZK_HD Fu2<P> ecm_k(const Fu2<P>& a, const Fu2<P>& b) {
    if (HOT && ZK_FQ2_KARATSUBA) return fu2_mul_kara(a, b);
    return ecm<HOT>(a, b);
}
template <bool HOT, class F> ZK_HD F ecm_k(const F& a, const F& b) { return ecm<HOT>(a, b); }

// Bounds for the unsaturated field (fieldu.cuh): products come out < 2p; the comments "< kp" track the integer values so
// that every sub<K> has value(b) < K*p and every stored coordinate stays < 8p (X < 3p after fe_relax, Y < 4p, ZZ/ZZZ < 2p).
// For the saturated field sub<K> is the plain modular subtraction and fe_relax the identity.
template <bool HOT = false, class F>
ZK_HD Xyzz<F> xyzz_dbl_affine_inl(const Aff<F>& p) {
    if (p.is_inf() || fe_is_zero_modp(p.y)) return Xyzz<F>::inf();
    F U = fe_dbl(p.y);                                   // < 2p (affine coordinates are canonical-sized)
    F V = ecs<HOT>(U);
    F W = ecm<HOT>(U, V);
    F S = ecm<HOT>(p.x, V);
    F X2 = ecs<HOT>(p.x);
    F M = fe_add(fe_dbl(X2), X2);                        // < 6p
    F X3 = fe_relax(fe_sub_k<4>(ecs<HOT>(M), fe_dbl(S)));  // 2S < 4p
    F Y3 = fe_sub_k<2>(ecm<HOT>(M, fe_sub_k<4>(S, X3)), ecm<HOT>(W, p.y));   // < 4p
    return {X3, Y3, V, W};
}
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_dbl_affine(const Aff<F>& p) {
    return xyzz_dbl_affine_inl(p);
}

// dbl-2008-s-1
template <class F>
ZK_HD Xyzz<F> xyzz_dbl_inl(const Xyzz<F>& p) {
    if (p.is_inf() || fe_is_zero_modp(p.y)) return Xyzz<F>::inf();
    F U = fe_dbl(p.y);                                   // < 8p
    F V = ec_sqr(U);
    F W = ec_mul(U, V);
    F S = ec_mul(p.x, V);
    F X2 = ec_sqr(p.x);
    F M = fe_add(fe_dbl(X2), X2);                        // < 6p
    F X3 = fe_relax(fe_sub_k<4>(ec_sqr(M), fe_dbl(S)));
    F Y3 = fe_sub_k<2>(ec_mul(M, fe_sub_k<4>(S, X3)), ec_mul(W, p.y));
    return {X3, Y3, ec_mul(V, p.zz), ec_mul(W, p.zzz)};
}
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_dbl(const Xyzz<F>& p) {
    return xyzz_dbl_inl(p);
}

// acc + affine (madd-2008-s) with the exceptional cases handled
template <class F>
ZK_HD Xyzz<F> xyzz_madd(const Xyzz<F>& a, const Aff<F>& p) {
    if (p.is_inf()) return a;
    if (a.is_inf()) return {p.x, p.y, F::one(), F::one()};
    F U2 = ec_mul(p.x, a.zz);
    F S2 = ec_mul(p.y, a.zzz);
    F Pp = fe_sub(U2, a.x);
    F R = fe_sub(S2, a.y);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_affine(p);
        return Xyzz<F>::inf();
    }
    F PP = ec_sqr(Pp);
    F PPP = ec_mul(Pp, PP);
    F Q = ec_mul(a.x, PP);
    F X3 = fe_sub(fe_sub(ec_sqr(R), PPP), fe_dbl(Q));
    F Y3 = fe_sub(ec_mul(R, fe_sub(Q, X3)), ec_mul(a.y, PPP));
    return {X3, Y3, ec_mul(a.zz, PP), ec_mul(a.zzz, PPP)};
}

// general add (add-2008-s)
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_add(const Xyzz<F>& a, const Xyzz<F>& b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F U1 = ec_mul(a.x, b.zz);
    F U2 = ec_mul(b.x, a.zz);
    F S1 = ec_mul(a.y, b.zzz);
    F S2 = ec_mul(b.y, a.zzz);
    F Pp = fe_sub(U2, U1);
    F R = fe_sub(S2, S1);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl(a);
        return Xyzz<F>::inf();
    }
    F PP = ec_sqr(Pp);
    F PPP = ec_mul(Pp, PP);
    F Q = ec_mul(U1, PP);
    F X3 = fe_sub(fe_sub(ec_sqr(R), PPP), fe_dbl(Q));
    F Y3 = fe_sub(ec_mul(R, fe_sub(Q, X3)), ec_mul(S1, PPP));
    return {X3, Y3, ec_mul(ec_mul(a.zz, b.zz), PP), ec_mul(ec_mul(a.zzz, b.zzz), PPP)};
}

// k * p for a small unsigned k (left-to-right double-and-add)
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_mul_u32(const Xyzz<F>& p, u32 k) {
    Xyzz<F> r = Xyzz<F>::inf();
    for (int i = 31; i >= 0; --i) {
        r = xyzz_dbl(r);
        if ((k >> i) & 1) r = xyzz_add(r, p);
    }
    return r;
}
// scalar given as little-endian 32-bit limbs (canonical integer)
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_mul_limbs(const Xyzz<F>& p, const u32* k, int nlimbs) {
    Xyzz<F> r = Xyzz<F>::inf();
    for (int i = 32 * nlimbs - 1; i >= 0; --i) {
        r = xyzz_dbl(r);
        if ((k[i >> 5] >> (i & 31)) & 1) r = xyzz_add(r, p);
    }
    return r;
}

template <class F>
ZK_HD_CALL Aff<F> xyzz_to_affine(const Xyzz<F>& p) {
    if (p.is_inf()) return Aff<F>::inf();
    // one inversion: i3 = 1/ZZZ; the invariant ZZ^3 = ZZZ^2 gives 1/ZZ = (ZZ/ZZZ)^2
    F i3 = fe_inv(p.zzz);
    F i2 = ec_sqr(ec_mul(p.zz, i3));
    return {ec_mul(p.x, i2), ec_mul(p.y, i3)};
}

// ---- hot-loop variants for the MSM kernels ----
// Everything is inlined, including the exceptional cases (equal or opposite x): an out-of-line call would
// force the accumulator through scratch memory (its address escapes), which costs more than the few extra
// instructions of a doubling that is almost never executed.
// a += (neg ? -p : p), p affine and not infinity.  The sign of a signed bucket digit is applied where it is cheapest: y only
// enters the common path through the product S2 = ZZZ1 * y, which takes the negated y without a carry round (fe_cneg_for_mul);
// the rare paths (empty accumulator, doubling) negate properly.
template <bool HOT = false, class F>
ZK_HD void xyzz_madd_acc(Xyzz<F>& a, const Aff<F>& p, bool neg = false) {
    if (a.is_inf()) {
        a.x = p.x; a.y = neg ? fe_neg(p.y) : p.y; a.zz = F::one(); a.zzz = F::one();
        return;
    }
    // (products take the operand that may reach 4p first: the three-product Fq2 form bounds its first operand)
    F Pp = fe_sub_k<4>(ecm_k<HOT>(a.zz, p.x), a.x);         // X1 < 3p;  Pp < 6p
    F R = fe_sub_k<4>(ecm_k<HOT>(a.zzz, fe_cneg_for_mul(p.y, neg)), a.y);         // Y1 < 4p;  R < 6p
    if (fe_is_zero_modp(Pp)) {
        if (fe_is_zero_modp(R)) {
            const Aff<F> q{p.x, neg ? fe_neg(p.y) : p.y};
            a = xyzz_dbl_affine_inl<HOT>(q);
        } else {
            a = Xyzz<F>::inf();
        }
        return;
    }
    F PP = ecs<HOT>(Pp);
    F PPP = ecm<HOT>(Pp, PP);                                // Pp < 6p: the four-product form
    F Q = ecm_k<HOT>(a.x, PP);
    a.zz = ecm_k<HOT>(a.zz, PP);
    a.zzz = ecm_k<HOT>(a.zzz, PPP);
    F X3 = fe_relax(fu_x3_numerator(ecs<HOT>(R), PPP, Q));                    // R^2 - PPP - 2Q: < 10p before, < 3p after
    a.y = ec_mulsub<HOT>(R, fe_sub_k<4>(Q, X3), a.y, PPP);                    // one reduction: < 3p (G1 fused: < 2p)
    a.x = X3;
}
// The common case of a += (+-)p for the accumulation kernel's fast path, cut in two so that the kernel can look at Pp before it
// commits: xyzz_madd_begin computes Pp = U2 - X1 and R = S2 - Y1 (ys: the base's y with the digit's sign applied, normalised);
// xyzz_madd_finish turns them into the sum, branch-free — the caller has established that a is not empty and Pp != 0 mod p.
template <bool HOT, class F>
ZK_HD void xyzz_madd_begin(const Xyzz<F>& a, const F& px, const F& ys, F& Pp, F& R) {
    Pp = fe_sub_k<4>(ecm_k<HOT>(a.zz, px), a.x);            // X1 < 3p;  Pp < 6p
    R = fe_sub_k<4>(ecm_k<HOT>(a.zzz, ys), a.y);            // Y1 < 4p;  R < 6p
}
template <bool HOT, class F>
ZK_HD void xyzz_madd_finish(Xyzz<F>& a, const F& Pp, const F& R) {
    F PP = ecs<HOT>(Pp);
    F PPP = ecm<HOT>(Pp, PP);
    F Q = ecm_k<HOT>(a.x, PP);
    a.zz = ecm_k<HOT>(a.zz, PP);
    a.zzz = ecm_k<HOT>(a.zzz, PPP);
    F X3 = fe_relax(fu_x3_numerator(ecs<HOT>(R), PPP, Q));
    a.y = ec_mulsub<HOT>(R, fe_sub_k<4>(Q, X3), a.y, PPP);
    a.x = X3;
}

// The same addition OUT OF LINE, for the accumulation kernel's general path (the wavefront's vote found an infinite base, a
// doubling or a cancellation: practically never on full-width scalars).  Inlined there, its registers — the doubling's
// temporaries next to the addition's — set the kernel's allocation and the hot loop pays for them in spills; as a call it has a
// frame of its own (arguments and result by value: nothing of the caller escapes to memory outside the cold branch).
#ifndef ZK_ACCUM_COLD_CALL
#define ZK_ACCUM_COLD_CALL 1
#endif
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_madd_cold(const Xyzz<F> a, const Aff<F> p) {
    Xyzz<F> t = a;
    xyzz_madd_acc<false>(t, p);
    return t;
}

#ifndef ZK_FOLD_DBL_CALL
#define ZK_FOLD_DBL_CALL 1
#endif
template <class F>
ZK_HD void xyzz_add_acc(Xyzz<F>& a, const Xyzz<F>& b) {   // a += b, both XYZZ
    if (b.is_inf()) return;
    if (a.is_inf()) { a = b; return; }
    F U1 = ec_mul(a.x, b.zz);
    F S1 = ec_mul(a.y, b.zzz);
    F Pp = fe_sub_k<2>(ec_mul(b.x, a.zz), U1);            // < 4p
    F R = fe_sub_k<2>(ec_mul(b.y, a.zzz), S1);
    if (fe_is_zero_modp(Pp)) {
        // (equal points: practically never among bucket sums — the doubling is a CALL, so that its temporaries do not set the register
        // count of the fold kernels, whose waves sit beside the accumulation's: ZK_FOLD_DBL_CALL)
        a = fe_is_zero_modp(R) ? (ZK_FOLD_DBL_CALL ? xyzz_dbl(a) : xyzz_dbl_inl(a)) : Xyzz<F>::inf();
        return;
    }
    F PP = ec_sqr(Pp);
    F PPP = ec_mul(Pp, PP);
    F Q = ec_mul(U1, PP);
    a.zz = ec_mul(ec_mul(a.zz, b.zz), PP);
    a.zzz = ec_mul(ec_mul(a.zzz, b.zzz), PPP);
    F X3 = fe_relax(fe_sub_k<4>(fe_sub_k<2>(ec_sqr(R), PPP), fe_dbl(Q)));
    a.y = fe_sub_k<2>(ec_mul(R, fe_sub_k<4>(Q, X3)), ec_mul(S1, PPP));
    a.x = X3;
}

// a += *bp for the fold kernels, whose second operand always sits in memory (LDS or the partial sums in HBM): the same products as
// xyzz_add_acc in an order that lets every coordinate of b be fetched where it is used and die at once, and every coordinate of a
// be overwritten by what replaces it (X1 by U1, Y1 by S1, ZZ1 by ZZ1 ZZ2, ...): nine field elements alive at the widest point, and
// the addition fits the accumulation kernels' register budgets (128 for G1, 256 for G2) without a spill.
// The doubling is not a second formula but a second set of INPUTS to the same tail: dbl-2008-s-1 is add-2008-s with
//     Pp := 2 Y,  U1 := X,  R := 3 X^2,  S1 := Y,  ZZ1 ZZ2 := ZZ,  ZZZ1 ZZZ2 := ZZZ     and X3 without its "- PPP" term,
// so two equal bucket sums (practically never) reload b, take ONE out-of-line square and rejoin the common code.  Round 5's fold
// kernels called an out-of-line doubling instead: its frame — the point in and out, everything alive across the call — was the
// 0.65 KB (G1) to 2 KB (G2) of scratch per lane those kernels reserved on every queue they ran on, untouched by their hot path.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_MEM_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)   /* neither loads nor arithmetic move across it */
#else
#define ZK_MEM_ORDER() ((void)0)
#endif
template <class P> ZK_HD Fu<P> ec_sqr_cold(const Fu<P>& a) { return fu_mul(a, a); }
template <class P> ZK_HD Fu2<P> ec_sqr_cold(const Fu2<P>& a) { return fu2_sqr_call(a); }
template <class P> ZK_HD Fe<P> ec_sqr_cold(const Fe<P>& a) { return ec_sqr(a); }
template <class P> ZK_HD Fe2<P> ec_sqr_cold(const Fe2<P>& a) { return ec_sqr(a); }
// (neg: a -= *bp — b's Y is negated where it is fetched)
template <class F>
ZK_HD F xyzz_y_of(const Xyzz<F>* bp, bool neg) {
    const F y = bp->y;
    return neg ? fe_relax(fe_sub_k<4>(F::zero(), y)) : y;      // Y < 4p -> 4p - Y, relaxed: < 3p
}
template <class F>
ZK_HD void xyzz_add_from(Xyzz<F>& a, const Xyzz<F>* bp, bool neg = false) {
    F U1, S1, Pp, R;
    {
        const F bzz = bp->zz;
        if (bzz.is_zero()) return;
        if (a.is_inf()) { a = *bp; if (neg) a.y = xyzz_y_of(bp, true); return; }
        U1 = ec_mul(a.x, bzz);                               // X1 dies here
        ZK_MEM_ORDER();
        Pp = fe_sub_k<2>(ec_mul(bp->x, a.zz), U1);           // < 4p
        ZK_MEM_ORDER();
        a.zz = ec_mul(a.zz, bzz);                            // ZZ1 ZZ2 (times PP below)
    }
    ZK_MEM_ORDER();
    {
        const F bzzz = bp->zzz;
        S1 = ec_mul(a.y, bzzz);                              // Y1 dies here
        ZK_MEM_ORDER();
        R = fe_sub_k<2>(ec_mul(xyzz_y_of(bp, neg), a.zzz), S1);
        ZK_MEM_ORDER();
        a.zzz = ec_mul(a.zzz, bzzz);
    }
    ZK_MEM_ORDER();
    bool dbl = false;
    if (fe_is_zero_modp(Pp)) {
        // equal x: opposite points, or the same one — then a + b = 2b, and b is still where it was
        if (!fe_is_zero_modp(R)) { a = Xyzz<F>::inf(); return; }
        {
            const F bx = bp->x;
            const F X2 = ec_sqr_cold(bx);                    // (nothing but addresses is alive across this call)
            R = fe_add(fe_dbl(X2), X2);                      // M = 3 X^2 < 6p
        }
        ZK_MEM_ORDER();
        S1 = xyzz_y_of(bp, neg);
        Pp = fe_dbl(S1);                                     // U = 2 Y < 8p
        if (fe_is_zero_modp(Pp)) { a = Xyzz<F>::inf(); return; }       // (a point of order two: on neither curve's r-torsion)
        U1 = bp->x;
        a.zz = bp->zz;
        a.zzz = bp->zzz;
        dbl = true;
    }
    const F PP = ec_sqr(Pp);                                 // V
    ZK_MEM_ORDER();
    a.zz = ec_mul(a.zz, PP);
    ZK_MEM_ORDER();
    const F PPP = ec_mul(Pp, PP);                            // W
    ZK_MEM_ORDER();
    a.zzz = ec_mul(a.zzz, PPP);
    ZK_MEM_ORDER();
    const F Q = ec_mul(U1, PP);                              // S
    ZK_MEM_ORDER();
    const F X3 = fe_relax(fe_sub_k<4>(fe_sub_k<2>(ec_sqr(R), fe_select(dbl, F::zero(), PPP)), fe_dbl(Q)));
    ZK_MEM_ORDER();
    const F t = ec_mul(R, fe_sub_k<4>(Q, X3));
    ZK_MEM_ORDER();
    a.y = fe_sub_k<2>(t, ec_mul(S1, PPP));
    a.x = X3;
}

// a = 2a in place (dbl-2008-s-1), products in an order that overwrites every coordinate where it dies: the running point of a
// register-resident double-and-add loop (bind.cuh) — at most a, M, U, V alive beside one product's columns
template <class F>
ZK_HD void xyzz_dbl_acc(Xyzz<F>& a) {
    if (a.is_inf()) return;
    const F U = fe_dbl(a.y);                                 // < 8p
    if (fe_is_zero_modp(U)) { a = Xyzz<F>::inf(); return; }
    F M;
    {
        const F X2 = ec_sqr(a.x);
        M = fe_add(fe_dbl(X2), X2);                          // < 6p
    }
    ZK_MEM_ORDER();
    const F V = ec_sqr(U);
    ZK_MEM_ORDER();
    a.zz = ec_mul(V, a.zz);
    ZK_MEM_ORDER();
    const F S = ec_mul(a.x, V);                              // X dies here
    ZK_MEM_ORDER();
    const F W = ec_mul(U, V);                                // U, V die here
    ZK_MEM_ORDER();
    a.zzz = ec_mul(W, a.zzz);
    ZK_MEM_ORDER();
    const F t = ec_mul(W, a.y);                              // Y, W die here
    ZK_MEM_ORDER();
    const F X3 = fe_relax(fe_sub_k<4>(ec_sqr(M), fe_dbl(S)));
    ZK_MEM_ORDER();
    a.y = fe_sub_k<2>(ec_mul(M, fe_sub_k<4>(S, X3)), t);
    a.x = X3;
}

// out-of-line a += b for the cold kernels (fold, heavy-bucket reduction): one copy of the addition per point
// type keeps their code small; the accumulator then lives in scratch memory, which is fine off the hot path.
// It deliberately wraps the out-of-line xyzz_add (whose doubling case is a further call): a single large
// function with the doubling inlined hangs on gfx950 for Fq2 (ROCm 7.2 code generation; tools/fold_probe.hip).
template <class F>
ZK_HD_CALL void xyzz_add_to(Xyzz<F>* a, const Xyzz<F>* b) {
    *a = xyzz_add(*a, *b);
}

}  // namespace zk
