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
#include "field.cuh"

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
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_dbl_affine(const Aff<F>& p) {
    if (p.is_inf() || p.y.is_zero()) return Xyzz<F>::inf();
    F U = fe_dbl(p.y);
    F V = fe_sqr(U);
    F W = fe_mul(U, V);
    F S = fe_mul(p.x, V);
    F X2 = fe_sqr(p.x);
    F M = fe_add(fe_dbl(X2), X2);
    F X3 = fe_sub(fe_sqr(M), fe_dbl(S));
    F Y3 = fe_sub(fe_mul(M, fe_sub(S, X3)), fe_mul(W, p.y));
    return {X3, Y3, V, W};
}

// dbl-2008-s-1
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_dbl(const Xyzz<F>& p) {
    if (p.is_inf() || p.y.is_zero()) return Xyzz<F>::inf();
    F U = fe_dbl(p.y);
    F V = fe_sqr(U);
    F W = fe_mul(U, V);
    F S = fe_mul(p.x, V);
    F X2 = fe_sqr(p.x);
    F M = fe_add(fe_dbl(X2), X2);
    F X3 = fe_sub(fe_sqr(M), fe_dbl(S));
    F Y3 = fe_sub(fe_mul(M, fe_sub(S, X3)), fe_mul(W, p.y));
    return {X3, Y3, fe_mul(V, p.zz), fe_mul(W, p.zzz)};
}

// acc + affine (madd-2008-s) with the exceptional cases handled
template <class F>
ZK_HD Xyzz<F> xyzz_madd(const Xyzz<F>& a, const Aff<F>& p) {
    if (p.is_inf()) return a;
    if (a.is_inf()) return {p.x, p.y, F::one(), F::one()};
    F U2 = fe_mul(p.x, a.zz);
    F S2 = fe_mul(p.y, a.zzz);
    F Pp = fe_sub(U2, a.x);
    F R = fe_sub(S2, a.y);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_affine(p);
        return Xyzz<F>::inf();
    }
    F PP = fe_sqr(Pp);
    F PPP = fe_mul(Pp, PP);
    F Q = fe_mul(a.x, PP);
    F X3 = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    F Y3 = fe_sub(fe_mul(R, fe_sub(Q, X3)), fe_mul(a.y, PPP));
    return {X3, Y3, fe_mul(a.zz, PP), fe_mul(a.zzz, PPP)};
}

// general add (add-2008-s)
template <class F>
ZK_HD_CALL Xyzz<F> xyzz_add(const Xyzz<F>& a, const Xyzz<F>& b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F U1 = fe_mul(a.x, b.zz);
    F U2 = fe_mul(b.x, a.zz);
    F S1 = fe_mul(a.y, b.zzz);
    F S2 = fe_mul(b.y, a.zzz);
    F Pp = fe_sub(U2, U1);
    F R = fe_sub(S2, S1);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl(a);
        return Xyzz<F>::inf();
    }
    F PP = fe_sqr(Pp);
    F PPP = fe_mul(Pp, PP);
    F Q = fe_mul(U1, PP);
    F X3 = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    F Y3 = fe_sub(fe_mul(R, fe_sub(Q, X3)), fe_mul(S1, PPP));
    return {X3, Y3, fe_mul(fe_mul(a.zz, b.zz), PP), fe_mul(fe_mul(a.zzz, b.zzz), PPP)};
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
    F i2 = fe_sqr(fe_mul(p.zz, i3));
    return {fe_mul(p.x, i2), fe_mul(p.y, i3)};
}

}  // namespace zk
