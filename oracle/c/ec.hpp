// oracle/c/ec.hpp — TEST ORACLE ONLY.
// Short-Weierstrass a = 0 groups in Jacobian coordinates with mixed addition, restating
// [UPSTREAM] ark-ec 0.3.0 `short_weierstrass_jacobian::{GroupAffine, GroupProjective}`
// (add-2007-bl, madd-2007-bl, dbl-2009-l) — SURVEY.md App. A.1/A.5.
#pragma once
#include <vector>
#include "ff.hpp"

namespace orc {

template <class F>
struct Affine {
    F x, y;
    bool inf;
    static Affine infinity() { return {F::zero(), F::one(), true}; }
    Affine neg() const { return {x, y.neg(), inf}; }
    bool operator==(const Affine& o) const { return inf == o.inf && (inf || (x == o.x && y == o.y)); }
    static constexpr int BYTES = 2 * F::BYTES;
    // ark `serialize_uncompressed`: x | y, infinity flag = bit 6 of the last byte (App. B.3)
    static Affine from_bytes(const uint8_t* b) {
        uint8_t tmp[F::BYTES];
        memcpy(tmp, b + F::BYTES, F::BYTES);
        bool inf = tmp[F::BYTES - 1] & 0x40;
        tmp[F::BYTES - 1] &= 0x3f;
        return {F::from_bytes(b), F::from_bytes(tmp), inf};
    }
    void to_bytes(uint8_t* b) const {
        if (inf) { F::zero().to_bytes(b); F::one().to_bytes(b + F::BYTES); b[2 * F::BYTES - 1] |= 0x40; }
        else { x.to_bytes(b); y.to_bytes(b + F::BYTES); }
    }
};

template <class F>
struct Jac {
    F X, Y, Z;
    static Jac infinity() { return {F::one(), F::one(), F::zero()}; }
    static Jac from_affine(const Affine<F>& a) { return a.inf ? infinity() : Jac{a.x, a.y, F::one()}; }
    bool is_inf() const { return Z.is_zero(); }
    Jac neg() const { return {X, Y.neg(), Z}; }

    Jac dbl() const {
        if (is_inf() || Y.is_zero()) return infinity();
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = ((X + B).sqr() - A - C).dbl();
        F E = A.dbl() + A;
        F Fv = E.sqr();
        F X3 = Fv - D.dbl();
        F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
        F Z3 = (Y * Z).dbl();
        return {X3, Y3, Z3};
    }
    Jac add(const Jac& o) const {
        if (is_inf()) return o;
        if (o.is_inf()) return *this;
        F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        F U1 = X * Z2Z2, U2 = o.X * Z1Z1;
        F S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
        if (U1 == U2) return S1 == S2 ? dbl() : infinity();
        F H = U2 - U1, R = S2 - S1;
        F HH = H.sqr(), HHH = H * HH, V = U1 * HH;
        F X3 = R.sqr() - HHH - V.dbl();
        F Y3 = R * (V - X3) - S1 * HHH;
        F Z3 = Z * o.Z * H;
        return {X3, Y3, Z3};
    }
    Jac add_mixed(const Affine<F>& o) const {
        if (o.inf) return *this;
        if (is_inf()) return from_affine(o);
        F Z1Z1 = Z.sqr();
        F U2 = o.x * Z1Z1, S2 = o.y * Z * Z1Z1;
        if (X == U2) return Y == S2 ? dbl() : infinity();
        F H = U2 - X, R = S2 - Y;
        F HH = H.sqr(), HHH = H * HH, V = X * HH;
        F X3 = R.sqr() - HHH - V.dbl();
        F Y3 = R * (V - X3) - Y * HHH;
        F Z3 = Z * H;
        return {X3, Y3, Z3};
    }
    Affine<F> to_affine() const {
        if (is_inf()) return Affine<F>::infinity();
        F zi = Z.inverse(), zi2 = zi.sqr();
        return {X * zi2, Y * zi2 * zi, false};
    }
    // scalar = little-endian u64 limbs (canonical integer)
    Jac mul_limbs(const u64* k, int n) const {
        Jac r = infinity();
        for (int i = 64 * n - 1; i >= 0; --i) {
            r = r.dbl();
            if ((k[i / 64] >> (i % 64)) & 1) r = r.add(*this);
        }
        return r;
    }
};

// Batch normalisation (Montgomery's trick), infinity-safe.
template <class F>
void batch_to_affine(const std::vector<Jac<F>>& in, std::vector<Affine<F>>& out) {
    size_t n = in.size();
    out.resize(n);
    std::vector<F> pref(n);
    F acc = F::one();
    for (size_t i = 0; i < n; ++i) {
        pref[i] = acc;
        if (!in[i].is_inf()) acc = acc * in[i].Z;
    }
    F inv = acc.inverse();
    for (size_t i = n; i-- > 0;) {
        if (in[i].is_inf()) { out[i] = Affine<F>::infinity(); continue; }
        F zi = inv * pref[i];
        inv = inv * in[i].Z;
        F zi2 = zi.sqr();
        out[i] = {in[i].X * zi2, in[i].Y * zi2 * zi, false};
    }
}

}  // namespace orc
