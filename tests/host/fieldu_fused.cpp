// TEST ONLY: the fused products of the hot mixed addition (fu2_mul_kara: Karatsuba Fq2 product with two side-by-side
// reductions; fu_mul4_inl / fu2_mulsub_inl / ec_mulsub: a*b - c*d with one reduction) against the plain products of
// fieldu.cuh, random and edge operands (zero components, values up to 3p), both base fields.
#include <cstdio>
#include <cstdlib>
#include <random>
#define ZK_FU_MUL_INLINE 1
#include "ec.cuh"
using namespace zk;
template <class P> static Fu<P> rnd(std::mt19937_64& g, int mult) {   // a TIGHT element with value up to ~mult * p
    Fe<P> x;
    for (int i = 0; i < P::N; ++i) x.v[i] = (u32)g();
    x.v[P::N - 1] &= 0x0fffffff;
    while (!(
        [&] { for (int i = P::N - 1; i >= 0; --i) if (x.v[i] != P::mod(i)) return x.v[i] < P::mod(i); return false; }())) x.v[P::N - 1] >>= 1;
    Fu<P> u = fu_from_fe(fe_to_mont(x));
    Fu<P> r = u;
    for (int k = 1; k < mult; ++k) r = fe_add(r, u);
    return r;
}
template <class P> static bool eq(const Fu<P>& a, const Fu<P>& b) { return fu_to_fe(a).equals(fu_to_fe(b)); }
template <class P> int run(const char* name) {
    std::mt19937_64 g(12345);
    int bad = 0;
    for (int it = 0; it < 20000; ++it) {
        Fu2<P> a{rnd<P>(g, 1 + it % 3), rnd<P>(g, 1 + (it / 3) % 3)}, b{rnd<P>(g, 1), rnd<P>(g, 1)};
        if (it % 7 == 0) a.c1 = Fu<P>::zero();
        if (it % 11 == 0) b.c0 = Fu<P>::zero();
        Fu2<P> want = fu2_mul_inl(a, b), got = fu2_mul_kara(a, b);
        if (!eq(want.c0, got.c0) || !eq(want.c1, got.c1)) ++bad;
        Fu2<P> c{rnd<P>(g, 1 + it % 4), rnd<P>(g, 2)}, d{rnd<P>(g, 1), rnd<P>(g, 1)};
        Fu2<P> w2 = fe_sub_k<2>(fu2_mul_inl(a, b), fu2_mul_inl(c, d)), g2 = fu2_mulsub_inl(a, b, c, d);
        if (!eq(w2.c0, g2.c0) || !eq(w2.c1, g2.c1)) ++bad;
        Fu<P> w1 = fe_sub_k<2>(fu_mul_inl(a.c0, b.c0), fu_mul_inl(c.c0, d.c0)), g1 = ec_mulsub<true>(a.c0, b.c0, c.c0, d.c0);
        if (!eq(w1, g1)) ++bad;
    }
    // the hot mixed addition (three-product Fq2 form, fused Y3) against the cold one and against the saturated field:
    // chains of additions of arbitrary (x, y) pairs — the chord formulas are identities of rational functions
    for (int it = 0; it < 300; ++it) {
        Xyzz<Fu<P>> h1 = Xyzz<Fu<P>>::inf(), c1 = h1;
        Xyzz<Fu2<P>> h2 = Xyzz<Fu2<P>>::inf(), c2 = h2;
        Xyzz<Fe<P>> s1 = Xyzz<Fe<P>>::inf();
        for (int k = 0; k < 12; ++k) {
            Aff<Fu<P>> p1{rnd<P>(g, 1), rnd<P>(g, 1)};
            Aff<Fu2<P>> p2{{rnd<P>(g, 1), rnd<P>(g, 1)}, {rnd<P>(g, 1), rnd<P>(g, 1)}};
            if (k == 5) { p1.y = fe_neg(p1.y); p2.y = fe_neg(p2.y); }
            xyzz_madd_acc<true>(h1, p1); xyzz_madd_acc<false>(c1, p1);
            xyzz_madd_acc<true>(h2, p2); xyzz_madd_acc<false>(c2, p2);
            s1 = xyzz_madd(s1, Aff<Fe<P>>{fu_to_fe(p1.x), fu_to_fe(p1.y)});
        }
        // compare as fractions: X / ZZ and Y / ZZZ
        auto same1 = [&](const Xyzz<Fu<P>>& a, const Xyzz<Fu<P>>& b) {
            return eq(fu_mul_inl(a.x, b.zz), fu_mul_inl(b.x, a.zz)) && eq(fu_mul_inl(a.y, b.zzz), fu_mul_inl(b.y, a.zzz));
        };
        auto same2 = [&](const Xyzz<Fu2<P>>& a, const Xyzz<Fu2<P>>& b) {
            Fu2<P> l = fu2_mul_inl(a.x, b.zz), r = fu2_mul_inl(b.x, a.zz), l2 = fu2_mul_inl(a.y, b.zzz), r2 = fu2_mul_inl(b.y, a.zzz);
            return eq(l.c0, r.c0) && eq(l.c1, r.c1) && eq(l2.c0, r2.c0) && eq(l2.c1, r2.c1);
        };
        if (!same1(h1, c1) || !same2(h2, c2)) ++bad;
        Xyzz<Fu<P>> s1u{fu_from_fe(s1.x), fu_from_fe(s1.y), fu_from_fe(s1.zz), fu_from_fe(s1.zzz)};
        if (!same1(h1, s1u)) ++bad;
    }
    printf("%s: %d mismatches\n", name, bad);
    return bad;
}
int main() {
    const int bad = run<Bn254Fq>("bn254") + run<Bls381Fq>("bls12_381");
    printf("%d failures\n", bad);
    return bad != 0;
}
