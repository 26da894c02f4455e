#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ec.cuh"
using namespace zk;
template <class P> Fe<P> rnd_fe() {
    Fe<P> r;
    for (int i = 0; i < P::N; ++i) r.v[i] = (u32)rand() * 2654435761u + (u32)rand();
    r.v[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;   // < 2^(BITS-1) < p
    return r;
}
template <class P> bool eq(const Fe<P>& a, const Fe<P>& b) { return a.equals(b); }
template <class P> int test(const char* name) {
    int bad = 0;
    typedef Fe<P> F; typedef Fu<P> U;
    for (int it = 0; it < 2000; ++it) {
        F a = rnd_fe<P>(), b = rnd_fe<P>();   // treat as Montgomery-form values
        U ua = fu_from_fe(a), ub = fu_from_fe(b);
        if (!eq(fu_to_fe(ua), a)) { if (bad++ < 3) printf("%s roundtrip mismatch\n", name); }
        if (!eq(fu_to_fe(fu_mul_inl(ua, ub)), fe_mul(a, b))) { if (bad++ < 3) printf("%s mul mismatch\n", name); }
        if (!eq(fu_to_fe(fe_add(ua, ub)), fe_add(a, b))) { if (bad++ < 3) printf("%s add mismatch\n", name); }
        if (!eq(fu_to_fe(fe_sub_k<2>(ua, ub)), fe_sub(a, b))) { if (bad++ < 3) printf("%s sub2 mismatch\n", name); }
        if (!eq(fu_to_fe(fe_sub_k<4>(ua, ub)), fe_sub(a, b))) { if (bad++ < 3) printf("%s sub4 mismatch\n", name); }
        if (!eq(fu_to_fe(fe_sub_k<8>(ua, ub)), fe_sub(a, b))) { if (bad++ < 3) printf("%s sub8 mismatch\n", name); }
        if (!eq(fu_to_fe(fe_dbl(ua)), fe_dbl(a))) { if (bad++ < 3) printf("%s dbl mismatch\n", name); }
        U big = fe_sub_k<8>(fe_sub_k<8>(fe_add(ua, ub), ub), ua);   // value up to ~ 20p, == 0 mod p... = a+b-b-a
        if (!fe_is_zero_modp(fe_relax(big)) ) { if (bad++ < 3) printf("%s zero-modp (relaxed) mismatch\n", name); }
        U big2 = fe_sub_k<8>(fe_add(ua, ub), ub);   // == a
        if (!eq(fu_to_fe(fe_relax(big2)), a)) { if (bad++ < 3) printf("%s relax mismatch\n", name); }
        if (fe_is_zero_modp(fe_sub_k<4>(ua, ub)) != a.equals(b)) { if (bad++ < 3) printf("%s iszero mismatch\n", name); }
        if (!fe_is_zero_modp(fe_sub_k<4>(ua, ua))) { if (bad++ < 3) printf("%s iszero(a-a) mismatch\n", name); }
        if (!eq(fu_to_fe(fu_mul2_inl(ua, ub, ub, ua)), fe_dbl(fe_mul(a, b)))) { if (bad++ < 3) printf("%s mul2 mismatch\n", name); }
        if (!eq(fu_to_fe(fu_sqr_inl(ua)), fe_mul(a, a))) { if (bad++ < 3) printf("%s sqr mismatch\n", name); }
        if (!eq(fu_to_fe(fu_sqr_inl(fe_sub_k<4>(ua, ub))), fe_sqr(fe_sub(a, b)))) { if (bad++ < 3) printf("%s sqr(loose) mismatch\n", name); }
        // Fq2
        Fe2<P> A{a, b}, Bq{b, fe_add(a, a)};
        Fu2<P> uA = fu_from_fe(A), uB = fu_from_fe(Bq);
        Fe2<P> m1 = fe_mul(A, Bq), m2 = fu_to_fe(ec_mul(uA, uB));
        if (!m1.equals(m2)) { if (bad++ < 3) printf("%s fq2 mul mismatch\n", name); }
        Fe2<P> s1 = fe_sqr(A), s2 = fu_to_fe(ec_sqr(uA));
        if (!s1.equals(s2)) { if (bad++ < 3) printf("%s fq2 sqr mismatch\n", name); }
    }
    printf("%s: %d failures\n", name, bad);
    return bad;
}
int main() { return test<Bn254Fq>("bn254") + test<Bls381Fq>("bls381"); }
