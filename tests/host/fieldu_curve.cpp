#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ec.cuh"
using namespace zk;
template <class P> Fe<P> rnd_fe() {
    Fe<P> r;
    for (int i = 0; i < P::N; ++i) r.v[i] = (u32)rand() * 2654435761u + (u32)rand();
    r.v[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;
    return r;
}
template <class P> Fe<P> sat(const Fu<P>& a) { return fu_to_fe(a); }
template <class P> Fe2<P> sat(const Fu2<P>& a) { return fu_to_fe(a); }
template <class FS, class U> bool same(const Xyzz<FS>& a, const Xyzz<U>& b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() == b.is_inf();
    return a.x.equals(sat(b.x)) && a.y.equals(sat(b.y)) && a.zz.equals(sat(b.zz)) && a.zzz.equals(sat(b.zzz));
}
template <class P> void rnd(Fe<P>& x) { x = rnd_fe<P>(); }
template <class P> void rnd(Fe2<P>& x) { x = {rnd_fe<P>(), rnd_fe<P>()}; }
template <class FS, class U> int test(const char* name) {
    int bad = 0;
    for (int it = 0; it < 300; ++it) {
        Aff<FS> p0, p1; rnd(p0.x); rnd(p0.y); rnd(p1.x); rnd(p1.y);
        Aff<U> q0{fu_from_fe(p0.x), fu_from_fe(p0.y)}, q1{fu_from_fe(p1.x), fu_from_fe(p1.y)};
        Xyzz<FS> a = Xyzz<FS>::inf(); Xyzz<U> b = Xyzz<U>::inf();
        xyzz_madd_acc(a, p0); xyzz_madd_acc(b, q0);
        if (!same(a, b)) { if (bad++ < 3) printf("%s first madd mismatch\n", name); }
        for (int k = 0; k < 6; ++k) {
            xyzz_madd_acc(a, p1); xyzz_madd_acc(b, q1);
            if (!same(a, b)) { if (bad++ < 3) printf("%s madd %d mismatch\n", name, k); break; }
            Aff<FS> t = p1; p1 = p0; p0 = t; Aff<U> tu = q1; q1 = q0; q0 = tu;
        }
        Xyzz<FS> c = a; Xyzz<U> d = b;
        xyzz_madd_acc(c, p0); xyzz_madd_acc(d, q0);
        Xyzz<FS> e = a; Xyzz<U> f = b;
        xyzz_add_acc(e, c); xyzz_add_acc(f, d);
        if (!same(e, f)) { if (bad++ < 3) printf("%s add mismatch\n", name); }
        Xyzz<FS> g = xyzz_dbl_inl(e); Xyzz<U> h = xyzz_dbl_inl(f);
        if (!same(g, h)) { if (bad++ < 3) printf("%s dbl mismatch\n", name); }
        // a + a through the add path must take the doubling branch
        Xyzz<FS> e2 = e; Xyzz<U> f2 = f;
        xyzz_add_acc(e2, e); xyzz_add_acc(f2, f);
        if (!same(e2, f2) || !same(g, f2)) { if (bad++ < 3) printf("%s add->dbl mismatch\n", name); }
        Xyzz<FS> g2 = xyzz_dbl_affine_inl(p0); Xyzz<U> h2 = xyzz_dbl_affine_inl(q0);
        if (!same(g2, h2)) { if (bad++ < 3) printf("%s dbl_affine mismatch\n", name); }
        // madd of the same point twice -> doubling branch
        Xyzz<FS> m = Xyzz<FS>::inf(); Xyzz<U> n = Xyzz<U>::inf();
        xyzz_madd_acc(m, p0); xyzz_madd_acc(m, p0); xyzz_madd_acc(n, q0); xyzz_madd_acc(n, q0);
        if (!same(m, n) || !same(g2, n)) { if (bad++ < 3) printf("%s madd->dbl mismatch\n", name); }
        // P + (-P)
        Aff<FS> np = p0; np.y = fe_neg(np.y); Aff<U> nq = q0; nq.y = fe_neg(nq.y);
        Xyzz<FS> z1 = Xyzz<FS>::inf(); Xyzz<U> z2 = Xyzz<U>::inf();
        xyzz_madd_acc(z1, p0); xyzz_madd_acc(z1, np); xyzz_madd_acc(z2, q0); xyzz_madd_acc(z2, nq);
        if (!z1.is_inf() || !z2.is_inf()) { if (bad++ < 3) printf("%s P-P mismatch %d %d\n", name, z1.is_inf(), z2.is_inf()); }
        // the sign of a bucket digit handed to the mixed addition (no separate negation of the base) == adding the negated point:
        // common path, empty accumulator, doubling (acc = -P, then P with the sign) and cancellation (acc = P, then P with the sign)
        Xyzz<FS> s1 = a; Xyzz<U> s2 = b;
        xyzz_madd_acc(s1, np); xyzz_madd_acc(s2, q0, true);
        if (!same(s1, s2)) { if (bad++ < 3) printf("%s signed madd mismatch\n", name); }
        Xyzz<FS> s3 = Xyzz<FS>::inf(); Xyzz<U> s4 = Xyzz<U>::inf();
        xyzz_madd_acc(s3, np); xyzz_madd_acc(s4, q0, true);
        if (!same(s3, s4)) { if (bad++ < 3) printf("%s signed madd from empty mismatch\n", name); }
        xyzz_madd_acc(s3, np); xyzz_madd_acc(s4, q0, true);
        if (!same(s3, s4) || s4.is_inf()) { if (bad++ < 3) printf("%s signed madd->dbl mismatch\n", name); }
        Xyzz<U> s5 = Xyzz<U>::inf();
        xyzz_madd_acc(s5, q0); xyzz_madd_acc(s5, q0, true);
        if (!s5.is_inf()) { if (bad++ < 3) printf("%s signed P-P mismatch\n", name); }
        // a long chain with alternating signs keeps every bound (ZK_CHECK_OVERFLOW aborts on a column that leaves 64 bits)
        for (int k = 0; k < 40; ++k) {
            xyzz_madd_acc(a, (k & 1) ? np : p1); xyzz_madd_acc(b, (k & 1) ? q0 : q1, (k & 1) != 0);
            if (!same(a, b)) { if (bad++ < 3) printf("%s signed chain %d mismatch\n", name, k); break; }
        }
    }
    printf("%s: %d failures\n", name, bad);
    return bad;
}
int main() {
    int b = 0;
    b += test<Fe<Bn254Fq>, Fu<Bn254Fq>>("bn254 G1");
    b += test<Fe2<Bn254Fq>, Fu2<Bn254Fq>>("bn254 G2");
    b += test<Fe<Bls381Fq>, Fu<Bls381Fq>>("bls381 G1");
    b += test<Fe2<Bls381Fq>, Fu2<Bls381Fq>>("bls381 G2");
    return b;
}
