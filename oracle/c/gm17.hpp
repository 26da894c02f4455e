// gm17.hpp — C++ CPU restatement of [UPSTREAM] ark-gm17 0.3.0 (R1CStoSAP, generate_parameters, create_proof), the
// algorithm behind /root/reference/zokrates_ark/src/gm17.rs:19-78.  TEST ORACLE / CPU BASELINE ONLY.
// Included by oracle_g16.cpp inside namespace orc (it reuses Domain, msm, FixedBase, Circuit).  Mirrors oracle/gm17.py,
// which documents the SAP layout and what pins this restatement (in-tree verification equations, closed form,
// uniqueness of the proof for fixed r + d1).
#pragma once

template <class C>
struct PkGm17 : PkBase {
    typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    Affine<Fq2> h_g2; Affine<Fq> g_alpha_g1; Affine<Fq2> h_beta_g2; Affine<Fq> g_gamma_g1; Affine<Fq2> h_gamma_g2;
    std::vector<Affine<Fq>> query;
    std::vector<Affine<Fq>> a_query; std::vector<Affine<Fq2>> b_query; std::vector<Affine<Fq>> c_query_1, c_query_2;
    Affine<Fq> g_gamma_z; Affine<Fq2> h_gamma_z; Affine<Fq> g_ab_gamma_z, g_gamma2_z2;
    std::vector<Affine<Fq>> g_gamma2_z_t;

    size_t byte_size() const {
        size_t g1 = Affine<Fq>::BYTES, g2 = Affine<Fq2>::BYTES;
        return 3 * g2 + 2 * g1 + 8 + query.size() * g1 + 8 + a_query.size() * g1 + 8 + b_query.size() * g2 + 8 + c_query_1.size() * g1 + 8 +
               c_query_2.size() * g1 + 3 * g1 + g2 + 8 + g_gamma2_z_t.size() * g1;
    }
    void serialize(uint8_t* p) const {   // ark_gm17::ProvingKey::serialize_unchecked: struct field order
        typedef Pk<C> P;
        p = P::put(p, h_g2); p = P::put(p, g_alpha_g1); p = P::put(p, h_beta_g2); p = P::put(p, g_gamma_g1); p = P::put(p, h_gamma_g2);
        p = P::putv(p, query);
        p = P::putv(p, a_query); p = P::putv(p, b_query); p = P::putv(p, c_query_1); p = P::putv(p, c_query_2);
        p = P::put(p, g_gamma_z); p = P::put(p, h_gamma_z); p = P::put(p, g_ab_gamma_z); p = P::put(p, g_gamma2_z2);
        p = P::putv(p, g_gamma2_z_t);
    }
    bool parse(const uint8_t* p, size_t len) {
        typedef Pk<C> P;
        typename P::Rd r{p, p + len};
        P::get(r, h_g2); P::get(r, g_alpha_g1); P::get(r, h_beta_g2); P::get(r, g_gamma_g1); P::get(r, h_gamma_g2);
        P::getv(r, query);
        P::getv(r, a_query); P::getv(r, b_query); P::getv(r, c_query_1); P::getv(r, c_query_2);
        P::get(r, g_gamma_z); P::get(r, h_gamma_z); P::get(r, g_ab_gamma_z); P::get(r, g_gamma2_z2);
        P::getv(r, g_gamma2_z_t);
        return r.ok && r.p == r.e;
    }
};

struct SapShape { u64 M, D0, D; };
template <class C>
static SapShape sap_shape(const Circuit<C>& cs) {
    SapShape s;
    s.M = 1 + 2 * (cs.l - 1) + cs.w + cs.n;
    s.D0 = 2 * cs.n + 2 * (cs.l - 1) + 1;
    s.D = 1; while (s.D < s.D0) s.D <<= 1;
    return s;
}

// R1CStoSAP::instance_map_with_evaluation: u_i(t), w_i(t) per SAP variable
template <class C>
struct SapT { std::vector<typename C::Fr> a, c; typename C::Fr zt; u64 D; };
template <class C>
static SapT<C> sap_at_t(const Circuit<C>& cs, const typename C::Fr& t) {
    typedef typename C::Fr Fr;
    const SapShape sh = sap_shape(cs);
    SapT<C> q;
    q.D = sh.D;
    Domain<C> dom(sh.D);
    u64 Dl[1] = {sh.D};
    q.zt = t.pow_limbs(Dl, 1) - Fr::one();
    std::vector<Fr> den(sh.D), wk(sh.D), pref(sh.D), u(sh.D);
    Fr p = Fr::one();
    for (u64 k = 0; k < sh.D; ++k) { wk[k] = p; den[k] = t - p; p = p * dom.omega; }
    Fr acc = Fr::one();
    for (u64 k = 0; k < sh.D; ++k) { pref[k] = acc; acc = acc * den[k]; }
    Fr inv = acc.inverse();
    Fr zn = q.zt * dom.n_inv;
    for (u64 k = sh.D; k-- > 0;) { Fr di = inv * pref[k]; inv = inv * den[k]; u[k] = zn * wk[k] * di; }
    const u64 n = cs.n, l = cs.l, m = cs.m();
    q.a.assign(sh.M, Fr::zero()); q.c.assign(sh.M, Fr::zero());
    std::thread ta([&] {
        for (u64 i = 0; i < n; ++i) {
            Fr u_add = u[2 * i] + u[2 * i + 1], u_sub = u[2 * i] - u[2 * i + 1];
            for (u64 k = cs.A.rowptr[i]; k < cs.A.rowptr[i + 1]; ++k) q.a[cs.A.col[k]] = q.a[cs.A.col[k]] + u_add * cs.A.val[k];
            for (u64 k = cs.B.rowptr[i]; k < cs.B.rowptr[i + 1]; ++k) q.a[cs.B.col[k]] = q.a[cs.B.col[k]] + u_sub * cs.B.val[k];
        }
    });
    for (u64 i = 0; i < n; ++i) {
        Fr u4 = u[2 * i] + u[2 * i]; u4 = u4 + u4;
        for (u64 k = cs.Cm.rowptr[i]; k < cs.Cm.rowptr[i + 1]; ++k) q.c[cs.Cm.col[k]] = q.c[cs.Cm.col[k]] + u4 * cs.Cm.val[k];
        q.c[m + i] = q.c[m + i] + u[2 * i] + u[2 * i + 1];
    }
    ta.join();
    q.a[0] = q.a[0] + u[2 * n];
    q.c[0] = q.c[0] + u[2 * n];
    for (u64 i = 1; i < l; ++i) {
        Fr u1 = u[2 * n + 2 * i - 1], u2 = u[2 * n + 2 * i];
        q.a[i] = q.a[i] + u1 + u2;
        q.a[0] = q.a[0] + u1 - u2;
        Fr u14 = u1 + u1; u14 = u14 + u14;
        q.c[i] = q.c[i] + u14;
        q.c[m + n - 1 + i] = q.c[m + n - 1 + i] + u1 + u2;
    }
    return q;
}

template <class C>
struct ToxicGm17 { typename C::Fr alpha, beta, gamma, t; };

// generate_parameters with fixed generators
template <class C>
static PkGm17<C>* setup_gm17(const Circuit<C>& cs, const ToxicGm17<C>& tx, int threads) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    auto q = sap_at_t(cs, tx.t);
    const u64 M = q.a.size(), l = cs.l;
    FixedBase<Fq> t1(C::g1(), Fr::P().bits);
    FixedBase<Fq2> t2(C::g2(), Fr::P().bits);
    auto* pk = new PkGm17<C>();
    const Fr g = tx.gamma, ab = tx.alpha + tx.beta, gz = g * q.zt, g2z = g * gz;
    std::vector<Affine<Fq>> tmp1; std::vector<Affine<Fq2>> tmp2;
    fixed_mul_vec(t1, std::vector<Fr>{tx.alpha, g, gz, ab * gz, gz * gz}, tmp1, 1);
    pk->g_alpha_g1 = tmp1[0]; pk->g_gamma_g1 = tmp1[1]; pk->g_gamma_z = tmp1[2]; pk->g_ab_gamma_z = tmp1[3]; pk->g_gamma2_z2 = tmp1[4];
    fixed_mul_vec(t2, std::vector<Fr>{Fr::one(), tx.beta, g, gz}, tmp2, 1);
    pk->h_g2 = tmp2[0]; pk->h_beta_g2 = tmp2[1]; pk->h_gamma_g2 = tmp2[2]; pk->h_gamma_z = tmp2[3];
    std::vector<Fr> s(M);
    for (u64 i = 0; i < M; ++i) s[i] = q.a[i] * g;
    fixed_mul_vec(t1, s, pk->a_query, threads);
    fixed_mul_vec(t2, s, pk->b_query, threads);
    std::vector<Fr> v(M);
    for (u64 i = 0; i < M; ++i) v[i] = g * q.c[i] + ab * q.a[i];
    s.assign(v.begin(), v.begin() + l);
    fixed_mul_vec(t1, s, pk->query, threads);
    s.resize(M - l);
    for (u64 i = l; i < M; ++i) s[i - l] = v[i] * g;
    fixed_mul_vec(t1, s, pk->c_query_1, threads);
    s.resize(M);
    const Fr dg2z = g2z + g2z;
    for (u64 i = 0; i < M; ++i) s[i] = q.a[i] * dg2z;
    fixed_mul_vec(t1, s, pk->c_query_2, threads);
    s.resize(q.D + 1);
    Fr p = g2z;
    for (u64 i = 0; i <= q.D; ++i) { s[i] = p; p = p * tx.t; }
    fixed_mul_vec(t1, s, pk->g_gamma2_z_t, threads);
    return pk;
}

// R1CStoSAP::witness_map: full_input_assignment (ext) and h (D + 1 coefficients)
template <class C>
static void witness_map_gm17(const Circuit<C>& cs, const std::vector<typename C::Fr>& z, const typename C::Fr& d1, const typename C::Fr& d2,
                             std::vector<typename C::Fr>& ext, std::vector<typename C::Fr>& h, int th) {
    typedef typename C::Fr Fr;
    const SapShape sh = sap_shape(cs);
    const u64 n = cs.n, l = cs.l, m = cs.m(), D = sh.D;
    Domain<C> dom(D);
    ext.assign(z.begin(), z.end());
    ext.resize(sh.M, Fr::zero());
    std::vector<Fr> a(D, Fr::zero()), c(D, Fr::zero());
    parallel_for(th, n, [&](size_t s, size_t e) {
        for (size_t i = s; i < e; ++i) {
            Fr ai = cs.row_dot(cs.A, i, z), bi = cs.row_dot(cs.B, i, z), ci = cs.row_dot(cs.Cm, i, z);
            Fr d = ai - bi, ev = d * d;
            ext[m + i] = ev;
            a[2 * i] = ai + bi; a[2 * i + 1] = d;
            Fr c4 = ci + ci; c4 = c4 + c4;
            c[2 * i] = c4 + ev; c[2 * i + 1] = ev;
        }
    });
    a[2 * n] = Fr::one(); c[2 * n] = Fr::one();
    for (u64 i = 1; i < l; ++i) {
        Fr d = z[i] - Fr::one(), f = d * d;
        ext[m + n - 1 + i] = f;
        a[2 * n + 2 * i - 1] = z[i] + Fr::one(); a[2 * n + 2 * i] = d;
        Fr x4 = z[i] + z[i]; x4 = x4 + x4;
        c[2 * n + 2 * i - 1] = x4 + f; c[2 * n + 2 * i] = f;
    }
    dom.ifft(a, th);
    const Fr d1_double = d1 + d1, d1d1 = d1 * d1;
    h.assign(D + 1, Fr::zero());
    parallel_for(th, D, [&](size_t s, size_t e) { for (size_t i = s; i < e; ++i) h[i] = d1_double * a[i]; });
    h[0] = h[0] - d2 - d1d1;
    h[D] = d1d1;
    dom.coset_fft(a, th);
    dom.ifft(c, th); dom.coset_fft(c, th);
    u64 Dl[1] = {D};
    Fr zinv = (dom.g.pow_limbs(Dl, 1) - Fr::one()).inverse();
    parallel_for(th, D, [&](size_t s, size_t e) { for (size_t i = s; i < e; ++i) a[i] = (a[i] * a[i] - c[i]) * zinv; });
    dom.coset_ifft(a, th);
    parallel_for(th, D - 1, [&](size_t s, size_t e) { for (size_t i = s; i < e; ++i) h[i] = h[i] + a[i]; });
}

// create_proof: the same terms, in the same grouping (inputs / aux split of every MSM)
template <class C>
static int prove_gm17(const Circuit<C>& cs, const PkGm17<C>& pk, const std::vector<typename C::Fr>& z, const typename C::Fr& d1,
                      const typename C::Fr& d2, const typename C::Fr& r, ProofT<C>& out, int th, Timings* tm) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    const SapShape sh = sap_shape(cs);
    const u64 M = sh.M, l = cs.l, D = sh.D;
    if (z.size() != cs.m() || pk.a_query.size() != M || pk.b_query.size() != M || pk.c_query_2.size() != M || pk.c_query_1.size() != M - l ||
        pk.g_gamma2_z_t.size() != D + 1)
        return -1;
    double t00 = now();
    std::vector<Fr> ext, h;
    witness_map_gm17(cs, z, d1, d2, ext, h, th);
    double t0 = now();
    const int NB = Fr::P().bits;
    std::vector<u64> er, hr;
    to_repr(ext.data(), M, er, th);
    to_repr(h.data(), D + 1, hr, th);
    const u64* in_s = er.data() + Fr::N;        // ext[1..l)
    const u64* aux_s = er.data() + l * Fr::N;   // ext[l..M)
    u64 rl[Fr::N], d1l[Fr::N], d2l[Fr::N], r2l[Fr::N], d1r2l[Fr::N];
    r.to_canonical_limbs(rl); d1.to_canonical_limbs(d1l); d2.to_canonical_limbs(d2l);
    (r * r).to_canonical_limbs(r2l); (d1 * (r + r)).to_canonical_limbs(d1r2l);
    // A
    Jac<Fq> gz = Jac<Fq>::from_affine(pk.g_gamma_z);
    Jac<Fq> gA = gz.mul_limbs(rl, Fr::N).add_mixed(pk.a_query[0]).add(gz.mul_limbs(d1l, Fr::N));
    gA = gA.add(msm<Fq, Fr::N>(pk.a_query.data() + 1, in_s, l - 1, NB, th)).add(msm<Fq, Fr::N>(pk.a_query.data() + l, aux_s, M - l, NB, th));
    double t1 = now();
    // B
    Jac<Fq2> hz = Jac<Fq2>::from_affine(pk.h_gamma_z);
    Jac<Fq2> gB = hz.mul_limbs(rl, Fr::N).add_mixed(pk.b_query[0]).add(hz.mul_limbs(d1l, Fr::N));
    gB = gB.add(msm<Fq2, Fr::N>(pk.b_query.data() + 1, in_s, l - 1, NB, th)).add(msm<Fq2, Fr::N>(pk.b_query.data() + l, aux_s, M - l, NB, th));
    double t2 = now();
    // C
    Jac<Fq> c1 = msm<Fq, Fr::N>(pk.c_query_1.data(), aux_s, M - l, NB, th);
    Jac<Fq> c2 = msm<Fq, Fr::N>(pk.c_query_2.data() + 1, in_s, l - 1, NB, th).add(msm<Fq, Fr::N>(pk.c_query_2.data() + l, aux_s, M - l, NB, th));
    double t3 = now();
    Jac<Fq> gacc = msm<Fq, Fr::N>(pk.g_gamma2_z_t.data(), hr.data(), l, NB, th)
                       .add(msm<Fq, Fr::N>(pk.g_gamma2_z_t.data() + l, hr.data() + l * Fr::N, D + 1 - l, NB, th));
    double t4 = now();
    Jac<Fq> z2 = Jac<Fq>::from_affine(pk.g_gamma2_z2), abz = Jac<Fq>::from_affine(pk.g_ab_gamma_z);
    Jac<Fq> gC = c1.add(z2.mul_limbs(r2l, Fr::N)).add(abz.mul_limbs(rl, Fr::N)).add(abz.mul_limbs(d1l, Fr::N))
                     .add(Jac<Fq>::from_affine(pk.c_query_2[0]).mul_limbs(rl, Fr::N)).add(z2.mul_limbs(d1r2l, Fr::N)).add(c2.mul_limbs(rl, Fr::N))
                     .add(Jac<Fq>::from_affine(pk.g_gamma2_z_t[0]).mul_limbs(d2l, Fr::N)).add(gacc);
    out.a = gA.to_affine(); out.b = gB.to_affine(); out.c = gC.to_affine();
    if (tm) { tm->matvec = 0; tm->fft = t0 - t00; tm->msm_a = t1 - t0; tm->msm_b2 = t2 - t1; tm->msm_l = t3 - t2; tm->msm_h = t4 - t3; tm->msm_b1 = 0; tm->total = now() - t00; }
    return 0;
}

// closed form (oracle/gm17.py: trapdoor_scalars)
template <class C>
static void trapdoor_gm17(const Circuit<C>& cs, const ToxicGm17<C>& tx, const std::vector<typename C::Fr>& z, const typename C::Fr& d1,
                          const typename C::Fr& r, ProofT<C>& out, int th) {
    typedef typename C::Fr Fr;
    auto q = sap_at_t(cs, tx.t);
    std::vector<Fr> ext, h;
    const SapShape sh = sap_shape(cs);
    ext.assign(z.begin(), z.end());
    ext.resize(sh.M, Fr::zero());
    const u64 n = cs.n, l = cs.l, m = cs.m();
    parallel_for(th, n, [&](size_t s, size_t e) {
        for (size_t i = s; i < e; ++i) { Fr d = cs.row_dot(cs.A, i, z) - cs.row_dot(cs.B, i, z); ext[m + i] = d * d; }
    });
    for (u64 i = 1; i < l; ++i) { Fr d = z[i] - Fr::one(); ext[m + n - 1 + i] = d * d; }
    Fr U = Fr::zero(), W = Fr::zero(), Ua = Fr::zero(), Wa = Fr::zero();
    for (u64 i = 0; i < sh.M; ++i) {
        Fr ua = ext[i] * q.a[i], wc = ext[i] * q.c[i];
        U = U + ua; W = W + wc;
        if (i >= l) { Ua = Ua + ua; Wa = Wa + wc; }
    }
    const Fr g = tx.gamma, ab = tx.alpha + tx.beta, rho = r + d1, zt = q.zt;
    Fr la = g * (U + rho * zt);
    Fr lc = g * g * Wa + ab * g * Ua + g * g * (U * U - W) + g * g * zt * ((rho + rho) * U + rho * rho * zt) + ab * g * rho * zt;
    u64 k[Fr::N];
    la.to_canonical_limbs(k);
    out.a = Jac<typename C::Fq>::from_affine(C::g1()).mul_limbs(k, Fr::N).to_affine();
    out.b = Jac<typename C::Fq2>::from_affine(C::g2()).mul_limbs(k, Fr::N).to_affine();
    lc.to_canonical_limbs(k);
    out.c = Jac<typename C::Fq>::from_affine(C::g1()).mul_limbs(k, Fr::N).to_affine();
}
