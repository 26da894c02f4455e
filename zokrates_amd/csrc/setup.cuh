// setup.cuh — Groth16 key generation on the device ("next" row N3 of SURVEY.md §8f).
//
// Replaces `Groth16::<E>::circuit_specific_setup(computation, rng)` at
// /root/reference/zokrates_ark/src/groth16.rs:95 ([UPSTREAM] ark_groth16::generate_random_parameters,
// SURVEY.md App. A.6) with the randomness made explicit (toxic waste and the two group generators are
// arguments), and writes the key in ark's `serialize_unchecked` layout (App. B.3) — the bytes
// `zkhip_pk_load_g16` and the reference's own `generate-proof` consume.
//
//   u_k    = Z(tau)/N * w^k / (tau - w^k)                       Lagrange basis at tau      (device)
//   a_i    = sum_k A[k][i] u_k  (+ u_{n+i} for i < l), b_i, c_i  column sums                (host, one thread per matrix)
//   a_query[i] = a_i G1, b_g1_query[i] = b_i G1, b_g2_query[i] = b_i G2                     (device, fixed-base)
//   gamma_abc[i] = (beta a_i + alpha b_i + c_i)/gamma G1 (i < l), l_query likewise with /delta
//   h_query[i] = tau^i Z(tau)/delta G1, i < N-1
#pragma once
#include <thread>

namespace zk {

// u[k] = scale * w^k / (tau - w^k)
template <class F>
__global__ void k_lagrange_at(F* __restrict__ u, F tau, F omega, F scale, u64 n) {
    u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    F w = fe_pow_u64(omega, k);
    u[k] = fe_mul(fe_mul(scale, w), fe_inv(fe_sub(tau, w)));
}
// out[i] = (beta a_i + alpha b_i + c_i) * (i < l ? ginv : dinv)
template <class F>
__global__ void k_lc_coeff(const F* __restrict__ a, const F* __restrict__ b, const F* __restrict__ c, F alpha, F beta, F ginv, F dinv, u64 l,
                           F* __restrict__ out, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F t = fe_add(fe_add(fe_mul(beta, a[i]), fe_mul(alpha, b[i])), c[i]);
    out[i] = fe_mul(t, i < l ? ginv : dinv);
}
template <class C>
struct Setup {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    static constexpr int FQB = Fq::BYTES;
    static constexpr int NWIN = 32;   // 8-bit windows over a 256-bit scalar

    static Fr read_fr(const uint8_t* b, bool nonzero, const char* what) {
        Fr c = fe_from_bytes_canon<Fr>(b);
        require(canon_lt_mod(c), ZKHIP_ERR_BAD_ARG, "toxic waste element is not a canonical field element");
        require(!nonzero || !c.is_zero(), ZKHIP_ERR_BAD_ARG, what);
        return fe_to_mont(c);
    }

    template <class F, int NC>
    static Aff<F> read_generator(const uint8_t* bytes, const u32* std_gen) {
        Aff<F> g;
        if (bytes) {
            uint8_t dec[4 * FQB];
            require(!(bytes[FQB * NC - 1] & 0x40), ZKHIP_ERR_BAD_ARG, "generator must not be the point at infinity");
            decode_point<FQB, NC>(bytes, dec);
            memcpy(&g, dec, sizeof(g));
        } else {
            memcpy(&g, std_gen, sizeof(g));
        }
        return PkLoader<C>::to_mont_point(g);
    }

    // table of d * 2^(8j) * G on the device
    template <class F>
    static void build_table(zkhip_ctx* ctx, const Aff<F>& g, DBuf& tbl) {
        std::vector<Aff<F>> pj(NWIN);
        Xyzz<F> cur = Xyzz<F>::from_affine(g);
        for (int j = 0; j < NWIN; ++j) {
            pj[j] = xyzz_to_affine(cur);
            for (int b = 0; b < 8; ++b) cur = xyzz_dbl(cur);
        }
        fixed_base_table<F>(ctx, pj.data(), NWIN, tbl);
    }

    // out (host, ark encoding) <- scalars[i] * G for i < count; d_scalars canonical on the device
    template <class F, int NC>
    static void mul_and_write(zkhip_ctx* ctx, const DBuf& tbl, const Fr* d_scalars, u64 count, uint8_t* out) {
        if (!count) return;
        Stream s = ctx->stream;
        constexpr int PB = FQB * NC;
        DBuf d_pts;
        d_pts.ensure(count * sizeof(Aff<F>));
        fixed_base_mul<F>(ctx, tbl, NWIN, (const u32*)d_scalars, count, ptr<Aff<F>>(d_pts));
        ZK_LAUNCH((k_from_mont<Fq>), dim3(blocks_for(count * NC, 256)), dim3(256), 0, s, ptr<Fq>(d_pts), ptr<Fq>(d_pts), count * NC);
        dev_d2h(out, d_pts.p, count * PB, s);
        stream_sync(s);
        for (u64 i = 0; i < count; ++i) {   // infinity: ark writes x = 0, y = 1, flag bit 6 of the last byte
            uint8_t* p = out + i * PB;
            bool zero = true;
            for (int b = 0; b < PB && zero; ++b) zero = p[b] == 0;
            if (zero) { p[PB / 2] = 1; p[PB - 1] |= 0x40; }
        }
    }
    static uint8_t* put_len(uint8_t* p, u64 n) { memcpy(p, &n, 8); return p + 8; }

    static void run(zkhip_ctx* ctx, const zkhip_r1cs* cs, const uint8_t* toxic, const uint8_t* g1b, const uint8_t* g2b, uint8_t* out, u64 cap) {
        constexpr int G1B = 2 * FQB, G2B = 4 * FQB;
        const u64 n = cs->n, l = cs->l, w = cs->w, m = l + w, N = cs->N;
        const u64 need = (u64)G1B + 3 * G2B + 8 + l * G1B + 2 * G1B + 8 + m * G1B + 8 + m * G1B + 8 + m * G2B + 8 + (N - 1) * G1B + 8 + w * G1B;
        require(cap >= need, ZKHIP_ERR_BAD_ARG, "output buffer too small (see zkhip_setup_g16_size)");
        const Fr alpha = read_fr(toxic, false, ""), beta = read_fr(toxic + 32, false, "");
        const Fr gamma = read_fr(toxic + 64, true, "gamma must be non-zero"), delta = read_fr(toxic + 96, true, "delta must be non-zero");
        const Fr tau = read_fr(toxic + 128, false, "");
        NttPlan<C>* pl = get_plan<C>(ctx, cs->logN);
        const Fr zt = fe_sub(fe_pow_u64(tau, N), Fr::one());
        require(!zt.is_zero(), ZKHIP_ERR_BAD_ARG, "tau lies in the evaluation domain");
        const Fr ginv = fe_inv(gamma), dinv = fe_inv(delta);
        Stream s = ctx->stream;
        const unsigned T = 256;

        // ---- Lagrange basis at tau, then the per-variable column sums on the host
        DBuf d_u;
        d_u.ensure(N * sizeof(Fr));
        ZK_LAUNCH((k_lagrange_at<Fr>), dim3(blocks_for(N, T)), dim3(T), 0, s, ptr<Fr>(d_u), tau, pl->omega, fe_mul(zt, pl->n_inv), N);
        std::vector<Fr> u(N);
        dev_d2h(u.data(), d_u.p, N * sizeof(Fr), s);
        stream_sync(s);
        std::vector<Fr> col[3];
        HostCsr hm;
        hm.fetch(ctx, cs);
        {
            HostThreads th;
            for (int k = 0; k < 3; ++k)
                th.run([&, k] {
                    std::vector<Fr>& acc = col[k];
                    acc.assign(m, Fr::zero());
                    const u64* rp = hm.rp[k].data();
                    const u32* ci = hm.col[k].data();
                    const uint8_t* va = hm.val[k].data();      // Montgomery form, as resident
                    for (u64 i = 0; i < n; ++i)
                        for (u64 q = rp[i]; q < rp[i + 1]; ++q) {
                            Fr v = fe_from_bytes_canon<Fr>(va + q * 32);
                            acc[ci[q]] = fe_add(acc[ci[q]], fe_mul(v, u[i]));
                        }
                    if (k == 0)
                        for (u64 j = 0; j < l; ++j) acc[j] = fe_add(acc[j], u[n + j]);
                });
            th.join();
        }
        DBuf d_a, d_b, d_c, d_lc, d_h;
        d_a.ensure(m * sizeof(Fr)); d_b.ensure(m * sizeof(Fr)); d_c.ensure(m * sizeof(Fr)); d_lc.ensure(m * sizeof(Fr));
        d_h.ensure(N * sizeof(Fr));
        dev_h2d(d_a.p, col[0].data(), m * sizeof(Fr), s);
        dev_h2d(d_b.p, col[1].data(), m * sizeof(Fr), s);
        dev_h2d(d_c.p, col[2].data(), m * sizeof(Fr), s);
        ZK_LAUNCH((k_lc_coeff<Fr>), dim3(blocks_for(m, T)), dim3(T), 0, s, ptr<Fr>(d_a), ptr<Fr>(d_b), ptr<Fr>(d_c), alpha, beta, ginv, dinv, l,
                  ptr<Fr>(d_lc), m);
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(N, T)), dim3(T), 0, s, ptr<Fr>(d_h), tau, fe_mul(zt, dinv), N, 0u, 0u, 0);
        // fixed-base kernels read canonical integers
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(m, T)), dim3(T), 0, s, ptr<Fr>(d_a), ptr<Fr>(d_a), m);
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(m, T)), dim3(T), 0, s, ptr<Fr>(d_b), ptr<Fr>(d_b), m);
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(m, T)), dim3(T), 0, s, ptr<Fr>(d_lc), ptr<Fr>(d_lc), m);
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(N, T)), dim3(T), 0, s, ptr<Fr>(d_h), ptr<Fr>(d_h), N);
        // the six single points: alpha, beta, delta in G1; beta, gamma, delta in G2
        Fr singles[6] = {fe_from_mont(alpha), fe_from_mont(beta), fe_from_mont(delta), fe_from_mont(beta), fe_from_mont(gamma), fe_from_mont(delta)};
        DBuf d_single;
        d_single.ensure(sizeof(singles));
        dev_h2d(d_single.p, singles, sizeof(singles), s);
        stream_sync(s);

        DBuf tbl1, tbl2;
        build_table<Fq>(ctx, read_generator<Fq, 2>(g1b, C::g1_gen()), tbl1);
        build_table<Fq2>(ctx, read_generator<Fq2, 4>(g2b, C::g2_gen()), tbl2);

        uint8_t s1[3 * G1B], s2[3 * G2B];
        mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_single), 3, s1);
        mul_and_write<Fq2, 4>(ctx, tbl2, ptr<Fr>(d_single) + 3, 3, s2);
        uint8_t* p = out;
        memcpy(p, s1, G1B); p += G1B;                         // alpha_g1
        memcpy(p, s2, 3 * G2B); p += 3 * G2B;                 // beta_g2, gamma_g2, delta_g2
        p = put_len(p, l);
        mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_lc), l, p); p += l * G1B;            // gamma_abc_g1
        memcpy(p, s1 + G1B, 2 * G1B); p += 2 * G1B;           // beta_g1, delta_g1
        p = put_len(p, m);
        mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_a), m, p); p += m * G1B;             // a_query
        p = put_len(p, m);
        mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_b), m, p); p += m * G1B;             // b_g1_query
        p = put_len(p, m);
        mul_and_write<Fq2, 4>(ctx, tbl2, ptr<Fr>(d_b), m, p); p += m * G2B;            // b_g2_query
        p = put_len(p, N - 1);
        mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_h), N - 1, p); p += (N - 1) * G1B;   // h_query
        p = put_len(p, w);
        mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_lc) + l, w, p); p += w * G1B;        // l_query
        require((u64)(p - out) == need, ZKHIP_ERR_DEVICE, "internal: key size mismatch");
    }
};

}  // namespace zk
