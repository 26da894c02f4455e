// gm17.cuh — the GM17 proving scheme on the same device machinery (config 5 of BASELINE.json: "second proof system
// behind same Backend trait").
//
// Replaces `<Ark as Backend<T, GM17>>::generate_proof` (/root/reference/zokrates_ark/src/gm17.rs:43-78) from the point
// where the reference hands over to ark — `ProvingKey::deserialize_unchecked` (:60-62) and `GM17::prove` (:64),
// [UPSTREAM] ark-gm17 0.3.0 `create_proof` / `R1CStoSAP::witness_map` — and `NonUniversalBackend<T, GM17>::setup`
// (:19-41, `circuit_specific_setup` -> [UPSTREAM] `generate_parameters` / `R1CStoSAP::instance_map_with_evaluation`).
// Restatement of the algorithm: oracle/gm17.py.
//
// R1CS -> SAP (squares only).  Variables [1, x_1..x_{l-1}, aux_0..aux_{w-1}, e_0..e_{n-1}, f_1..f_{l-1}], M of them;
// rows (D0 = 2n + 2(l-1) + 1, padded to the radix-2 domain D):
//   2k: (A_k+B_k)^2 = 4C_k + e_k     2k+1: (A_k-B_k)^2 = e_k     2n: 1 = 1
//   2n+2i-1: (x_i+1)^2 = 4x_i + f_i  2n+2i: (x_i-1)^2 = f_i
//
// Restructuring that keeps the proof bit-identical (for a fixed key and assignment the three proof points depend on
// d1, d2, r only through rho = r + d1 — d2 cancels — and are unique group elements):
//   A = a_query . ext + rho g_gamma_z                        lane 0   bases [a_query, g_gamma_z, inf]
//   B = b_query . ext + rho h_gamma_z                (G2)    lane 3   bases [b_query, h_gamma_z, inf]
//   C = c_query_1 . ext[l..] + rho g_ab_gamma_z              lane 2   bases [inf x l, c_query_1, g_ab_gamma_z, inf]
//     + rho (c_query_2 . ext)                                lane 1   bases [c_query_2, inf, inf]   (the key carries the 2)
//     + rho^2 g_gamma2_z2                                    host
//     + g_gamma2_z_t . h0,  h0 = (U^2 - W)/Z                 lane 4   bases sigma-permuted like h_query
// with scalars [ext_0..ext_{M-1}, rho, 0]: the four MSMs over the extended assignment share one digit/sort pass, as in
// the Groth16 prover.  ark folds 2 d1 U + d1^2 Z - d2 into the quotient instead; the sums are the same group elements.
#pragma once

namespace zk {

// SAP evaluation vectors and the extension of the assignment from the R1CS row products a = A z, b = B z, c = C z
// (k_matvec), one work-item per R1CS row / per extra row:
//   i < n      : e = (a_i - b_i)^2
//   i == n     : the constant row
//   n < i < n+l: public input x = z[i-n], f = (x-1)^2
// ra/rb/rc/z and sa/sc: R'-form (kernels_ntt.cuh), natural order (the tail [2n+2l-1, D) is zeroed by the caller);
// ext: canonical integers
template <class P>
__global__ void k_sap_rows(const Fe<P>* __restrict__ ra, const Fe<P>* __restrict__ rb, const Fe<P>* __restrict__ rc, const Fe<P>* __restrict__ z,
                           Fe<P>* __restrict__ sa, Fe<P>* __restrict__ sc, Fe<P>* __restrict__ ext, u64 n, u64 l, u64 m) {
    ZK_PRIO_HIGH();
    typedef Fe<P> F;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + l) return;
    // (sc == nullptr: a key bound to the system carries W's share in its bases — rc is not read, sc not written)
    if (i < n) {
        const F a = ra[i], b = rb[i];
        const F d = fe_sub(a, b);
        const F e = rp_sqr(d);
        sa[2 * i] = fe_add(a, b);
        sa[2 * i + 1] = d;
        if (sc) {
            sc[2 * i] = fe_add(fe_dbl(fe_dbl(rc[i])), e);
            sc[2 * i + 1] = e;
        }
        ext[m + i] = rp_to_plain(e);
    } else if (i == n) {
        sa[2 * n] = rp_one<P>();
        if (sc) sc[2 * n] = rp_one<P>();
    } else {
        const u64 j = i - n;   // 1 <= j < l
        const F x = z[j], one = rp_one<P>();
        const F d = fe_sub(x, one);
        const F f = rp_sqr(d);
        sa[2 * n + 2 * j - 1] = fe_add(x, one);
        sa[2 * n + 2 * j] = d;
        if (sc) {
            sc[2 * n + 2 * j - 1] = fe_add(fe_dbl(fe_dbl(x)), f);
            sc[2 * n + 2 * j] = f;
        }
        ext[m + n - 1 + j] = rp_to_plain(f);
    }
}
// quotient evaluations on the coset, square part: out = a^2 * zinv     (R'-form operand < 2^256, canonical R'-form out); the
// `- c` is subtracted in coefficient form by the last transform (see k_quotient)
template <class P>
__global__ void k_sap_quotient(const Fe<P>* __restrict__ a, Fe<P> zinv, Fe<P>* __restrict__ out, u64 n) {
    ZK_PRIO_HIGH();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fu<P> aa = fu_sqr_inl(fu_unpack<P>(a[i].v));
    out[i] = rp_canon(fu_mul_inl(aa, fu_unpack<P>(zinv.v)));
}
// setup: the per-variable key scalars (canonical) from u_i(t) = a[i], w_i(t) = c[i]
//   aq = gamma a;  c1 = gamma^2 c + (alpha+beta) gamma a;  c2 = 2 gamma^2 Z a;  vq = gamma c + (alpha+beta) a
template <class F>
__global__ void k_gm17_coeffs(const F* __restrict__ a, const F* __restrict__ c, F gamma, F ab, F two_g2z, F* __restrict__ aq, F* __restrict__ c1,
                              F* __restrict__ c2, F* __restrict__ vq, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F v = fe_add(fe_mul(gamma, c[i]), fe_mul(ab, a[i]));
    aq[i] = fe_from_mont(fe_mul(gamma, a[i]));
    c1[i] = fe_from_mont(fe_mul(gamma, v));
    c2[i] = fe_from_mont(fe_mul(two_g2z, a[i]));
    vq[i] = fe_from_mont(v);
}

struct SapShape {
    u64 M, D0, D;
    int logD;
};
static inline SapShape sap_shape(u64 n, u64 l, u64 w) {
    SapShape s;
    s.M = 1 + 2 * (l - 1) + w + n;
    s.D0 = 2 * n + 2 * (l - 1) + 1;
    s.logD = ilog2_ceil(s.D0);
    s.D = (u64)1 << s.logD;
    return s;
}

template <class C>
struct Gm17 {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    typedef PkLoader<C> L;
    typedef Prover<C> P;
    static constexpr int FQB = Fq::BYTES;
    static constexpr int G1B = 2 * FQB, G2B = 4 * FQB;

    // ------------------------------------------------------------ key load
    // ark `serialize_unchecked` of ark_gm17::ProvingKey: vk{h_g2, g_alpha_g1, h_beta_g2, g_gamma_g1, h_gamma_g2, query[]},
    // a_query[], b_query[] (G2), c_query_1[], c_query_2[], g_gamma_z, h_gamma_z (G2), g_ab_gamma_z, g_gamma2_z2,
    // g_gamma2_z_t[]
    static void load(zkhip_ctx* ctx, const uint8_t* bytes, size_t len, zkhip_pk* pk) {
        typename L::Rd rd{bytes, bytes + len};
        rd.take(G2B); rd.take(G1B); rd.take(G2B); rd.take(G1B); rd.take(G2B);   // vk points (verifier only)
        const u64 l = rd.len(G1B);
        rd.take(l * G1B);                                                        // vk.query (verifier only)
        const u64 M = rd.len(G1B);
        const uint8_t* a_q = rd.take(M * G1B);
        const u64 Mb = rd.len(G2B);
        const uint8_t* b_q = rd.take(Mb * G2B);
        const u64 n1 = rd.len(G1B);
        const uint8_t* c1_q = rd.take(n1 * G1B);
        const u64 Mc2 = rd.len(G1B);
        const uint8_t* c2_q = rd.take(Mc2 * G1B);
        const uint8_t* g_gamma_z = rd.take(G1B);
        const uint8_t* h_gamma_z = rd.take(G2B);
        const uint8_t* g_ab_gamma_z = rd.take(G1B);
        const uint8_t* g_gamma2_z2 = rd.take(G1B);
        const u64 tl = rd.len(G1B);
        const uint8_t* t_q = rd.take(tl * G1B);
        require(rd.p == rd.e, ZKHIP_ERR_PARSE, "trailing bytes after proving key");
        require(l >= 1 && M >= l && Mb == M && Mc2 == M && n1 == M - l, ZKHIP_ERR_PARSE, "inconsistent query lengths in GM17 proving key");
        require(tl >= 2 && ((tl - 1) & (tl - 2)) == 0, ZKHIP_ERR_PARSE, "g_gamma2_z_t length - 1 is not a power of two");
        require(M + 2 < ((u64)1 << 31), ZKHIP_ERR_BAD_ARG, "too many variables");
        const u64 D = tl - 1;
        pk->scheme = 1;
        pk->m = M; pk->w = n1; pk->l = l; pk->hlen = tl; pk->N = D; pk->logN = ilog2_floor(D);
        NttPlan<C>* plan = get_plan<C>(ctx, pk->logN);
        pk->ntt_log1 = plan->split();
        pk->g_gamma2_z2_canon.assign(g_gamma2_z2, g_gamma2_z2 + G1B);

        const u64 me = M + 2;   // extended by the (., rho) pair and one unused slot (same shape as the Groth16 key)
        L::template upload_decoded<2>(ctx, pk->a_ext, me, a_q, M, 0, g_gamma_z, M);
        L::template upload_decoded<2>(ctx, pk->b1_ext, me, c2_q, M, 0, nullptr, 0);
        L::template upload_decoded<2>(ctx, pk->l_ext, me, c1_q, n1, l, g_ab_gamma_z, M);
        L::template upload_decoded<4>(ctx, pk->b2_ext, me, b_q, M, 0, h_gamma_z, M);
        // g_gamma2_z_t[0..D), permuted into the sigma order the NTT pipeline leaves the quotient in (entry D pairs with
        // the d1^2 coefficient of ark's h, which the restructured prover does not produce)
        DBuf t_nat;
        L::template upload_decoded<2>(ctx, t_nat, D, t_q, D, 0, nullptr, 0);
        pk->h_sigma.ensure(D * G1B);
        ZK_LAUNCH((k_sigma_gather_points<Aff<Fq>>), dim3(blocks_for(D, 256)), dim3(256), 0, ctx->stream, ptr<Aff<Fq>>(t_nat),
                  ptr<Aff<Fq>>(pk->h_sigma), D, D, plan->N1, plan->N2, plan->N3);
        stream_sync(ctx->stream);
        t_nat.release();
        L::finish_tables(ctx, pk, me, D);
    }

    static void check_match(const zkhip_pk* pk, const zkhip_r1cs* cs) {
        require(pk->curve == C::ID && cs->curve == C::ID, ZKHIP_ERR_BAD_ARG, "curve mismatch between key and constraint system");
        require(pk->scheme == 1, ZKHIP_ERR_BAD_ARG, "this is a Groth16 proving key: use zkhip_prove_g16");
        const SapShape sh = sap_shape(cs->n, cs->l, cs->w);
        require(pk->m == sh.M && pk->l == cs->l && pk->N == sh.D, ZKHIP_ERR_BAD_ARG,
                "GM17 proving key does not match the constraint system (SAP variables, instance size or domain)");
    }

    // ------------------------------------------------------------ prover
    static Fr add_mod(const Fr& a, const Fr& b) {   // canonical integers
        return fe_from_mont(fe_add(fe_to_mont(a), fe_to_mont(b)));
    }
    // every kernel and copy of one proof, no host synchronisation (the slot discipline of Prover<C>::enqueue)
    static void enqueue(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* src_dev,
                        const uint8_t* d1, const uint8_t* r) {
        check_match(pk, cs);
        require(!sl.busy, ZKHIP_ERR_DEVICE, "internal: proof slot still in flight");
        make_pipe_streams(ctx);
        slot_init(ctx, sl);
        const u64 m = cs->l + cs->w, n = cs->n, l = cs->l, M = pk->m, D = pk->N;
        Fr dd = fe_from_bytes_canon<Fr>(d1), rr = fe_from_bytes_canon<Fr>(r);
        require(canon_lt_mod(dd) && canon_lt_mod(rr), ZKHIP_ERR_BAD_ARG, "d1 or r not a canonical field element");
        const Fr rho = add_mod(dd, rr);
        const bool bound = pk->bound_uid != 0 && pk->bound_uid == cs->uid;   // (zkhip_pk_bind_r1cs: W's share and the last transform live in the bases)
        NttPlan<C>* pl = get_plan<C>(ctx, pk->logN);
        require(pl->split() == pk->ntt_log1, ZKHIP_ERR_BAD_ARG, "the key's quotient bases were ordered for another NTT split (NTT_SINGLE_MAX_LOG changed): reload the key");
        sl.t_start = std::chrono::steady_clock::now();
        memcpy(sl.r, rho.v, 32);
        memset(sl.s, 0, 32);
        ctx->cur = &sl;
        Stream st = ctx->stream;
        ctx->ws = st;
        sl.scalars.ensure((M + 2) * 32);
        if (z_host) {
            Fr z0 = fe_from_bytes_canon<Fr>(z_host);
            Fr one = Fr::zero(); one.v[0] = 1;
            require(z0.equals(one), ZKHIP_ERR_BAD_ARG, "z[0] must be 1 (ark instance variable 0 is the constant ONE)");
            dev_h2d(sl.scalars.p, z_host, m * 32, st);
            sl.zflag.ensure(4);
            dev_memset(sl.zflag.p, 0, 4, st);
            ZK_LAUNCH((k_check_canonical<Fr>), dim3(blocks_for(m, 256)), dim3(256), 0, st, ptr<Fr>(sl.scalars), m, ptr<u32>(sl.zflag));
        } else {
            dev_d2d(sl.scalars.p, src_dev, m * 32, st);
            sl.zflag.ensure(4);
            dev_memset(sl.zflag.p, 0, 4, st);
        }
        uint8_t* d_scalars = (uint8_t*)sl.scalars.p;
        sl.zmont.ensure(m * 32);
        dev_h2d(d_scalars + M * 32, sl.r, 32, st);
        dev_memset(d_scalars + (M + 1) * 32, 0, 32, st);
        ZK_LAUNCH((k_mul_const<Fr>), dim3(blocks_for(m, 256)), dim3(256), 0, st, (const Fr*)d_scalars, ptr<Fr>(sl.zmont), m, pl->k_to_rp);   // R'-form
        event_record(sl.ev[0], st);

        // ---- SAP rows + the extension of the assignment (K1')
        sl.va.ensure(2 * D * sizeof(Fr));      // the two evaluation vectors back to back: they share every NTT launch
        Fr *sa = ptr<Fr>(sl.va), *sc = sa + D;
        const u64 D0 = 2 * n + 2 * (l - 1) + 1;
        if (D0 < D) {
            dev_memset(sa + D0, 0, (D - D0) * sizeof(Fr), st);
            if (!bound) dev_memset(sc + D0, 0, (D - D0) * sizeof(Fr), st);
        }
        sl.vc.ensure(3 * std::max<u64>(n, 1) * sizeof(Fr));      // the three row-product vectors
        Fr *ra = ptr<Fr>(sl.vc), *rb = ra + n, *rc = rb + n;
        if (n) P::matvec(ctx, cs, ptr<Fr>(sl.zmont), ra, rb, rc, n, 0, n, bound ? 2 : 3);
        ZK_LAUNCH((k_sap_rows<typename Fr::Params>), dim3(blocks_for(n + l, 256)), dim3(256), 0, st, ra, rb, rc, ptr<Fr>(sl.zmont), sa, bound ? (Fr*)nullptr : sc,
                  (Fr*)d_scalars, n, l, m);
        event_record(sl.ev[1], st);

        // ---- the four MSMs over S = [ext_0..ext_{M-1}, rho, 0] share one digit/sort pass
        const MsmShape shz = msm_shape(ctx, pk->z_n, Fr::Params::BITS, true, pk->c_z, pk->s_z);
        const MsmShape shh = msm_shape(ctx, pk->h_n, Fr::Params::BITS, true, pk->c_h, pk->s_h);
        require(shz.c == pk->c_z && shh.c == pk->c_h && (int)shz.sets == pk->s_z && (int)shh.sets == pk->s_h, ZKHIP_ERR_BAD_ARG,
                "the key's tables were built for a window width this build cannot sort: reload the key");
        const int Wmax = (int)std::max(shz.nsums(), shh.nsums());
        sl.ws1.ensure((size_t)4 * Wmax * sizeof(Xyzz<Fq>));
        sl.ws2.ensure((size_t)Wmax * sizeof(Xyzz<Fq2>));
        Xyzz<Fq>* ws1 = ptr<Xyzz<Fq>>(sl.ws1);
        P::host_sums(sl, Wmax);
        Xyzz<Fq>* hs1 = (Xyzz<Fq>*)sl.h_ws;                    // the host mirrors of ws1 / ws2: every lane copies its own sums out
        Xyzz<Fq2>* hs2 = (Xyzz<Fq2>*)((uint8_t*)sl.h_ws + (size_t)4 * Wmax * sizeof(Xyzz<Fq>));
        // (a sharded key covers only its index range of the bases and pairs them with the same range of the scalars)
        // (the G1 lanes wait for the transforms, as in Prover::enqueue: the G2 lane starts at once)
        const int gate = z_gate(ctx);
        const MsmSort& sort_b = (pk->thin_mask & 8) ? sl.sorts[2] : sl.sorts[0];   // the list b2_ext pairs with (zkhip_pk::thin_mask)
        const bool inf_b2 = (pk->thin_mask & 8) ? pk->inf_many_thin[3] : pk->inf_many[3];
        if (pk->z_n) {
            msm_prepare(ctx, st, sl.sorts[0], (const u32*)d_scalars + pk->z_lo * 8, shz, pk->z_n);
            if (pk->thin_mask) msm_prepare(ctx, st, sl.sorts[2], (const u32*)d_scalars + pk->z_lo * 8, shz, pk->z_n, ptr<u32>(pk->thin_keep), &sl.sorts[0]);
            if (gate < 2)
                msm_run<Fq2>(ctx, sl.lanes[3], sort_b, pk->b2_ext.p, with_inf(shz, inf_b2), ptr<Xyzz<Fq2>>(sl.ws2), sl.acc_b[4], sl.acc_e[4], nullptr, hs2);   // longest first
        }

        // ---- quotient h0 = (U^2 - W)/Z: iNTT + coset NTT of U, pointwise square, coset iNTT minus W's coefficients / Z (sigma order, canonical)
        // the transforms and the h-sort run on the NTT stream (the SAP rows above feed the z-sort and stay on the main one)
        Stream wn = ctx->serial ? st : ctx_ntt_stream(ctx);
        stream_wait_event(wn, sl.ev[1]);
        ctx->ws = wn;
        event_record(sl.ntt_b, wn);
        if (bound) {
            // TWO transforms: U to its coefficients and on to the coset; U^2 / Z(g) there, as canonical integers in NATURAL order (a
            // plain-integer factor takes the R'-form product out of the Montgomery domain), pairs with G' = the coset inverse transform
            // applied to g_gamma2_z_t once (PkLoader::bind) — and W's share rides on the c_query_1 lane's bases
            ntt_kind_a<C>(ctx, pl, sa, true, ptr<Fr>(pl->s_coset), 1, 0);
            ntt_kind_b<C>(ctx, pl, sa, false, nullptr, 1, 0);
            ZK_LAUNCH((k_sap_quotient<typename Fr::Params>), dim3(blocks_for(D, 256)), dim3(256), 0, wn, sa, fe_from_mont(pl->zinv), sa, D);
        } else {
            // four transforms (ark-gm17's witness_map runs five): W only needs its coefficients, as c in the Groth16 prover
            ntt_kind_a<C>(ctx, pl, sa, true, ptr<Fr>(pl->s_coset), 1, 0);
            ntt_kind_a<C>(ctx, pl, sc, true, ptr<Fr>(pl->s_cexit), 1, 0, 1);
            ntt_kind_b<C>(ctx, pl, sa, false, nullptr, 1, 0);
            ZK_LAUNCH((k_sap_quotient<typename Fr::Params>), dim3(blocks_for(D, 256)), dim3(256), 0, wn, sa, pl->zinv_rp, sa, D);
            ntt_kind_a<C>(ctx, pl, sa, true, ptr<Fr>(pl->s_cosetinv_canon), 1, 0, 1, sc);
        }
        ctx->ws = ctx->stream;
        event_record(sl.ntt_e, wn);
        event_record(sl.ev[2], wn);

        if (pk->z_n) {
            const Event h_ready = gate ? sl.ev[2] : nullptr;
            if (gate >= 2)
                msm_run<Fq2>(ctx, sl.lanes[3], sort_b, pk->b2_ext.p, with_inf(shz, inf_b2), ptr<Xyzz<Fq2>>(sl.ws2), sl.acc_b[4], sl.acc_e[4], h_ready, hs2);
            P::run_z_g1(ctx, sl, pk, shz, ws1, Wmax, h_ready, bound, hs1);
        } else {
            P::empty_msm(ctx, sl, ws1, 3 * Wmax, ptr<Xyzz<Fq2>>(sl.ws2), Wmax, 0, 4);
        }

        // ---- G = MSM(g_gamma2_z_t, h0)   (a bound key: U^2 / Z(g) in natural order against G')
        if (pk->h_n) {
            msm_prepare(ctx, wn, sl.sorts[1], ptr<u32>(sl.va) + pk->h_lo * 8, shh, pk->h_n);
            msm_run<Fq>(ctx, sl.lanes[4], sl.sorts[1], bound ? pk->h_bound.p : pk->h_sigma.p, with_inf(shh, bound ? pk->inf_many_bound[1] : pk->inf_many[4]),
                        ws1 + 3 * Wmax, sl.acc_b[3], sl.acc_e[3], nullptr, hs1 + 3 * Wmax);
        } else {
            P::empty_msm(ctx, sl, ws1 + 3 * Wmax, Wmax, nullptr, 0, 4, 5);
        }
        P::copy_out(ctx, sl, Wmax);
    }

    // C = C1 + rho C2 + rho^2 g_gamma2_z2 + G
    static void finish(zkhip_ctx* ctx, ProofSlot& sl, const zkhip_pk* pk, uint8_t* out, zkhip_timings* tm) {
        require(pk->world == 1, ZKHIP_ERR_BAD_ARG, "this proving key is one shard of a multi-GPU key: use zkhip_prove_gm17_partial + zkhip_combine_gm17");
        const typename P::Sums g = P::collect(ctx, sl, pk);
        const auto t_fin = std::chrono::steady_clock::now();
        Fr rho;
        memcpy(rho.v, sl.r, 32);
        assemble(pk, g, rho, out);
        P::fill_timings(sl, tm, t_fin);
    }
    static void assemble(const zkhip_pk* pk, const typename P::Sums& g, const Fr& rho, uint8_t* out) {
        const Fr rho2 = fe_from_mont(fe_sqr(fe_to_mont(rho)));
        uint8_t dec[2 * FQB];
        decode_point<FQB, 2>(pk->g_gamma2_z2_canon.data(), dec);
        Aff<Fq> z2;
        memcpy(&z2, dec, sizeof(z2));
        z2 = L::to_mont_point(z2);
        Xyzz<Fq> t1, t2;
        {
            HostThreads th;
            th.run([&] { t1 = xyzz_mul_limbs(g.b1, rho.v, Fr::N); });
            t2 = xyzz_mul_limbs(Xyzz<Fq>::from_affine(z2), rho2.v, Fr::N);
        }
        Xyzz<Fq> gC = xyzz_add(g.l, t1);
        gC = xyzz_add(gC, t2);
        gC = xyzz_add(gC, g.h);
        Aff<Fq> pa = xyzz_to_affine(g.a), pc = xyzz_to_affine(gC);
        Aff<Fq2> pb = xyzz_to_affine(g.b2);
        memset(out, 0, 8 * FQB + 3);
        if (!g.a.is_inf()) { write_fe(pa.x, out); write_fe(pa.y, out + FQB); }
        if (!g.b2.is_inf()) {
            write_fe(pb.x.c0, out + 2 * FQB); write_fe(pb.x.c1, out + 3 * FQB);
            write_fe(pb.y.c0, out + 4 * FQB); write_fe(pb.y.c1, out + 5 * FQB);
        }
        if (!gC.is_inf()) { write_fe(pc.x, out + 6 * FQB); write_fe(pc.y, out + 7 * FQB); }
        out[8 * FQB] = g.a.is_inf(); out[8 * FQB + 1] = g.b2.is_inf(); out[8 * FQB + 2] = gC.is_inf();
    }
    // one rank's share of a proof (as Prover<C>::prove_partial): the five partial sums, canonical records
    static void prove_partial(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* z_dev, const uint8_t* rnd,
                              uint8_t* partial_out, zkhip_timings* tm) {
        check_d2(rnd);
        enqueue(ctx, ctx->slots[0], pk, cs, z_host, z_dev, rnd, rnd + 64);
        typename P::Sums g = P::collect(ctx, ctx->slots[0], pk);
        const auto t_fin = std::chrono::steady_clock::now();
        P::canonicalise(g);
        memcpy(partial_out, &g, sizeof(g));
        P::fill_timings(ctx->slots[0], tm, t_fin);
    }
    // (as Prover<C>::prove_device_sums: the share stays on the device for the RCCL exchange)
    static void prove_device_sums(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const uint8_t* rnd,
                                  const void** d_ws1, size_t* b1, const void** d_ws2, size_t* b2, zkhip_timings* tm) {
        check_d2(rnd);
        enqueue(ctx, ctx->slots[0], pk, cs, z_host, nullptr, rnd, rnd + 64);
        P::wait_device_sums(ctx, ctx->slots[0], pk, d_ws1, b1, d_ws2, b2);
        P::fill_timings(ctx->slots[0], tm, std::chrono::steady_clock::now());
    }
    static void combine(const zkhip_pk* pk, u32 count, const uint8_t* partials, const uint8_t* rnd, uint8_t* out) {
        check_d2(rnd);
        Fr dd = fe_from_bytes_canon<Fr>(rnd), rr = fe_from_bytes_canon<Fr>(rnd + 64);
        require(canon_lt_mod(dd) && canon_lt_mod(rr), ZKHIP_ERR_BAD_ARG, "d1 or r not a canonical field element");
        typename P::Sums t;
        t.a = t.b1 = t.l = t.h = Xyzz<Fq>::inf();
        t.b2 = Xyzz<Fq2>::inf();
        for (u32 i = 0; i < count; ++i) {
            typename P::Sums g;
            memcpy(&g, partials + (size_t)i * sizeof(g), sizeof(g));
            t.a = xyzz_add(t.a, g.a); t.b1 = xyzz_add(t.b1, g.b1); t.l = xyzz_add(t.l, g.l); t.h = xyzz_add(t.h, g.h);
            t.b2 = xyzz_add(t.b2, g.b2);
        }
        assemble(pk, t, add_mod(dd, rr), out);
    }

    // rnd = d1 | d2 | r (3 x 32 B): d2 is validated and otherwise unused — it cancels out of the proof
    static void check_d2(const uint8_t* rnd) {
        require(canon_lt_mod(fe_from_bytes_canon<Fr>(rnd + 32)), ZKHIP_ERR_BAD_ARG, "d2 not a canonical field element");
    }
    static void prove(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, const uint8_t* z_host, const void* z_dev, const uint8_t* rnd,
                      uint8_t* out, zkhip_timings* tm) {
        check_d2(rnd);
        enqueue(ctx, ctx->slots[0], pk, cs, z_host, z_dev, rnd, rnd + 64);
        finish(ctx, ctx->slots[0], pk, out, tm);
    }
    // `count` proofs, ZK_NSLOTS in flight (see Prover<C>::prove_batch); rnd: count x 96 B
    static void prove_batch(zkhip_ctx* ctx, const zkhip_pk* pk, const zkhip_r1cs* cs, u32 count, const uint8_t* z_host, void* const* z_dev,
                            const uint8_t* rnd, uint8_t* proofs_out, zkhip_timings* tm) {
        const size_t proof_bytes = 8 * FQB + 3;
        const u64 m = cs->l + cs->w;
        zkhip_timings acc, one;
        memset(&acc, 0, sizeof(acc));
        const auto t0 = std::chrono::steady_clock::now();
        try {
            const u32 NS = (u32)ctx->nslots;   // proofs in flight
            for (u32 i = 0; i < count + NS - 1; ++i) {
                if (i < count) {
                    check_d2(rnd + (size_t)i * 96);
                    enqueue(ctx, ctx->slots[i % NS], pk, cs, z_host ? z_host + (size_t)i * m * 32 : nullptr, z_host ? nullptr : z_dev[i],
                            rnd + (size_t)i * 96, rnd + (size_t)i * 96 + 64);
                }
                if (i >= NS - 1 && i - (NS - 1) < count) {
                    const u32 j = i - (NS - 1);
                    finish(ctx, ctx->slots[j % NS], pk, proofs_out + (size_t)j * proof_bytes, &one);
                    float* a = (float*)&acc; const float* b = (const float*)&one;
                    for (size_t k = 0; k < sizeof(acc) / sizeof(float); ++k) a[k] += b[k];
                }
            }
        } catch (...) {
            for (auto& sl : ctx->slots) sl.busy = false;
            dev_sync_all();
            throw;
        }
        if (tm) {
            *tm = acc;
            tm->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }

    // ------------------------------------------------------------ setup
    // toxic = alpha, beta, gamma, t (4 x 32 B); generators as in zkhip_setup_g16
    static u64 key_bytes(u64 n, u64 l, u64 w) {
        const SapShape sh = sap_shape(n, l, w);
        return (u64)3 * G2B + 2 * G1B + 8 + l * G1B + 8 + sh.M * G1B + 8 + sh.M * G2B + 8 + (sh.M - l) * G1B + 8 + sh.M * G1B + 3 * G1B + G2B + 8 +
               (sh.D + 1) * G1B;
    }
    static void setup(zkhip_ctx* ctx, const zkhip_r1cs* cs, const uint8_t* toxic, const uint8_t* g1b, const uint8_t* g2b, uint8_t* out, u64 cap) {
        typedef Setup<C> S;
        const u64 n = cs->n, l = cs->l, w = cs->w, m = l + w;
        const SapShape sh = sap_shape(n, l, w);
        const u64 M = sh.M, D = sh.D;
        const u64 need = key_bytes(n, l, w);
        require(cap >= need, ZKHIP_ERR_BAD_ARG, "output buffer too small (see zkhip_setup_gm17_size)");
        const Fr alpha = S::read_fr(toxic, false, ""), beta = S::read_fr(toxic + 32, false, "");
        const Fr gamma = S::read_fr(toxic + 64, true, "gamma must be non-zero");
        const Fr t = S::read_fr(toxic + 96, false, "");
        NttPlan<C>* pl = get_plan<C>(ctx, sh.logD);
        const Fr zt = fe_sub(fe_pow_u64(t, D), Fr::one());
        require(!zt.is_zero(), ZKHIP_ERR_BAD_ARG, "t lies in the evaluation domain");
        Stream s = ctx->stream;
        const unsigned T = 256;

        // ---- Lagrange basis at t, then u_i(t), w_i(t) per SAP variable on the host (instance_map_with_evaluation)
        DBuf d_u;
        d_u.ensure(D * sizeof(Fr));
        ZK_LAUNCH((k_lagrange_at<Fr>), dim3(blocks_for(D, T)), dim3(T), 0, s, ptr<Fr>(d_u), t, pl->omega, fe_mul(zt, pl->n_inv), D);
        std::vector<Fr> u(D);
        dev_d2h(u.data(), d_u.p, D * sizeof(Fr), s);
        stream_sync(s);
        std::vector<Fr> av(M, Fr::zero()), cv(M, Fr::zero());
        HostCsr hm;
        hm.fetch(ctx, cs);
        auto val_at = [&](int k, u64 q) { return fe_from_bytes_canon<Fr>(hm.val[k].data() + q * 32); };   // Montgomery form, as resident
        HostThreads ta;
        ta.run([&] {
            for (u64 i = 0; i < n; ++i) {
                const Fr u_add = fe_add(u[2 * i], u[2 * i + 1]), u_sub = fe_sub(u[2 * i], u[2 * i + 1]);
                for (u64 q = hm.rp[0][i]; q < hm.rp[0][i + 1]; ++q) { Fr& x = av[hm.col[0][q]]; x = fe_add(x, fe_mul(u_add, val_at(0, q))); }
                for (u64 q = hm.rp[1][i]; q < hm.rp[1][i + 1]; ++q) { Fr& x = av[hm.col[1][q]]; x = fe_add(x, fe_mul(u_sub, val_at(1, q))); }
            }
            av[0] = fe_add(av[0], u[2 * n]);
            for (u64 i = 1; i < l; ++i) {
                const Fr u1 = u[2 * n + 2 * i - 1], u2 = u[2 * n + 2 * i];
                av[i] = fe_add(av[i], fe_add(u1, u2));
                av[0] = fe_add(av[0], fe_sub(u1, u2));
            }
        });
        for (u64 i = 0; i < n; ++i) {
            const Fr u4 = fe_dbl(fe_dbl(u[2 * i]));
            for (u64 q = hm.rp[2][i]; q < hm.rp[2][i + 1]; ++q) { Fr& x = cv[hm.col[2][q]]; x = fe_add(x, fe_mul(u4, val_at(2, q))); }
            cv[m + i] = fe_add(cv[m + i], fe_add(u[2 * i], u[2 * i + 1]));
        }
        cv[0] = fe_add(cv[0], u[2 * n]);
        for (u64 i = 1; i < l; ++i) {
            const Fr u1 = u[2 * n + 2 * i - 1], u2 = u[2 * n + 2 * i];
            cv[i] = fe_add(cv[i], fe_dbl(fe_dbl(u1)));
            cv[m + n - 1 + i] = fe_add(cv[m + n - 1 + i], fe_add(u1, u2));
        }
        ta.join();

        const Fr ab = fe_add(alpha, beta);
        const Fr gz = fe_mul(gamma, zt), g2z = fe_mul(gamma, gz);
        DBuf d_a, d_c, d_aq, d_c1, d_c2, d_vq, d_t;
        d_a.ensure(M * sizeof(Fr)); d_c.ensure(M * sizeof(Fr));
        d_aq.ensure(M * sizeof(Fr)); d_c1.ensure(M * sizeof(Fr)); d_c2.ensure(M * sizeof(Fr)); d_vq.ensure(M * sizeof(Fr));
        d_t.ensure((D + 1) * sizeof(Fr));
        dev_h2d(d_a.p, av.data(), M * sizeof(Fr), s);
        dev_h2d(d_c.p, cv.data(), M * sizeof(Fr), s);
        ZK_LAUNCH((k_gm17_coeffs<Fr>), dim3(blocks_for(M, T)), dim3(T), 0, s, ptr<Fr>(d_a), ptr<Fr>(d_c), gamma, ab, fe_dbl(g2z), ptr<Fr>(d_aq),
                  ptr<Fr>(d_c1), ptr<Fr>(d_c2), ptr<Fr>(d_vq), M);
        ZK_LAUNCH((k_pow_table<Fr>), dim3(blocks_for(D + 1, T)), dim3(T), 0, s, ptr<Fr>(d_t), t, g2z, D + 1, 0u, 0u, 0);
        ZK_LAUNCH((k_from_mont<Fr>), dim3(blocks_for(D + 1, T)), dim3(T), 0, s, ptr<Fr>(d_t), ptr<Fr>(d_t), D + 1);
        // single points: G1: alpha, gamma, gamma Z, (alpha+beta) gamma Z, gamma^2 Z^2 ; G2: 1, beta, gamma, gamma Z
        Fr one_c = Fr::zero(); one_c.v[0] = 1;
        Fr singles[9] = {fe_from_mont(alpha), fe_from_mont(gamma), fe_from_mont(gz), fe_from_mont(fe_mul(ab, gz)), fe_from_mont(fe_sqr(gz)),
                         one_c, fe_from_mont(beta), fe_from_mont(gamma), fe_from_mont(gz)};
        DBuf d_single;
        d_single.ensure(sizeof(singles));
        dev_h2d(d_single.p, singles, sizeof(singles), s);
        stream_sync(s);

        DBuf tbl1, tbl2;
        S::template build_table<Fq>(ctx, S::template read_generator<Fq, 2>(g1b, C::g1_gen()), tbl1);
        S::template build_table<Fq2>(ctx, S::template read_generator<Fq2, 4>(g2b, C::g2_gen()), tbl2);
        uint8_t s1[5 * G1B], s2[4 * G2B];
        S::template mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_single), 5, s1);
        S::template mul_and_write<Fq2, 4>(ctx, tbl2, ptr<Fr>(d_single) + 5, 4, s2);
        uint8_t* p = out;
        memcpy(p, s2, G2B); p += G2B;                               // vk.h_g2
        memcpy(p, s1, G1B); p += G1B;                               // vk.g_alpha_g1
        memcpy(p, s2 + G2B, G2B); p += G2B;                         // vk.h_beta_g2
        memcpy(p, s1 + G1B, G1B); p += G1B;                         // vk.g_gamma_g1
        memcpy(p, s2 + 2 * G2B, G2B); p += G2B;                     // vk.h_gamma_g2
        p = S::put_len(p, l);
        S::template mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_vq), l, p); p += l * G1B;                 // vk.query
        p = S::put_len(p, M);
        S::template mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_aq), M, p); p += M * G1B;                 // a_query
        p = S::put_len(p, M);
        S::template mul_and_write<Fq2, 4>(ctx, tbl2, ptr<Fr>(d_aq), M, p); p += M * G2B;                // b_query
        p = S::put_len(p, M - l);
        S::template mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_c1) + l, M - l, p); p += (M - l) * G1B;   // c_query_1
        p = S::put_len(p, M);
        S::template mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_c2), M, p); p += M * G1B;                 // c_query_2
        memcpy(p, s1 + 2 * G1B, G1B); p += G1B;                     // g_gamma_z
        memcpy(p, s2 + 3 * G2B, G2B); p += G2B;                     // h_gamma_z
        memcpy(p, s1 + 3 * G1B, 2 * G1B); p += 2 * G1B;             // g_ab_gamma_z, g_gamma2_z2
        p = S::put_len(p, D + 1);
        S::template mul_and_write<Fq, 2>(ctx, tbl1, ptr<Fr>(d_t), D + 1, p); p += (D + 1) * G1B;        // g_gamma2_z_t
        require((u64)(p - out) == need, ZKHIP_ERR_DEVICE, "internal: key size mismatch");
    }
};

}  // namespace zk
