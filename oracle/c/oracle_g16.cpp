// oracle/c/oracle_g16.cpp — TEST ORACLE / CPU BASELINE ONLY.  Not part of libzkhip; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this file.
//
// C++ restatement of the CPU path the reference drives at
//   /root/reference/zokrates_ark/src/groth16.rs:44   Groth16::<E>::prove(&pk, computation, rng)
//   /root/reference/zokrates_ark/src/groth16.rs:95   Groth16::<E>::circuit_specific_setup(...)
// whose arithmetic lives in the un-vendored crates ark-groth16/ark-poly/ark-ec/ark-ff 0.3.0
// (/root/reference/Cargo.lock:79-378).  Algorithm per SURVEY.md App. A.3 (prover), A.4 (radix-2 domain),
// A.5 (Pippenger: c = ln-heuristic, zero skip, scalar-one fast path, unsigned digits, windows in parallel),
// A.6 (setup; here with fixed generators + caller-supplied toxic waste), A.9 (closed-form trapdoor proof).
// Parity status: "parity unpinned" against the real crates (no Rust toolchain); pinned by
// tests/test_oracle_c.py against the python big-int oracle (O1 == O2, O3 accepts) bit-for-bit.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <functional>
#include <memory>
#include <thread>
#include <vector>
#include "ec.hpp"

namespace orc {

// ---------------- parameters ----------------
#define ORC_FIELD(Tag, NL, ...)                                         \
    struct Tag {                                                        \
        static constexpr int N = NL;                                    \
        static const FieldParams<NL>& params() {                        \
            static const u64 m[NL] = {__VA_ARGS__};                     \
            static const FieldParams<NL> p(m);                          \
            return p;                                                   \
        }                                                               \
    };

ORC_FIELD(Bn254FrTag, 4, 0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull)
ORC_FIELD(Bn254FqTag, 4, 0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull)
ORC_FIELD(Bls381FrTag, 4, 0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull)
ORC_FIELD(Bls381FqTag, 6, 0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull,
          0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull)

static void hex_to_bytes_le(const char* hex, uint8_t* out, int nbytes) {
    std::string s(hex);
    memset(out, 0, nbytes);
    int n = (int)s.size();
    for (int i = 0; i < n; ++i) {
        char ch = s[n - 1 - i];
        int d = ch <= '9' ? ch - '0' : (ch | 0x20) - 'a' + 10;
        out[i / 2] |= (uint8_t)(d << (4 * (i & 1)));
    }
}
template <class F>
static F fhex(const char* h) { uint8_t b[F::BYTES]; hex_to_bytes_le(h, b, F::BYTES); return F::from_bytes(b); }

struct Bn254 {
    typedef Fp<Bn254FrTag> Fr;
    typedef Fp<Bn254FqTag> Fq;
    typedef Fp2<Fq> Fq2;
    static constexpr int TWO_ADICITY = 28;
    static constexpr u64 GENERATOR = 5;
    static Affine<Fq> g1() { return {Fq::from_u64(1), Fq::from_u64(2), false}; }
    static Affine<Fq2> g2() {
        return {{fhex<Fq>("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"),
                 fhex<Fq>("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2")},
                {fhex<Fq>("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"),
                 fhex<Fq>("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b")},
                false};
    }
};
struct Bls381 {
    typedef Fp<Bls381FrTag> Fr;
    typedef Fp<Bls381FqTag> Fq;
    typedef Fp2<Fq> Fq2;
    static constexpr int TWO_ADICITY = 32;
    static constexpr u64 GENERATOR = 7;
    static Affine<Fq> g1() {
        return {fhex<Fq>("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
                fhex<Fq>("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"),
                false};
    }
    static Affine<Fq2> g2() {
        return {{fhex<Fq>("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
                 fhex<Fq>("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")},
                {fhex<Fq>("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
                 fhex<Fq>("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")},
                false};
    }
};

// ---------------- threading ----------------
static void parallel_for(int threads, size_t total, const std::function<void(size_t, size_t)>& fn) {
    if (threads <= 1 || total < 2) { fn(0, total); return; }
    size_t nt = std::min<size_t>(threads, total);
    std::vector<std::thread> th;
    size_t chunk = (total + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        size_t b = t * chunk, e = std::min(total, b + chunk);
        if (b >= e) break;
        th.emplace_back([=, &fn] { fn(b, e); });
    }
    for (auto& t : th) t.join();
}
static void parallel_dynamic(int threads, size_t total, const std::function<void(size_t)>& fn) {
    if (threads <= 1) { for (size_t i = 0; i < total; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&] { for (size_t i; (i = next++) < total;) fn(i); });
    for (auto& t : th) t.join();
}

struct SplitMix64 {
    u64 s;
    explicit SplitMix64(u64 seed) : s(seed) {}
    u64 next() {
        s += 0x9E3779B97F4A7C15ull;
        u64 z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    template <class F>
    F field() {  // rejection sampling on bit_length(p) bits; same stream as oracle/fields.py
        const auto& p = F::P();
        for (;;) {
            u64 c[F::N];
            for (int i = 0; i < F::N; ++i) c[i] = next();
            int top = p.bits - 64 * (F::N - 1);
            if (top < 64) c[F::N - 1] &= ((u64)1 << top) - 1;
            if (!FieldParams<F::N>::geq(c, p.mod)) return F::from_canonical_limbs(c);
        }
    }
};

// ---------------- radix-2 domain (App. A.4) ----------------
template <class C>
struct Domain {
    typedef typename C::Fr Fr;
    size_t N; int logN;
    Fr omega, omega_inv, n_inv, g, g_inv;
    explicit Domain(size_t n) : N(n) {
        logN = 0; while (((size_t)1 << logN) < N) ++logN;
        Fr gen = Fr::from_u64(C::GENERATOR);
        u64 e[Fr::N]; memcpy(e, Fr::P().mod, sizeof(e));   // (r-1) >> S
        e[0] -= 1;
        for (int s = 0; s < C::TWO_ADICITY; ++s) {
            for (int i = 0; i < Fr::N; ++i) e[i] = (e[i] >> 1) | (i + 1 < Fr::N ? e[i + 1] << 63 : 0);
        }
        Fr root = gen.pow_limbs(e, Fr::N);
        for (int i = logN; i < C::TWO_ADICITY; ++i) root = root.sqr();
        omega = root; omega_inv = root.inverse();
        n_inv = Fr::from_u64(N).inverse();
        g = gen; g_inv = gen.inverse();
    }
    void fft_core(std::vector<Fr>& a, const Fr& root, int threads) const {
        // bit reversal, then DIT stages (natural in / natural out)
        for (size_t i = 1, j = 0; i < N; ++i) {
            size_t bit = N >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j |= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        std::vector<Fr> tw(N / 2 ? N / 2 : 1);
        tw[0] = Fr::one();
        for (size_t i = 1; i < N / 2; ++i) tw[i] = tw[i - 1] * root;
        for (size_t len = 2; len <= N; len <<= 1) {
            size_t half = len / 2, step = N / len;
            parallel_for(threads, N / 2, [&](size_t b, size_t e) {
                for (size_t k = b; k < e; ++k) {
                    size_t grp = k / half, t = k % half;
                    size_t i0 = grp * len + t, i1 = i0 + half;
                    Fr u = a[i0], v = a[i1] * tw[t * step];
                    a[i0] = u + v; a[i1] = u - v;
                }
            });
        }
    }
    void scale_powers(std::vector<Fr>& a, const Fr& base, const Fr& c0, int threads) const {
        parallel_for(threads, N, [&](size_t b, size_t e) {
            u64 bl[1] = {b};
            Fr p = base.pow_limbs(bl, 1) * c0;
            for (size_t i = b; i < e; ++i) { a[i] = a[i] * p; p = p * base; }
        });
    }
    void fft(std::vector<Fr>& a, int th) const { a.resize(N, Fr::zero()); fft_core(a, omega, th); }
    void ifft(std::vector<Fr>& a, int th) const {
        a.resize(N, Fr::zero()); fft_core(a, omega_inv, th);
        parallel_for(th, N, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) a[i] = a[i] * n_inv; });
    }
    void coset_fft(std::vector<Fr>& a, int th) const { a.resize(N, Fr::zero()); scale_powers(a, g, Fr::one(), th); fft_core(a, omega, th); }
    void coset_ifft(std::vector<Fr>& a, int th) const { a.resize(N, Fr::zero()); fft_core(a, omega_inv, th); scale_powers(a, g_inv, n_inv, th); }
};

// ---------------- Pippenger MSM (App. A.5) ----------------
static int ark_window_size(size_t size) {
    if (size < 32) return 3;
    int lg = 0; while (((size_t)1 << lg) < size) ++lg;   // ark_std::log2 = ceil(log2)
    return lg * 69 / 100 + 2;
}
template <int NL>
static inline u64 window_digit(const u64* k, int start, int c) {
    int limb = start / 64, off = start % 64;
    u64 d = k[limb] >> off;
    if (off + c > 64 && limb + 1 < NL) d |= k[limb + 1] << (64 - off);
    return d & (((u64)1 << c) - 1);
}
// scalars: canonical limbs, NL per scalar
template <class F, int NL>
static Jac<F> msm(const Affine<F>* bases, const u64* scalars, size_t size, int num_bits, int threads) {
    if (size == 0) return Jac<F>::infinity();
    int c = ark_window_size(size);
    int nwin = (num_bits + c - 1) / c;
    std::vector<Jac<F>> wsum(nwin);
    parallel_dynamic(std::min(threads, nwin), nwin, [&](size_t w) {
        int start = (int)w * c;
        Jac<F> res = Jac<F>::infinity();
        std::vector<Jac<F>> buckets(((size_t)1 << c) - 1, Jac<F>::infinity());
        for (size_t i = 0; i < size; ++i) {
            const u64* k = scalars + i * NL;
            bool zero = true, one = k[0] == 1;
            for (int j = 0; j < NL; ++j) { zero &= k[j] == 0; if (j) one &= k[j] == 0; }
            if (zero) continue;
            if (one) { if (start == 0) res = res.add_mixed(bases[i]); continue; }
            u64 d = window_digit<NL>(k, start, c);
            if (d) buckets[d - 1] = buckets[d - 1].add_mixed(bases[i]);
        }
        Jac<F> running = Jac<F>::infinity();
        for (size_t b = buckets.size(); b-- > 0;) { running = running.add(buckets[b]); res = res.add(running); }
        wsum[w] = res;
    });
    Jac<F> total = Jac<F>::infinity();
    for (int w = nwin - 1; w >= 1; --w) {
        total = total.add(wsum[w]);
        for (int i = 0; i < c; ++i) total = total.dbl();
    }
    return total.add(wsum[0]);
}

// ---------------- R1CS in ark variable order ----------------
template <class C>
struct Csr {
    std::vector<u64> rowptr;
    std::vector<uint32_t> col;
    std::vector<typename C::Fr> val;
    void push_row() { rowptr.push_back(col.size()); }
    void begin() { rowptr.assign(1, 0); }
    void add(uint32_t c, const typename C::Fr& v) { col.push_back(c); val.push_back(v); }
    void end_row() { rowptr.push_back(col.size()); }
};

struct CircuitBase { int curve; virtual ~CircuitBase() {} };
struct PkBase { int curve; virtual ~PkBase() {} };

template <class C>
struct Circuit : CircuitBase {
    typedef typename C::Fr Fr;
    u64 n = 0, l = 0, w = 0;
    Csr<C> A, B, Cm;
    std::vector<Fr> z;     // optional satisfying assignment (synthetic circuits)
    u64 m() const { return l + w; }
    u64 domain() const { u64 N = 1; while (N < n + l) N <<= 1; return N; }
    Fr row_dot(const Csr<C>& M, u64 i, const std::vector<Fr>& zz) const {
        Fr acc = Fr::zero();
        for (u64 k = M.rowptr[i]; k < M.rowptr[i + 1]; ++k) acc = acc + M.val[k] * zz[M.col[k]];
        return acc;
    }
};

template <class C>
struct Pk : PkBase {
    typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    Affine<Fq> alpha_g1; Affine<Fq2> beta_g2, gamma_g2, delta_g2;
    std::vector<Affine<Fq>> gamma_abc_g1;
    Affine<Fq> beta_g1, delta_g1;
    std::vector<Affine<Fq>> a_query, b_g1_query, h_query, l_query;
    std::vector<Affine<Fq2>> b_g2_query;

    size_t byte_size() const {
        size_t g1 = Affine<Fq>::BYTES, g2 = Affine<Fq2>::BYTES;
        return g1 + 3 * g2 + 8 + gamma_abc_g1.size() * g1 + 2 * g1 + 8 + a_query.size() * g1 + 8 + b_g1_query.size() * g1 +
               8 + b_g2_query.size() * g2 + 8 + h_query.size() * g1 + 8 + l_query.size() * g1;
    }
    template <class A> static uint8_t* put(uint8_t* p, const A& a) { a.to_bytes(p); return p + A::BYTES; }
    template <class A> static uint8_t* putv(uint8_t* p, const std::vector<A>& v) {
        u64 n = v.size(); memcpy(p, &n, 8); p += 8;
        for (auto& a : v) p = put(p, a);
        return p;
    }
    void serialize(uint8_t* p) const {   // ark ProvingKey::serialize_unchecked (App. B.3)
        p = put(p, alpha_g1); p = put(p, beta_g2); p = put(p, gamma_g2); p = put(p, delta_g2);
        p = putv(p, gamma_abc_g1); p = put(p, beta_g1); p = put(p, delta_g1);
        p = putv(p, a_query); p = putv(p, b_g1_query); p = putv(p, b_g2_query); p = putv(p, h_query); p = putv(p, l_query);
    }
    struct Rd { const uint8_t* p; const uint8_t* e; bool ok = true; };
    template <class A> static void get(Rd& r, A& a) {
        if (!r.ok || r.p + A::BYTES > r.e) { r.ok = false; return; }
        a = A::from_bytes(r.p); r.p += A::BYTES;
    }
    template <class A> static void getv(Rd& r, std::vector<A>& v) {
        if (!r.ok || r.p + 8 > r.e) { r.ok = false; return; }
        u64 n; memcpy(&n, r.p, 8); r.p += 8;
        if (n > (u64)(r.e - r.p) / A::BYTES) { r.ok = false; return; }
        v.resize(n);
        for (auto& a : v) get(r, a);
    }
    bool parse(const uint8_t* p, size_t len) {
        Rd r{p, p + len};
        get(r, alpha_g1); get(r, beta_g2); get(r, gamma_g2); get(r, delta_g2); getv(r, gamma_abc_g1);
        get(r, beta_g1); get(r, delta_g1);
        getv(r, a_query); getv(r, b_g1_query); getv(r, b_g2_query); getv(r, h_query); getv(r, l_query);
        return r.ok && r.p == r.e;
    }
};

// ---------------- synthetic circuits (same stream as oracle/groth16.py) ----------------
template <class C>
static Circuit<C>* synth(u64 n, u64 seed, int kind) {
    typedef typename C::Fr Fr;
    auto* c = new Circuit<C>();
    SplitMix64 rng(seed);
    Fr one = Fr::one();
    Fr x = rng.field<Fr>();
    c->z = {one, x, rng.field<Fr>(), rng.field<Fr>()};
    c->l = 2; c->n = n;
    c->A.begin(); c->B.begin(); c->Cm.begin();
    uint32_t i0 = 2, i1 = 3;
    for (u64 k = 0; k < n; ++k) {
        bool boolean = kind == 1 && (rng.next() % 10) != 0;
        uint32_t col = (uint32_t)c->z.size();
        if (boolean) {
            u64 bit = rng.next() & 1;
            c->z.push_back(bit ? one : Fr::zero());
            c->A.add(col, one); c->A.end_row();
            c->B.add(col, one); c->B.add(0, one.neg()); c->B.end_row();
            c->Cm.end_row();
        } else {
            Fr cc = Fr::from_u64(rng.next()), dd = Fr::from_u64(rng.next());
            Fr val = (c->z[i0] + cc) * (c->z[i1] + dd * x);
            c->z.push_back(val);
            c->A.add(i0, one); c->A.add(0, cc); c->A.end_row();
            c->B.add(i1, one); c->B.add(1, dd); c->B.end_row();
            c->Cm.add(col, one); c->Cm.end_row();
            i0 = i1; i1 = col;
        }
    }
    c->w = c->z.size() - c->l;
    return c;
}

// ---------------- QAP evaluation at tau (App. A.6) ----------------
template <class C>
struct QapTau {
    typedef typename C::Fr Fr;
    std::vector<Fr> a, b, c;
    Fr zt;
    u64 N;
};
template <class C>
static QapTau<C> qap_at_tau(const Circuit<C>& cs, const typename C::Fr& tau, int threads) {
    typedef typename C::Fr Fr;
    QapTau<C> q;
    q.N = cs.domain();
    Domain<C> dom(q.N);
    u64 Nl[1] = {q.N};
    q.zt = tau.pow_limbs(Nl, 1) - Fr::one();
    // u_k = zt/N * w^k / (tau - w^k); batch inversion of (tau - w^k)
    std::vector<Fr> den(q.N), wk(q.N);
    Fr p = Fr::one();
    for (u64 k = 0; k < q.N; ++k) { wk[k] = p; den[k] = tau - p; p = p * dom.omega; }
    std::vector<Fr> pref(q.N);
    Fr acc = Fr::one();
    for (u64 k = 0; k < q.N; ++k) { pref[k] = acc; acc = acc * den[k]; }
    Fr inv = acc.inverse();
    std::vector<Fr> u(q.N);
    Fr zn = q.zt * dom.n_inv;
    for (u64 k = q.N; k-- > 0;) { Fr di = inv * pref[k]; inv = inv * den[k]; u[k] = zn * wk[k] * di; }
    q.a.assign(cs.m(), Fr::zero()); q.b.assign(cs.m(), Fr::zero()); q.c.assign(cs.m(), Fr::zero());
    auto scatter = [&](const Csr<C>& M, std::vector<Fr>& out) {
        for (u64 i = 0; i < cs.n; ++i)
            for (u64 k = M.rowptr[i]; k < M.rowptr[i + 1]; ++k) out[M.col[k]] = out[M.col[k]] + M.val[k] * u[i];
    };
    std::thread ta([&] { scatter(cs.A, q.a); }), tb([&] { scatter(cs.B, q.b); });
    scatter(cs.Cm, q.c);
    ta.join(); tb.join();
    for (u64 i = 0; i < cs.l; ++i) q.a[i] = q.a[i] + u[cs.n + i];
    (void)threads;
    return q;
}

// fixed-base windowed multiplication table: tbl[j][d] = d * 2^(W*j) * G (affine)
template <class F>
struct FixedBase {
    static constexpr int W = 8;
    int nwin;
    std::vector<Affine<F>> tbl;   // nwin * 2^W
    FixedBase(const Affine<F>& g, int bits) {
        nwin = (bits + W - 1) / W;
        std::vector<Jac<F>> t((size_t)nwin << W);
        Jac<F> base = Jac<F>::from_affine(g);
        for (int j = 0; j < nwin; ++j) {
            Jac<F>* row = &t[(size_t)j << W];
            row[0] = Jac<F>::infinity();
            for (int d = 1; d < (1 << W); ++d) row[d] = row[d - 1].add(base);
            for (int i = 0; i < W; ++i) base = base.dbl();
        }
        batch_to_affine(t, tbl);
    }
    Jac<F> mul(const u64* k, int nl) const {
        Jac<F> acc = Jac<F>::infinity();
        for (int j = 0; j < nwin; ++j) {
            int bit = j * W, limb = bit / 64;
            if (limb >= nl) break;
            u64 d = (k[limb] >> (bit % 64)) & ((1 << W) - 1);
            if (d) acc = acc.add_mixed(tbl[((size_t)j << W) + d]);
        }
        return acc;
    }
};
template <class F, class Fr>
static void fixed_mul_vec(const FixedBase<F>& fb, const std::vector<Fr>& k, std::vector<Affine<F>>& out, int threads) {
    std::vector<Jac<F>> j(k.size());
    parallel_for(threads, k.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) { u64 c[Fr::N]; k[i].to_canonical_limbs(c); j[i] = fb.mul(c, Fr::N); }
    });
    // chunked batch normalisation in parallel
    out.resize(k.size());
    parallel_for(threads, k.size(), [&](size_t b, size_t e) {
        std::vector<Jac<F>> part(j.begin() + b, j.begin() + e);
        std::vector<Affine<F>> aff;
        batch_to_affine(part, aff);
        std::copy(aff.begin(), aff.end(), out.begin() + b);
    });
}

template <class C>
struct ToxicT { typename C::Fr alpha, beta, gamma, delta, tau; };

template <class C>
static Pk<C>* setup(const Circuit<C>& cs, const ToxicT<C>& tx, int threads) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    auto q = qap_at_tau(cs, tx.tau, threads);
    FixedBase<Fq> t1(C::g1(), Fr::P().bits);
    FixedBase<Fq2> t2(C::g2(), Fr::P().bits);
    auto* pk = new Pk<C>();
    Fr gi = tx.gamma.inverse(), di = tx.delta.inverse();
    u64 m = cs.m();
    std::vector<Fr> abc(m), s;
    for (u64 i = 0; i < m; ++i) abc[i] = tx.beta * q.a[i] + tx.alpha * q.b[i] + q.c[i];
    std::vector<Affine<Fq>> tmp1; std::vector<Affine<Fq2>> tmp2;
    fixed_mul_vec(t1, std::vector<Fr>{tx.alpha, tx.beta, tx.delta}, tmp1, 1);
    pk->alpha_g1 = tmp1[0]; pk->beta_g1 = tmp1[1]; pk->delta_g1 = tmp1[2];
    fixed_mul_vec(t2, std::vector<Fr>{tx.beta, tx.gamma, tx.delta}, tmp2, 1);
    pk->beta_g2 = tmp2[0]; pk->gamma_g2 = tmp2[1]; pk->delta_g2 = tmp2[2];
    s.resize(cs.l); for (u64 i = 0; i < cs.l; ++i) s[i] = abc[i] * gi;
    fixed_mul_vec(t1, s, pk->gamma_abc_g1, threads);
    fixed_mul_vec(t1, q.a, pk->a_query, threads);
    fixed_mul_vec(t1, q.b, pk->b_g1_query, threads);
    fixed_mul_vec(t2, q.b, pk->b_g2_query, threads);
    s.resize(q.N - 1);
    Fr p = q.zt * di;
    for (u64 i = 0; i + 1 < q.N; ++i) { s[i] = p; p = p * tx.tau; }
    fixed_mul_vec(t1, s, pk->h_query, threads);
    s.resize(cs.w); for (u64 j = 0; j < cs.w; ++j) s[j] = abc[cs.l + j] * di;
    fixed_mul_vec(t1, s, pk->l_query, threads);
    return pk;
}

// ---------------- prover (App. A.3) ----------------
struct Timings { double matvec, fft, msm_h, msm_l, msm_a, msm_b1, msm_b2, total; };
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class C>
static void witness_map(const Circuit<C>& cs, const std::vector<typename C::Fr>& z, std::vector<typename C::Fr>& h, int th, Timings* tm) {
    typedef typename C::Fr Fr;
    u64 N = cs.domain();
    Domain<C> dom(N);
    double t0 = now();
    std::vector<Fr> a(N, Fr::zero()), b(N, Fr::zero()), c(N, Fr::zero());
    parallel_for(th, cs.n, [&](size_t s, size_t e) {
        for (size_t i = s; i < e; ++i) { a[i] = cs.row_dot(cs.A, i, z); b[i] = cs.row_dot(cs.B, i, z); c[i] = cs.row_dot(cs.Cm, i, z); }
    });
    for (u64 j = 0; j < cs.l; ++j) a[cs.n + j] = z[j];
    double t1 = now();
    dom.ifft(a, th); dom.ifft(b, th);
    dom.coset_fft(a, th); dom.coset_fft(b, th);
    dom.ifft(c, th); dom.coset_fft(c, th);
    u64 Nl[1] = {N};
    Fr zinv = (dom.g.pow_limbs(Nl, 1) - Fr::one()).inverse();
    parallel_for(th, N, [&](size_t s, size_t e) { for (size_t i = s; i < e; ++i) a[i] = (a[i] * b[i] - c[i]) * zinv; });
    dom.coset_ifft(a, th);
    h.swap(a);
    if (tm) { tm->matvec = t1 - t0; tm->fft = now() - t1; }
}

template <class Fr>
static void to_repr(const Fr* v, size_t n, std::vector<u64>& out, int th) {
    out.resize(n * Fr::N);
    parallel_for(th, n, [&](size_t s, size_t e) { for (size_t i = s; i < e; ++i) v[i].to_canonical_limbs(&out[i * Fr::N]); });
}

template <class C>
struct ProofT { Affine<typename C::Fq> a; Affine<typename C::Fq2> b; Affine<typename C::Fq> c; };

template <class C>
static int prove(const Circuit<C>& cs, const Pk<C>& pk, const std::vector<typename C::Fr>& z, const typename C::Fr& r,
                 const typename C::Fr& s, ProofT<C>& out, int th, Timings* tm) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    u64 m = cs.m(), N = cs.domain();
    if (z.size() != m || pk.a_query.size() != m || pk.b_g1_query.size() != m || pk.b_g2_query.size() != m ||
        pk.h_query.size() != N - 1 || pk.l_query.size() != cs.w)
        return -1;
    double t00 = now();
    std::vector<Fr> h;
    witness_map(cs, z, h, th, tm);
    const int NB = Fr::P().bits;
    std::vector<u64> hr, zr;
    double t0 = now();
    to_repr(h.data(), N - 1, hr, th);
    Jac<Fq> H = msm<Fq, Fr::N>(pk.h_query.data(), hr.data(), N - 1, NB, th);
    double t1 = now();
    to_repr(z.data(), m, zr, th);
    Jac<Fq> L = msm<Fq, Fr::N>(pk.l_query.data(), zr.data() + cs.l * Fr::N, cs.w, NB, th);
    double t2 = now();
    u64 rl[Fr::N], sl[Fr::N], rsl[Fr::N];
    r.to_canonical_limbs(rl); s.to_canonical_limbs(sl); (r * s).to_canonical_limbs(rsl);
    Jac<Fq> d1 = Jac<Fq>::from_affine(pk.delta_g1);
    // calculate_coeff(initial = delta*rs, query, vk_param, assignment = z[1..])
    Jac<Fq> gA = msm<Fq, Fr::N>(pk.a_query.data() + 1, zr.data() + Fr::N, m - 1, NB, th);
    gA = d1.mul_limbs(rl, Fr::N).add_mixed(pk.a_query[0]).add(gA).add_mixed(pk.alpha_g1);
    double t3 = now();
    Jac<Fq> gB1 = Jac<Fq>::infinity();
    if (!r.is_zero()) {
        gB1 = msm<Fq, Fr::N>(pk.b_g1_query.data() + 1, zr.data() + Fr::N, m - 1, NB, th);
        gB1 = d1.mul_limbs(sl, Fr::N).add_mixed(pk.b_g1_query[0]).add(gB1).add_mixed(pk.beta_g1);
    }
    double t4 = now();
    Jac<Fq2> gB2 = msm<Fq2, Fr::N>(pk.b_g2_query.data() + 1, zr.data() + Fr::N, m - 1, NB, th);
    gB2 = Jac<Fq2>::from_affine(pk.delta_g2).mul_limbs(sl, Fr::N).add_mixed(pk.b_g2_query[0]).add(gB2).add_mixed(pk.beta_g2);
    double t5 = now();
    Jac<Fq> gC = gA.mul_limbs(sl, Fr::N).add(gB1.mul_limbs(rl, Fr::N)).add(d1.mul_limbs(rsl, Fr::N).neg()).add(L).add(H);
    out.a = gA.to_affine(); out.b = gB2.to_affine(); out.c = gC.to_affine();
    if (tm) { tm->msm_h = t1 - t0; tm->msm_l = t2 - t1; tm->msm_a = t3 - t2; tm->msm_b1 = t4 - t3; tm->msm_b2 = t5 - t4; tm->total = now() - t00; }
    return 0;
}

template <class C>
static void trapdoor(const Circuit<C>& cs, const ToxicT<C>& tx, const std::vector<typename C::Fr>& z, const typename C::Fr& r,
                     const typename C::Fr& s, ProofT<C>& out, int th) {
    typedef typename C::Fr Fr;
    auto q = qap_at_tau(cs, tx.tau, th);
    Fr Az = Fr::zero(), Bz = Fr::zero(), Cz = Fr::zero(), priv = Fr::zero();
    for (u64 i = 0; i < cs.m(); ++i) {
        Az = Az + z[i] * q.a[i]; Bz = Bz + z[i] * q.b[i]; Cz = Cz + z[i] * q.c[i];
        if (i >= cs.l) priv = priv + z[i] * (tx.beta * q.a[i] + tx.alpha * q.b[i] + q.c[i]);
    }
    Fr di = tx.delta.inverse();
    Fr la = tx.alpha + Az + r * tx.delta, lb = tx.beta + Bz + s * tx.delta;
    Fr lc = (priv + (Az * Bz - Cz)) * di + s * la + r * lb - r * s * tx.delta;
    u64 k[Fr::N];
    la.to_canonical_limbs(k); out.a = Jac<typename C::Fq>::from_affine(C::g1()).mul_limbs(k, Fr::N).to_affine();
    lb.to_canonical_limbs(k); out.b = Jac<typename C::Fq2>::from_affine(C::g2()).mul_limbs(k, Fr::N).to_affine();
    lc.to_canonical_limbs(k); out.c = Jac<typename C::Fq>::from_affine(C::g1()).mul_limbs(k, Fr::N).to_affine();
}

template <class C>
static void proof_to_raw(const ProofT<C>& p, uint8_t* out) {
    typedef typename C::Fq Fq;
    const int nb = Fq::BYTES;
    memset(out, 0, 8 * nb + 3);
    if (!p.a.inf) { p.a.x.to_bytes(out); p.a.y.to_bytes(out + nb); }
    if (!p.b.inf) { p.b.x.to_bytes(out + 2 * nb); p.b.y.to_bytes(out + 4 * nb); }
    if (!p.c.inf) { p.c.x.to_bytes(out + 6 * nb); p.c.y.to_bytes(out + 7 * nb); }
    out[8 * nb] = p.a.inf; out[8 * nb + 1] = p.b.inf; out[8 * nb + 2] = p.c.inf;
}

template <class C>
static ToxicT<C> toxic_from_bytes(const uint8_t* t) {
    typedef typename C::Fr Fr;
    return {Fr::from_bytes(t), Fr::from_bytes(t + 32), Fr::from_bytes(t + 64), Fr::from_bytes(t + 96), Fr::from_bytes(t + 128)};
}

#include "gm17.hpp"

}  // namespace orc

// =====================================================================================
// C API (ctypes).  curve: 0 = bn128 (BN254), 1 = bls12_381.  All field values canonical LE bytes.
// =====================================================================================
using namespace orc;
#define DISPATCH(curve, ...)                              \
    do {                                                  \
        if ((curve) == 0) { typedef Bn254 C; __VA_ARGS__; } \
        else { typedef Bls381 C; __VA_ARGS__; }           \
    } while (0)

extern "C" {

void* orc_circuit_synth(int curve, uint64_t n, uint64_t seed, int kind) {
    CircuitBase* c = nullptr;
    DISPATCH(curve, c = synth<C>(n, seed, kind));
    c->curve = curve;
    return c;
}

// Build from CSR arrays (values canonical LE, 32 B each).
void* orc_circuit_from_csr(int curve, uint64_t n, uint64_t l, uint64_t w, const uint64_t* rpA, const uint32_t* cA, const uint8_t* vA,
                           const uint64_t* rpB, const uint32_t* cB, const uint8_t* vB, const uint64_t* rpC, const uint32_t* cC,
                           const uint8_t* vC) {
    CircuitBase* out = nullptr;
    DISPATCH(curve, {
        auto* c = new Circuit<C>();
        c->n = n; c->l = l; c->w = w;
        auto fill = [&](Csr<C>& M, const uint64_t* rp, const uint32_t* ci, const uint8_t* v) {
            M.rowptr.assign(rp, rp + n + 1);
            M.col.assign(ci, ci + rp[n]);
            M.val.resize(rp[n]);
            for (u64 k = 0; k < rp[n]; ++k) M.val[k] = C::Fr::from_bytes(v + 32 * k);
        };
        fill(c->A, rpA, cA, vA); fill(c->B, rpB, cB, vB); fill(c->Cm, rpC, cC, vC);
        out = c;
    });
    out->curve = curve;
    return out;
}
void orc_circuit_free(void* c) { delete (CircuitBase*)c; }

void orc_circuit_dims(void* h, uint64_t* out /* n,l,w,nnzA,nnzB,nnzC,N */) {
    DISPATCH(((CircuitBase*)h)->curve, {
        auto* c = (Circuit<C>*)h;
        out[0] = c->n; out[1] = c->l; out[2] = c->w; out[3] = c->A.col.size(); out[4] = c->B.col.size(); out[5] = c->Cm.col.size();
        out[6] = c->domain();
    });
}
// which: 0 A, 1 B, 2 C
void orc_circuit_export(void* h, int which, uint64_t* rowptr, uint32_t* col, uint8_t* val) {
    DISPATCH(((CircuitBase*)h)->curve, {
        auto* c = (Circuit<C>*)h;
        const Csr<C>& M = which == 0 ? c->A : which == 1 ? c->B : c->Cm;
        memcpy(rowptr, M.rowptr.data(), M.rowptr.size() * 8);
        memcpy(col, M.col.data(), M.col.size() * 4);
        for (size_t k = 0; k < M.val.size(); ++k) M.val[k].to_bytes(val + 32 * k);
    });
}
void orc_circuit_assignment(void* h, uint8_t* z) {
    DISPATCH(((CircuitBase*)h)->curve, {
        auto* c = (Circuit<C>*)h;
        for (size_t i = 0; i < c->z.size(); ++i) c->z[i].to_bytes(z + 32 * i);
    });
}

void* orc_setup(void* h, const uint8_t* toxic /* alpha,beta,gamma,delta,tau: 5 x 32 B */, int threads) {
    PkBase* pk = nullptr;
    int curve = ((CircuitBase*)h)->curve;
    DISPATCH(curve, pk = setup<C>(*(Circuit<C>*)h, toxic_from_bytes<C>(toxic), threads));
    pk->curve = curve;
    return pk;
}
void* orc_pk_parse(int curve, const uint8_t* data, uint64_t len) {
    PkBase* out = nullptr;
    DISPATCH(curve, {
        auto* pk = new Pk<C>();
        if (pk->parse(data, len)) out = pk; else delete pk;
    });
    if (out) out->curve = curve;
    return out;
}
uint64_t orc_pk_size(void* h) { uint64_t s = 0; DISPATCH(((PkBase*)h)->curve, s = ((Pk<C>*)h)->byte_size()); return s; }
void orc_pk_serialize(void* h, uint8_t* out) { DISPATCH(((PkBase*)h)->curve, ((Pk<C>*)h)->serialize(out)); }
void orc_pk_free(void* h) { delete (PkBase*)h; }

// timings: 8 doubles (matvec, fft, msm_h, msm_l, msm_a, msm_b1, msm_b2, total) or NULL
int orc_prove(void* ch, void* pkh, const uint8_t* z, const uint8_t* r, const uint8_t* s, uint8_t* proof_raw, int threads, double* timings) {
    int rc = -1;
    DISPATCH(((CircuitBase*)ch)->curve, {
        auto* c = (Circuit<C>*)ch;
        std::vector<C::Fr> zz(c->m());
        for (size_t i = 0; i < zz.size(); ++i) zz[i] = C::Fr::from_bytes(z + 32 * i);
        ProofT<C> p;
        Timings tm{};
        rc = prove<C>(*c, *(Pk<C>*)pkh, zz, C::Fr::from_bytes(r), C::Fr::from_bytes(s), p, threads, &tm);
        if (rc == 0) proof_to_raw<C>(p, proof_raw);
        if (timings) memcpy(timings, &tm, sizeof(tm));
    });
    return rc;
}
int orc_trapdoor(void* ch, const uint8_t* toxic, const uint8_t* z, const uint8_t* r, const uint8_t* s, uint8_t* proof_raw, int threads) {
    DISPATCH(((CircuitBase*)ch)->curve, {
        auto* c = (Circuit<C>*)ch;
        std::vector<C::Fr> zz(c->m());
        for (size_t i = 0; i < zz.size(); ++i) zz[i] = C::Fr::from_bytes(z + 32 * i);
        ProofT<C> p;
        trapdoor<C>(*c, toxic_from_bytes<C>(toxic), zz, C::Fr::from_bytes(r), C::Fr::from_bytes(s), p, threads);
        proof_to_raw<C>(p, proof_raw);
    });
    return 0;
}
// h coefficients of the QAP quotient (length N), for testing the device witness_map
int orc_witness_map(void* ch, const uint8_t* z, uint8_t* h_out, int threads) {
    DISPATCH(((CircuitBase*)ch)->curve, {
        auto* c = (Circuit<C>*)ch;
        std::vector<C::Fr> zz(c->m()), h;
        for (size_t i = 0; i < zz.size(); ++i) zz[i] = C::Fr::from_bytes(z + 32 * i);
        witness_map<C>(*c, zz, h, threads, nullptr);
        for (size_t i = 0; i < h.size(); ++i) h[i].to_bytes(h_out + 32 * i);
    });
    return 0;
}

// ---- primitives ----
// dir: 0 fft, 1 ifft, 2 coset_fft, 3 coset_ifft ; data = 2^logn x 32 B canonical LE, in place
int orc_ntt(int curve, int logn, int dir, uint8_t* data, int threads) {
    DISPATCH(curve, {
        size_t N = (size_t)1 << logn;
        Domain<C> dom(N);
        std::vector<C::Fr> a(N);
        for (size_t i = 0; i < N; ++i) a[i] = C::Fr::from_bytes(data + 32 * i);
        if (dir == 0) dom.fft(a, threads); else if (dir == 1) dom.ifft(a, threads);
        else if (dir == 2) dom.coset_fft(a, threads); else dom.coset_ifft(a, threads);
        for (size_t i = 0; i < N; ++i) a[i].to_bytes(data + 32 * i);
    });
    return 0;
}
// group: 1 = G1, 2 = G2.  bases: ark uncompressed affine; scalars: 32 B canonical LE.
// out: coords canonical LE (G1: 2 x nb, G2: 4 x nb) followed by one infinity-flag byte.
int orc_msm(int curve, int group, uint64_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out, int threads) {
    DISPATCH(curve, {
        typedef C::Fr Fr;
        std::vector<u64> k(n * Fr::N);
        memcpy(k.data(), scalars, n * 32);
        if (group == 1) {
            typedef Affine<C::Fq> A;
            std::vector<A> b(n);
            for (u64 i = 0; i < n; ++i) b[i] = A::from_bytes(bases + i * A::BYTES);
            A res = msm<C::Fq, Fr::N>(b.data(), k.data(), n, Fr::P().bits, threads).to_affine();
            memset(out, 0, A::BYTES + 1);
            if (!res.inf) { res.x.to_bytes(out); res.y.to_bytes(out + C::Fq::BYTES); }
            out[A::BYTES] = res.inf;
        } else {
            typedef Affine<C::Fq2> A;
            std::vector<A> b(n);
            for (u64 i = 0; i < n; ++i) b[i] = A::from_bytes(bases + i * A::BYTES);
            A res = msm<C::Fq2, Fr::N>(b.data(), k.data(), n, Fr::P().bits, threads).to_affine();
            memset(out, 0, A::BYTES + 1);
            if (!res.inf) { res.x.to_bytes(out); res.y.to_bytes(out + C::Fq2::BYTES); }
            out[A::BYTES] = res.inf;
        }
    });
    return 0;
}
// op: 0 add 1 sub 2 mul 3 inverse(a) ; field: 0 Fr 1 Fq ; canonical LE in/out
int orc_field_op(int curve, int field, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    DISPATCH(curve, {
        auto run = [&](auto tag) {
            typedef decltype(tag) F;
            F x = F::from_bytes(a), y = F::from_bytes(b), r;
            r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : x.inverse();
            r.to_bytes(out);
        };
        if (field == 0) run(C::Fr{}); else run(C::Fq{});
    });
    return 0;
}
// generators, ark uncompressed
void orc_generators(int curve, uint8_t* g1, uint8_t* g2) { DISPATCH(curve, { C::g1().to_bytes(g1); C::g2().to_bytes(g2); }); }
int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

// ---------------- GM17 (oracle/c/gm17.hpp) ----------------
void* orc_gm17_setup(void* h, const uint8_t* toxic /* alpha,beta,gamma,t: 4 x 32 B */, int threads) {
    PkBase* pk = nullptr;
    int curve = ((CircuitBase*)h)->curve;
    DISPATCH(curve, {
        ToxicGm17<C> tx{C::Fr::from_bytes(toxic), C::Fr::from_bytes(toxic + 32), C::Fr::from_bytes(toxic + 64), C::Fr::from_bytes(toxic + 96)};
        pk = setup_gm17<C>(*(Circuit<C>*)h, tx, threads);
    });
    pk->curve = curve;
    return pk;
}
void* orc_gm17_pk_parse(int curve, const uint8_t* data, uint64_t len) {
    PkBase* out = nullptr;
    DISPATCH(curve, {
        auto* pk = new PkGm17<C>();
        if (pk->parse(data, len)) out = pk; else delete pk;
    });
    if (out) out->curve = curve;
    return out;
}
uint64_t orc_gm17_pk_size(void* h) { uint64_t s = 0; DISPATCH(((PkBase*)h)->curve, s = ((PkGm17<C>*)h)->byte_size()); return s; }
void orc_gm17_pk_serialize(void* h, uint8_t* out) { DISPATCH(((PkBase*)h)->curve, ((PkGm17<C>*)h)->serialize(out)); }
// timings: 8 doubles (-, witness map, G msm, C1+C2 msm, A msm, -, B msm, total) or NULL
int orc_gm17_prove(void* ch, void* pkh, const uint8_t* z, const uint8_t* d1, const uint8_t* d2, const uint8_t* r, uint8_t* proof_raw, int threads,
                   double* timings) {
    int rc = -1;
    DISPATCH(((CircuitBase*)ch)->curve, {
        auto* c = (Circuit<C>*)ch;
        std::vector<C::Fr> zz(c->m());
        for (size_t i = 0; i < zz.size(); ++i) zz[i] = C::Fr::from_bytes(z + 32 * i);
        ProofT<C> p;
        Timings tm{};
        rc = prove_gm17<C>(*c, *(PkGm17<C>*)pkh, zz, C::Fr::from_bytes(d1), C::Fr::from_bytes(d2), C::Fr::from_bytes(r), p, threads, &tm);
        if (rc == 0) proof_to_raw<C>(p, proof_raw);
        if (timings) memcpy(timings, &tm, sizeof(tm));
    });
    return rc;
}
int orc_gm17_trapdoor(void* ch, const uint8_t* toxic, const uint8_t* z, const uint8_t* d1, const uint8_t* r, uint8_t* proof_raw, int threads) {
    DISPATCH(((CircuitBase*)ch)->curve, {
        auto* c = (Circuit<C>*)ch;
        std::vector<C::Fr> zz(c->m());
        for (size_t i = 0; i < zz.size(); ++i) zz[i] = C::Fr::from_bytes(z + 32 * i);
        ToxicGm17<C> tx{C::Fr::from_bytes(toxic), C::Fr::from_bytes(toxic + 32), C::Fr::from_bytes(toxic + 64), C::Fr::from_bytes(toxic + 96)};
        ProofT<C> p;
        trapdoor_gm17<C>(*c, tx, zz, C::Fr::from_bytes(d1), C::Fr::from_bytes(r), p, threads);
        proof_to_raw<C>(p, proof_raw);
    });
    return 0;
}

}  // extern "C"
