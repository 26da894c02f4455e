// backend.cpp — zkhip_backend.hpp: the compiled host side of the `hip` backend (plain C++17 over the C ABI of zkhip.h; no
// HIP, no internal headers).  What it restates and where the reference does it:
//   * generate_proof        /root/reference/zokrates_ark/src/groth16.rs:20-53, gm17.rs:43-78 (inputs, key, prove, points)
//   * the blinding scalars  ark_groth16::create_random_proof draws `Fr::rand(rng)` twice (r, s), ark_gm17 three times
//                           (d1, d2, r), first thing, from the caller's RNG
//   * get_rng_from_entropy  /root/reference/zokrates_proof_systems/src/rng.rs:5-20 (Blake2b-512, first 32 bytes -> StdRng)
//   * StdRng                rand 0.8.5 = rand_chacha 0.3.1 ChaCha12Rng ([UPSTREAM]; known answers in tests/)
//   * Fr::rand              ark-ff 0.3.0 `impl Distribution<Fp256<P>> for Standard` ([UPSTREAM])
//   * proof points -> hex   parse_g1 / parse_g2 / parse_fr, /root/reference/zokrates_ark/src/lib.rs:150-226
//   * proof.json            TaggedProof + serde_json::to_string_pretty, /root/reference/zokrates_proof_systems/src/tagged.rs:14-37
#include "../../../include/zkhip_backend.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>

namespace zokrates_hip {

// ------------------------------------------------------------------ BLAKE2b-512 (RFC 7693), unkeyed
namespace {
const uint64_t B2_IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                           0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8);   // little-endian host
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= t;                                                      // (messages here are far below 2^64 bytes: t1 = 0)
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] = v[a] + v[b] + x; v[d] = rotr64(v[d] ^ v[a], 32);
        v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 24);
        v[a] = v[a] + v[b] + y; v[d] = rotr64(v[d] ^ v[a], 16);
        v[c] = v[c] + v[d];     v[b] = rotr64(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = B2_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
}  // namespace

std::array<uint8_t, 64> blake2b_512(const uint8_t* data, size_t len) {
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = B2_IV[i];
    h[0] ^= 0x01010040ull;                                           // digest length 64, no key, fanout = depth = 1
    uint8_t block[128];
    size_t off = 0;
    while (len - off > 128) {                                        // every block but the last
        b2_compress(h, data + off, (uint64_t)off + 128, false);
        off += 128;
    }
    memset(block, 0, 128);
    if (len > off) memcpy(block, data + off, len - off);
    b2_compress(h, block, (uint64_t)len, true);
    std::array<uint8_t, 64> out;
    memcpy(out.data(), h, 64);
    return out;
}

// ------------------------------------------------------------------ StdRng = ChaCha12
namespace {
inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    const uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                               (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    memcpy(x, init, sizeof(x));
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < rounds / 2; ++r) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + init[i];
}
}  // namespace

StdRng::StdRng(const std::array<uint8_t, 32>& seed) { memcpy(key_, seed.data(), 32); }
StdRng StdRng::from_os_entropy() {
    std::array<uint8_t, 32> seed{};
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(seed.data(), 1, 32, f) != 32) {
        if (f) fclose(f);
        throw Error(ZKHIP_ERR_DEVICE, "cannot read /dev/urandom");
    }
    fclose(f);
    return StdRng(seed);
}
uint32_t StdRng::next_u32() {
    if (index_ >= 16) {
        chacha_block(key_, counter_++, 0, 12, block_);
        index_ = 0;
    }
    return block_[index_++];
}
uint64_t StdRng::next_u64() {
    const uint64_t lo = next_u32();
    return lo | ((uint64_t)next_u32() << 32);
}
StdRng get_rng_from_entropy(const std::string& entropy) {
    const std::array<uint8_t, 64> h = blake2b_512((const uint8_t*)entropy.data(), entropy.size());
    std::array<uint8_t, 32> seed;
    memcpy(seed.data(), h.data(), 32);
    return StdRng(seed);
}

// ------------------------------------------------------------------ Fr::rand
namespace {
struct FrParams {
    uint64_t p[4];
    int shave;            // REPR_SHAVE_BITS = 256 - modulus bits
};
const FrParams FR_BN128 = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}, 2};
const FrParams FR_BLS12_381 = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}, 1};
bool lt(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; --i)
        if (a[i] != b[i]) return a[i] < b[i];
    return false;
}
// x * 2^-256 mod p (Montgomery reduction of a 256-bit value: the limbs ark samples ARE the Montgomery representation)
void from_mont(const uint64_t x[4], const FrParams& f, uint64_t out[4]) {
    uint64_t inv = 1;                                             // -p^-1 mod 2^64 by Newton iteration
    for (int i = 0; i < 6; ++i) inv *= 2 - f.p[0] * inv;
    inv = (uint64_t)0 - inv;
    uint64_t t[5] = {x[0], x[1], x[2], x[3], 0};
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[0] * inv;
        unsigned __int128 c = (unsigned __int128)m * f.p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (unsigned __int128)m * f.p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = (uint64_t)(c >> 64);
    }
    if (t[4] || !lt(t, f.p)) {                                    // one conditional subtraction
        unsigned __int128 b = 0;
        for (int j = 0; j < 4; ++j) {
            const unsigned __int128 d = (unsigned __int128)t[j] - f.p[j] - (uint64_t)b;
            t[j] = (uint64_t)d;
            b = (d >> 64) & 1;
        }
    }
    memcpy(out, t, 32);
}
}  // namespace

std::array<uint8_t, 32> fr_rand(StdRng& rng, int32_t curve) {
    if (curve != ZKHIP_CURVE_BN128 && curve != ZKHIP_CURVE_BLS12_381) throw Error(ZKHIP_ERR_BAD_ARG, "unknown curve id");
    const FrParams& f = curve == ZKHIP_CURVE_BN128 ? FR_BN128 : FR_BLS12_381;
    for (;;) {
        uint64_t limbs[4];
        for (int i = 0; i < 4; ++i) limbs[i] = rng.next_u64();
        limbs[3] &= ~(uint64_t)0 >> f.shave;
        if (!lt(limbs, f.p)) continue;
        uint64_t canon[4];
        from_mont(limbs, f, canon);
        std::array<uint8_t, 32> out;
        memcpy(out.data(), canon, 32);
        return out;
    }
}

// ------------------------------------------------------------------ handles
Key::~Key() { if (pk_) zkhip_pk_free(pk_); }
Key& Key::operator=(Key&& o) noexcept {
    if (this != &o) {
        if (pk_) zkhip_pk_free(pk_);
        pk_ = o.pk_;
        o.pk_ = nullptr;
    }
    return *this;
}
Program::Program(const uint8_t* bytes, size_t len) {
    const int32_t rc = zkhip_prog_parse(bytes, len, &prog_);
    if (rc != ZKHIP_OK) throw Error(rc, zkhip_last_error(nullptr));
    uint64_t d[8];
    zkhip_prog_dims(prog_, d);
    curve_ = (int32_t)d[0]; n_ = d[1]; l_ = d[2]; w_ = d[3];
}
Program::~Program() { if (prog_) zkhip_prog_free(prog_); }

Hip::Hip(int32_t device) {
    const int32_t rc = zkhip_ctx_create(device, &ctx_);
    if (rc != ZKHIP_OK) throw Error(rc, zkhip_last_error(nullptr));
}
Hip::~Hip() { if (ctx_) zkhip_ctx_free(ctx_); }
void Hip::init(int32_t hw_queues) {
    const int32_t rc = zkhip_init(hw_queues);
    if (rc != ZKHIP_OK) throw Error(rc, zkhip_last_error(nullptr));
}
// One proof, then the process ends: no window-multiple tables (ten times what they would save one proof), and every kernel on the
// context's one stream — the ~20 streams a resident prover overlaps its proofs on cost ~10 ms of queue set-up each, ~90 ms of a
// process whose proof takes 25 (profiles/r5_cli_start_profile.txt).
void Hip::one_shot() {
    check(zkhip_ctx_tune(ctx_, ZKHIP_TUNE_MSM_SETS, 64));
    check(zkhip_ctx_tune(ctx_, ZKHIP_TUNE_SERIAL, 1));
}
// A long-lived prover of LARGE DENSE circuits (accumulation-bound: 2^20 constraints and more): all sixteen streams of the context made
// in one go at its first proof, every accumulation lane type on a hardware dispatcher of its own and the fold chains on the fourth
// (core.cuh make_pipe_streams).  Measured per workload (profiles/r6t_*, r6w_*): +1-3 % proofs/s and a lone proof 0.25-0.4 ms sooner
// there, 5-15 % slower on thin circuits (SHA-256, Poseidon on BLS12-381, GM17) — so it is a request, never a default.
void Hip::separate_dispatchers() { check(zkhip_ctx_tune(ctx_, ZKHIP_TUNE_PIPE_PLAN, 1)); }
void Hip::check(int32_t rc) const {
    if (rc != ZKHIP_OK) throw Error(rc, zkhip_last_error(ctx_));
}
std::string Hip::describe() const {
    char buf[256];
    check(zkhip_describe(ctx_, buf, sizeof(buf)));
    return buf;
}
Key Hip::load_proving_key(Scheme scheme, int32_t curve, const uint8_t* bytes, size_t len) {
    Key k;
    check(scheme == Scheme::GM17 ? zkhip_pk_load_gm17(ctx_, curve, bytes, len, &k.pk_) : zkhip_pk_load_g16(ctx_, curve, bytes, len, &k.pk_));
    return k;
}
Key Hip::import_key_image(const uint8_t* bytes, size_t len) {
    Key k;
    check(zkhip_pk_import(ctx_, bytes, len, &k.pk_));
    return k;
}
std::vector<uint8_t> Hip::export_key_image(const Key& key) const {
    uint64_t size = 0;
    check(zkhip_pk_export_size(key.get(), &size));
    std::vector<uint8_t> out(size);
    check(zkhip_pk_export(key.get(), out.data(), size));
    return out;
}

// ------------------------------------------------------------------ proof points and JSON
namespace {
std::string hex_be(const uint8_t* le, size_t n) {
    static const char* D = "0123456789abcdef";
    std::string s = "0x";
    s.reserve(2 + 2 * n);
    for (size_t i = n; i-- > 0;) { s.push_back(D[le[i] >> 4]); s.push_back(D[le[i] & 15]); }
    return s;
}
double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

System::~System() { if (cs_) zkhip_r1cs_free(cs_); }
System& System::operator=(System&& o) noexcept {
    if (this != &o) {
        if (cs_) zkhip_r1cs_free(cs_);
        cs_ = o.cs_; prog_ = o.prog_;
        o.cs_ = nullptr; o.prog_ = nullptr;
    }
    return *this;
}
System Hip::load_system(const Program& program) {
    System s;
    check(zkhip_prog_r1cs_load(ctx_, program.get(), &s.cs_));
    s.prog_ = &program;
    return s;
}
bool Hip::bind(Key& key, const System& system) {
    const int32_t rc = zkhip_pk_bind_r1cs(ctx_, key.get(), system.get());
    if (rc == ZKHIP_ERR_NOMEM) return false;
    check(rc);
    return true;
}
bool Hip::is_bound(const Key& key, const System& system) const { return zkhip_pk_is_bound(key.get(), system.get()) == 1; }

Proof Hip::prove(Scheme scheme, const Program& program, const uint8_t* witness, size_t witness_len, const Key& key, StdRng& rng, Timings* tm) {
    // the constraint system on the device, for this one proof
    auto t0 = std::chrono::steady_clock::now();
    const System system = load_system(program);
    const double upload = ms_since(t0);
    Proof p = prove(scheme, system, witness, witness_len, key, rng, tm);
    if (tm) tm->r1cs_upload = upload;
    return p;
}

Proof Hip::prove(Scheme scheme, const System& system, const uint8_t* witness, size_t witness_len, const Key& key, StdRng& rng, Timings* tm) {
    const Program& program = system.program();
    const int32_t curve = program.curve();
    const size_t fq = curve == ZKHIP_CURVE_BN128 ? 32 : 48;
    // (1) the assignment in ark order and the public inputs as the ark backend computes them (groth16.rs:33-38)
    auto t0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> z(program.variables() * 32);
    const uint64_t cap = witness_len / 40 + 2;
    std::vector<uint8_t> inputs(cap * 32);
    uint64_t n_inputs = 0;
    int32_t rc = zkhip_prog_assignment(program.get(), witness, witness_len, z.data(), inputs.data(), cap, &n_inputs);
    if (rc != ZKHIP_OK) throw Error(rc, zkhip_last_error(nullptr));
    if (tm) tm->witness_to_assignment = ms_since(t0);
    // (2) the constraint system: resident
    zkhip_r1cs* cs = system.get();
    if (tm) tm->r1cs_upload = 0;
    // (3) the blinding scalars: the first draws ark makes from the caller's RNG; (4) the GPU
    t0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> raw(8 * fq + 3);
    if (scheme == Scheme::GM17) {
        uint8_t rnd[96];
        for (int i = 0; i < 3; ++i) memcpy(rnd + 32 * i, fr_rand(rng, curve).data(), 32);      // d1, d2, r
        rc = zkhip_prove_gm17(ctx_, key.get(), cs, z.data(), rnd, raw.data(), nullptr);
    } else {
        const std::array<uint8_t, 32> r = fr_rand(rng, curve), s = fr_rand(rng, curve);
        rc = zkhip_prove_g16(ctx_, key.get(), cs, z.data(), r.data(), s.data(), raw.data(), nullptr);
    }
    check(rc);
    if (tm) tm->prove = ms_since(t0);
    // (5) raw little-endian coordinates -> big-endian hex; a point at infinity is printed as ark's zero() = (0, 1)
    std::vector<uint8_t> one(fq, 0);
    one[0] = 1;
    if (raw[8 * fq]) memcpy(&raw[fq], one.data(), fq);
    if (raw[8 * fq + 1]) memcpy(&raw[4 * fq], one.data(), fq);
    if (raw[8 * fq + 2]) memcpy(&raw[7 * fq], one.data(), fq);
    Proof p;
    p.scheme = scheme == Scheme::GM17 ? "gm17" : "g16";
    p.curve = curve == ZKHIP_CURVE_BN128 ? "bn128" : "bls12_381";
    p.proof.a = {hex_be(&raw[0], fq), hex_be(&raw[fq], fq)};
    p.proof.b.x = {hex_be(&raw[2 * fq], fq), hex_be(&raw[3 * fq], fq)};
    p.proof.b.y = {hex_be(&raw[4 * fq], fq), hex_be(&raw[5 * fq], fq)};
    p.proof.c = {hex_be(&raw[6 * fq], fq), hex_be(&raw[7 * fq], fq)};
    for (uint64_t i = 0; i < n_inputs; ++i) p.inputs.push_back(hex_be(&inputs[32 * i], 32));
    return p;
}

// ------------------------------------------------------------------ setup
namespace {
std::string json_g1(const uint8_t* p, size_t fq, const std::string& ind) {
    std::vector<uint8_t> t(p, p + 2 * fq);
    t[2 * fq - 1] &= 0x3f;                                   // ark's flag bits live in the last byte
    return "[\n" + ind + "  \"" + hex_be(&t[0], fq) + "\",\n" + ind + "  \"" + hex_be(&t[fq], fq) + "\"\n" + ind + "]";
}
std::string json_g2(const uint8_t* p, size_t fq, const std::string& ind) {
    std::vector<uint8_t> t(p, p + 4 * fq);
    t[4 * fq - 1] &= 0x3f;
    auto pair = [&](size_t k) {
        return "[\n" + ind + "    \"" + hex_be(&t[k * fq], fq) + "\",\n" + ind + "    \"" + hex_be(&t[(k + 1) * fq], fq) + "\"\n" + ind + "  ]";
    };
    return "[\n" + ind + "  " + pair(0) + ",\n" + ind + "  " + pair(2) + "\n" + ind + "]";
}
std::string json_g1_list(const uint8_t* p, uint64_t count, size_t fq) {
    if (!count) return "[]";
    std::string s = "[\n";
    for (uint64_t i = 0; i < count; ++i) s += "    " + json_g1(p + i * 2 * fq, fq, "    ") + (i + 1 < count ? ",\n" : "\n");
    return s + "  ]";
}
}  // namespace

SetupKeypair Hip::setup(Scheme scheme, const Program& program, StdRng& rng) {
    const int32_t curve = program.curve();
    std::array<uint8_t, 32> tox[5];
    for (int k = 0; k < 5;) {                                   // alpha, beta, gamma, delta, tau: redrawn while zero
        tox[k] = fr_rand(rng, curve);
        bool zero = true;
        for (uint8_t b : tox[k]) zero = zero && b == 0;
        if (!zero) ++k;
    }
    zkhip_r1cs* cs = nullptr;
    check(zkhip_prog_r1cs_load(ctx_, program.get(), &cs));
    SetupKeypair kp;
    uint64_t size = 0;
    int32_t rc;
    if (scheme == Scheme::GM17) {
        uint8_t t4[128];
        memcpy(t4, tox[0].data(), 32); memcpy(t4 + 32, tox[1].data(), 32);
        memset(t4 + 64, 0, 32); t4[64] = 1;                     // ark-gm17's generate_random_parameters: gamma = 1
        memcpy(t4 + 96, tox[4].data(), 32);
        zkhip_setup_gm17_size(cs, &size);
        kp.pk.resize(size);
        rc = zkhip_setup_gm17(ctx_, cs, t4, nullptr, nullptr, kp.pk.data(), size);
    } else {
        uint8_t t5[160];
        for (int k = 0; k < 5; ++k) memcpy(t5 + 32 * k, tox[k].data(), 32);
        zkhip_setup_g16_size(cs, &size);
        kp.pk.resize(size);
        rc = zkhip_setup_g16(ctx_, cs, t5, nullptr, nullptr, kp.pk.data(), size);
    }
    zkhip_r1cs_free(cs);
    check(rc);
    kp.vk = verification_key_json(scheme, curve, kp.pk.data(), kp.pk.size());
    return kp;
}

// The verification key sits at the head of ark's proving key (ProvingKey { vk, .. }: serialize_unchecked writes it first):
// g16: alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1[]; gm17: h_g2, g_alpha_g1, h_beta_g2, g_gamma_g1, h_gamma_g2, query[].
std::string verification_key_json(Scheme scheme, int32_t curve, const uint8_t* pk, size_t len) {
    if (curve != ZKHIP_CURVE_BN128 && curve != ZKHIP_CURVE_BLS12_381) throw Error(ZKHIP_ERR_BAD_ARG, "verification_key_json: unsupported curve");
    const size_t fq = curve == ZKHIP_CURVE_BN128 ? 32 : 48, g1 = 2 * fq, g2 = 4 * fq;
    const size_t fixed = scheme == Scheme::GM17 ? 2 * g1 + 3 * g2 : g1 + 3 * g2;
    if (!pk || len < fixed + 8) throw Error(ZKHIP_ERR_PARSE, "proving key too short for its verification key");
    uint64_t count;
    memcpy(&count, pk + fixed, 8);
    if (count > (len - fixed - 8) / g1) throw Error(ZKHIP_ERR_PARSE, "proving key too short for its verification key");
    const std::string cname = curve == ZKHIP_CURVE_BN128 ? "bn128" : "bls12_381";
    std::string s = "{\n";
    if (scheme == Scheme::GM17) {      // (scheme/gm17.rs:19-27)
        size_t pos = 0;
        const uint8_t *h = pk; pos += g2;
        const uint8_t* g_alpha = pk + pos; pos += g1;
        const uint8_t* h_beta = pk + pos; pos += g2;
        const uint8_t* g_gamma = pk + pos; pos += g1;
        const uint8_t* h_gamma = pk + pos; pos += g2;
        s += "  \"scheme\": \"gm17\",\n  \"curve\": \"" + cname + "\",\n";
        s += "  \"h\": " + json_g2(h, fq, "  ") + ",\n  \"g_alpha\": " + json_g1(g_alpha, fq, "  ") + ",\n  \"h_beta\": " + json_g2(h_beta, fq, "  ") + ",\n";
        s += "  \"g_gamma\": " + json_g1(g_gamma, fq, "  ") + ",\n  \"h_gamma\": " + json_g2(h_gamma, fq, "  ") + ",\n";
        s += "  \"query\": " + json_g1_list(pk + pos + 8, count, fq) + "\n}";
    } else {                           // (scheme/groth16.rs:18-25)
        s += "  \"scheme\": \"g16\",\n  \"curve\": \"" + cname + "\",\n";
        s += "  \"alpha\": " + json_g1(pk, fq, "  ") + ",\n  \"beta\": " + json_g2(pk + g1, fq, "  ") + ",\n";
        s += "  \"gamma\": " + json_g2(pk + g1 + g2, fq, "  ") + ",\n  \"delta\": " + json_g2(pk + g1 + 2 * g2, fq, "  ") + ",\n";
        s += "  \"gamma_abc\": " + json_g1_list(pk + fixed + 8, count, fq) + "\n}";
    }
    return s;
}


Proof Hip::generate_proof(Scheme scheme, const uint8_t* program, size_t program_len, const uint8_t* witness, size_t witness_len,
                          const uint8_t* proving_key, size_t proving_key_len, StdRng& rng) {
    Program prog(program, program_len);
    Key key = load_proving_key(scheme, prog.curve(), proving_key, proving_key_len);
    return prove(scheme, prog, witness, witness_len, key, rng);
}

std::string Proof::to_json() const {
    std::string s;
    auto str = [](const std::string& v) { return "\"" + v + "\""; };
    auto g1 = [&](const G1Affine& p, const std::string& ind) {
        return "[\n" + ind + "  " + str(p.x) + ",\n" + ind + "  " + str(p.y) + "\n" + ind + "]";
    };
    auto pair = [&](const std::array<std::string, 2>& c, const std::string& ind) {
        return "[\n" + ind + "  " + str(c[0]) + ",\n" + ind + "  " + str(c[1]) + "\n" + ind + "]";
    };
    s += "{\n";
    s += "  \"scheme\": " + str(scheme) + ",\n";
    s += "  \"curve\": " + str(curve) + ",\n";
    s += "  \"proof\": {\n";
    s += "    \"a\": " + g1(proof.a, "    ") + ",\n";
    s += "    \"b\": [\n      " + pair(proof.b.x, "      ") + ",\n      " + pair(proof.b.y, "      ") + "\n    ],\n";
    s += "    \"c\": " + g1(proof.c, "    ") + "\n";
    s += "  },\n";
    if (inputs.empty()) {
        s += "  \"inputs\": []\n";
    } else {
        s += "  \"inputs\": [\n";
        for (size_t i = 0; i < inputs.size(); ++i) s += "    " + str(inputs[i]) + (i + 1 < inputs.size() ? ",\n" : "\n");
        s += "  ]\n";
    }
    s += "}";
    return s;
}

}  // namespace zokrates_hip
