// zkhip_backend.hpp — the host side of the `hip` proving backend in C++, above the C ABI of zkhip.h.
//
// The reference's host side is Rust: `impl Backend<T, G16> for Ark` / `impl Backend<T, GM17> for Ark`
// (/root/reference/zokrates_ark/src/groth16.rs:20-53, gm17.rs:43-78) behind the trait of
// /root/reference/zokrates_proof_systems/src/lib.rs:98-112, driven by `zokrates generate-proof`
// (/root/reference/zokrates_cli/src/ops/generate_proof.rs:95-202).  No Rust toolchain exists in this image, so the
// compiled host layer is C++ with the same names, argument meaning and error behaviour:
//
//   reference                                              here
//   Backend::generate_proof(program, witness, pk, rng)     Hip::generate_proof(scheme, program, witness, pk, rng)
//   Proof { proof: ProofPoints { a, b, c }, inputs }       Proof / ProofPoints / G1Affine / G2Affine   (lib.rs:33-96, scheme/groth16.rs:8-16)
//   TaggedProof -> serde_json::to_string_pretty            Proof::to_json()                            (tagged.rs:14-37)
//   Backend::verify(vk, proof) -> bool                     verify(VerificationKey, Proof) -> bool      (groth16.rs:55-87, gm17.rs:69-110; host CPU)
//   get_rng_from_entropy(&str) -> StdRng                   get_rng_from_entropy(std::string) -> StdRng (rng.rs:5-20)
//   StdRng::from_entropy()                                 StdRng::from_os_entropy()
//   panic!(..) on any failure (groth16.rs:41-44 unwrap)    throws zokrates_hip::Error (code = ZKHIP_ERR_*, message of the library)
//
// `program` and `witness` are the bytes of ZoKrates' own files (`out`, `witness`): the walk of
// `Computation::generate_constraints` (zokrates_ark/src/lib.rs:80-129) and `public_inputs_values` happen inside
// zkhip_prog_parse / zkhip_prog_assignment.  The Rust adapter that would sit behind the real trait is
// integration/zokrates_hip (source only); this layer is what is compiled, tested and shipped here (tools: zkhip-cli).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "zkhip.h"

namespace zokrates_hip {

enum class Scheme { G16, GM17 };

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// "0x" + lower-case hex, big-endian, zero-padded to the base-field width (parse_g1 / parse_g2: zokrates_ark/src/lib.rs:150-218)
struct G1Affine { std::string x, y; };
struct G2Affine { std::array<std::string, 2> x, y; };   // (c0, c1)
struct ProofPoints { G1Affine a; G2Affine b; G1Affine c; };
struct Proof {
    std::string scheme, curve;            // the tags of TaggedProof
    ProofPoints proof;
    std::vector<std::string> inputs;      // public_inputs_values as 32-byte big-endian hex (parse_fr, lib.rs:220-226)
    std::string to_json() const;          // serde_json::to_string_pretty of the tagged proof: the text of proof.json
    static Proof from_json(const std::string& text);      // serde_json::from_value::<Proof<T, S>> (ops/verify.rs:181-182); Error on a malformed file
    // `zokrates print-proof --format json|remix` (ops/print_proof.rs:85-114): the line to paste into the Solidity verifier; bn128 only
    std::string print(const std::string& format) const;
};

// `verification.key` (scheme/groth16.rs:18-25: alpha, beta, gamma, delta, gamma_abc; scheme/gm17.rs:19-27: h, g_alpha, h_beta,
// g_gamma, h_gamma, query), points as in the file, looked up by the file's field names
struct VerificationKey {
    std::string scheme, curve;
    std::map<std::string, G1Affine> g1;
    std::map<std::string, G2Affine> g2;
    std::vector<G1Affine> query;          // gamma_abc (g16) / query (gm17)
    static VerificationKey from_json(const std::string& text);
};

// Backend<T, S>::verify(vk, proof) -> bool (zokrates_ark/src/groth16.rs:55-87, gm17.rs:69-110): the pairing check, on the host
// CPU as in the reference (no GPU, no context; csrc/host/verify.cpp).  bn128, bls12_381 and — for the reference's own golden proofs —
// bls12_377; g16 and gm17.  false: the equation
// does not hold, or a point is off its curve / outside the r-torsion.  Error: curve or scheme of the two files differ (the CLI's
// messages, ops/verify.rs:95-107), a coordinate or input is not canonical, the input count does not fit the key (the reference
// panics through `unwrap` there).
bool verify(const VerificationKey& vk, const Proof& proof);
// the text of `verification.key` for the key at the head of an ark proving key (ProvingKey { vk, .. }); Error if the bytes are too short
std::string verification_key_json(Scheme scheme, int32_t curve, const uint8_t* proving_key, size_t len);
// prod_i e(g1_i, g2_i) == 1 in the target group (the Solidity verifier's `pairing` precompile call, solidity.rs:  pairingProd*)
bool pairing_product_is_one(const std::string& curve, const std::vector<std::pair<G1Affine, G2Affine>>& pairs);

// rand 0.8.5 `StdRng` (= rand_chacha 0.3.1 ChaCha12Rng): key = seed, 64-bit block counter, words of a block in order
class StdRng {
  public:
    explicit StdRng(const std::array<uint8_t, 32>& seed);
    static StdRng from_os_entropy();      // StdRng::from_entropy(): 32 bytes of /dev/urandom
    uint32_t next_u32();
    uint64_t next_u64();                  // two consecutive words, low first (rand_core BlockRng)

  private:
    uint32_t key_[8];
    uint64_t counter_ = 0;
    uint32_t block_[16];
    int index_ = 16;
};
// zokrates_proof_systems::rng::get_rng_from_entropy: seed = first 32 bytes of Blake2b-512(entropy)
StdRng get_rng_from_entropy(const std::string& entropy);
std::array<uint8_t, 64> blake2b_512(const uint8_t* data, size_t len);
// ark-ff 0.3.0 `Fr::rand(rng)` as 32 canonical little-endian bytes (curve: ZKHIP_CURVE_*)
std::array<uint8_t, 32> fr_rand(StdRng& rng, int32_t curve);

// A proving key resident on the GPU (ProvingKey::deserialize_unchecked + the MSM tables: zkhip_pk_load_* / zkhip_pk_import)
class Key {
  public:
    Key() = default;
    ~Key();
    Key(Key&& o) noexcept : pk_(o.pk_) { o.pk_ = nullptr; }
    Key& operator=(Key&& o) noexcept;
    Key(const Key&) = delete;
    Key& operator=(const Key&) = delete;
    zkhip_pk* get() const { return pk_; }
    explicit operator bool() const { return pk_ != nullptr; }

  private:
    friend class Hip;
    zkhip_pk* pk_ = nullptr;
};

// A compiled program (`out`) decoded into the R1CS in ark variable order (host side only: zkhip_prog_parse)
class Program {
  public:
    Program(const uint8_t* bytes, size_t len);     // ProgEnum::deserialize + Computation::generate_constraints' variable walk
    ~Program();
    Program(const Program&) = delete;
    Program& operator=(const Program&) = delete;
    int32_t curve() const { return curve_; }
    uint64_t constraints() const { return n_; }
    uint64_t variables() const { return l_ + w_; }
    zkhip_prog* get() const { return prog_; }

  private:
    zkhip_prog* prog_ = nullptr;
    int32_t curve_ = 0;
    uint64_t n_ = 0, l_ = 0, w_ = 0;
};

// A program's constraint system resident on the GPU (zkhip_prog_r1cs_load).  `Backend::generate_proof` consumes its program
// and the one-call form below uploads the system for every proof, as the reference rebuilds its ConstraintSystem for every proof
// (zokrates_ark/src/lib.rs:80-129); a caller that proves many witnesses of one program — zokrates_js calls generate_proof many
// times per process, zokrates_js/src/lib.rs:380-452 — keeps a System next to its Key and may bind the two (Hip::bind).
class System {
  public:
    System() = default;
    ~System();
    System(System&& o) noexcept : cs_(o.cs_), prog_(o.prog_) { o.cs_ = nullptr; o.prog_ = nullptr; }
    System& operator=(System&& o) noexcept;
    System(const System&) = delete;
    System& operator=(const System&) = delete;
    zkhip_r1cs* get() const { return cs_; }
    const Program& program() const { return *prog_; }     // (the Program must outlive the System)

  private:
    friend class Hip;
    zkhip_r1cs* cs_ = nullptr;
    const Program* prog_ = nullptr;
};

// NonUniversalBackend::setup's result (zokrates_proof_systems/src/lib.rs:59-65,113-118): the verification key as the text of
// `verification.key` (scheme/groth16.rs:18-25, scheme/gm17.rs:19-27) and the proving key in ark's serialize_unchecked bytes
struct SetupKeypair {
    std::string vk;
    std::vector<uint8_t> pk;
};

// where the wall clock of one proof went (milliseconds)
struct Timings {
    double witness_to_assignment = 0, r1cs_upload = 0, prove = 0;
};

// One GPU.  (The reference's backends are unit structs — `pub struct Ark;` — because they own nothing; this one owns a
// zkhip_ctx, so it is an object.  Not re-entrant, like the context.)
class Hip {
  public:
    explicit Hip(int32_t device = 0);
    ~Hip();
    // Process-wide, BEFORE the first Hip object (and before any other HIP user of the process starts the runtime): ask the HIP
    // runtime for `hw_queues` hardware queues (zkhip_init).  A resident prover keeps ~20 streams busy and wants 8 (16 if it is the
    // only context of its process); a one-proof process runs on one stream and need not call this.  The library never changes the
    // process environment on its own.
    static void init(int32_t hw_queues = 8);
    // One proof per process (what `generate-proof` is): keys loaded from now on skip the precomputed window multiples — building
    // them costs ten times what they save a single proof (0.15 s of kernels at 2^20 against ~5 ms of proof time) — and the MSMs
    // fold one bucket set per window instead (ZKHIP_TUNE_MSM_SETS).  A resident prover keeps the default.
    void one_shot();
    // A long-lived prover, before its first proof: the context's streams made in one go and placed on the GPU's four dispatchers by
    // plan (ZKHIP_TUNE_PIPE_PLAN; sixteen streams: call init(16) first, and keep the process below ~22 hardware queues in all).  Level
    // or better than streams in order of first use on every measured workload; not for one-proof processes (0.15 s).
    void separate_dispatchers();
    Hip(const Hip&) = delete;
    Hip& operator=(const Hip&) = delete;

    // Backend<T, S>::generate_proof(program, witness, proving_key, rng) -> Proof, on the bytes of the three files
    Proof generate_proof(Scheme scheme, const uint8_t* program, size_t program_len, const uint8_t* witness, size_t witness_len,
                         const uint8_t* proving_key, size_t proving_key_len, StdRng& rng);

    // the same in its parts, for callers that keep keys resident, cache their device layout or overlap the steps
    Key load_proving_key(Scheme scheme, int32_t curve, const uint8_t* bytes, size_t len);
    Key import_key_image(const uint8_t* bytes, size_t len);            // zkhip_pk_import
    std::vector<uint8_t> export_key_image(const Key& key) const;      // zkhip_pk_export (compact: level 0 of the tables)
    Proof prove(Scheme scheme, const Program& program, const uint8_t* witness, size_t witness_len, const Key& key, StdRng& rng,
                Timings* timings = nullptr);
    // a long-lived prover: the constraint system uploaded once, the key (Groth16 or GM17) bound to it (zkhip_pk_bind_r1cs: the quotient's
    // inverse transforms applied to the key's bases once, four transforms per proof afterwards, the same proof).  bind returns
    // false when the device has no room for the two extra tables: the key then proves as it was loaded.
    System load_system(const Program& program);
    bool bind(Key& key, const System& system);
    bool is_bound(const Key& key, const System& system) const;
    Proof prove(Scheme scheme, const System& system, const uint8_t* witness, size_t witness_len, const Key& key, StdRng& rng,
                Timings* timings = nullptr);
    // NonUniversalBackend<T, S>::setup(program, rng) -> SetupKeypair: toxic waste = five non-zero `Fr::rand` draws (alpha, beta,
    // gamma, delta, tau; GM17: gamma = 1 as in ark-gm17), standard group generators (ark samples random ones from the RNG: a
    // key made here is valid, not byte-equal to `zokrates setup --entropy`'s), key generation on the GPU (zkhip_setup_*)
    SetupKeypair setup(Scheme scheme, const Program& program, StdRng& rng);
    std::string describe() const;

  private:
    zkhip_ctx* ctx_ = nullptr;
    void check(int32_t rc) const;
};

}  // namespace zokrates_hip
