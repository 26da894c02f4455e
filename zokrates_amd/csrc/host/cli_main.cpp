// zkhip-cli — `zokrates generate-proof` for the hip backend, as a native executable over zkhip_backend.hpp.
//
//   zkhip-cli generate-proof -i out -w witness -p proving.key -j proof.json [-s g16|gm17] [--entropy TEXT]
//                            [--key-cache DIR] [--device N] [--timings] [--verify]
//
// Mirrors /root/reference/zokrates_cli/src/ops/generate_proof.rs:95-202: the compiled program (`out`), the witness and the
// proving key are read from files, the proof is written as JSON, one proof per process; `--entropy` seeds the RNG as
// `get_rng_from_entropy` does, otherwise the OS does (`StdRng::from_entropy`).  Failures print the message and exit 1 (the
// reference's panic hook and `exit(1)`: zokrates_cli/src/bin.rs:18-25,90-105).
// The three inputs are independent until the proof starts: the program is read and decoded on host threads (zkhip_prog_parse
// cuts the constraint section into parallel chunks) WHILE the key is uploaded and its window multiples are built on the GPU.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <thread>

#include "../../../include/zkhip_backend.hpp"

using namespace zokrates_hip;

namespace {
// a file mapped read-only (the page cache is the buffer: nothing is copied, nothing is zero-filled first)
struct Mapped {
    const uint8_t* data = nullptr;
    size_t size = 0;
    Mapped() = default;
    explicit Mapped(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error(ZKHIP_ERR_BAD_ARG, "cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); throw Error(ZKHIP_ERR_BAD_ARG, "cannot stat " + path); }
        size = (size_t)st.st_size;
        if (size) {
            void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (p == MAP_FAILED) { close(fd); throw Error(ZKHIP_ERR_NOMEM, "cannot map " + path); }
            data = (const uint8_t*)p;
        }
        close(fd);
    }
    Mapped(Mapped&& o) noexcept : data(o.data), size(o.size) { o.data = nullptr; o.size = 0; }
    Mapped& operator=(Mapped&& o) noexcept { std::swap(data, o.data); std::swap(size, o.size); return *this; }
    Mapped(const Mapped&) = delete;
    Mapped& operator=(const Mapped&) = delete;
    ~Mapped() { if (data) munmap((void*)data, size); }
};
double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
// the CPU this thread runs on and its NUMA node (for the record in --timings: the host-side stages — reading, decoding, the
// staging copies — depend on where the process sits relative to the GPU); -1 where the kernel does not say
std::pair<int, int> cpu_and_node() {
    unsigned cpu = 0, node = 0;
#ifdef SYS_getcpu
    if (syscall(SYS_getcpu, &cpu, &node, nullptr) == 0) return {(int)cpu, (int)node};
#endif
    return {-1, -1};
}
int usage() {
    fprintf(stderr, "usage: zkhip-cli generate-proof -i <out> -w <witness> -p <proving.key> -j <proof.json> [-s g16|gm17] [--entropy TEXT] "
                    "[--key-cache DIR] [--device N] [--timings] [--verify] [--full-tables]\n"
                    "       zkhip-cli setup -i <out> -p <proving.key> -v <verification.key> [-s g16|gm17] [--entropy TEXT] [--device N]\n"
                    "       zkhip-cli verify [-v <verification.key>] [-j <proof.json>]\n"
                    "       zkhip-cli print-proof [-j <proof.json>] [-f remix|json]\n");
    return 2;
}
}  // namespace

// zkhip-cli setup -i out -p proving.key -v verification.key [-s g16|gm17] [--entropy TEXT] [--device N]
// (/root/reference/zokrates_cli/src/ops/setup.rs: program in, proving.key + verification.key out)
int cmd_setup(int argc, char** argv) {
    std::string input = "out", pk_path = "proving.key", vk_path = "verification.key", scheme_s = "g16", entropy;
    bool have_entropy = false;
    int device = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-i" || a == "--input") input = val();
        else if (a == "-p" || a == "--proving-key-path") pk_path = val();
        else if (a == "-v" || a == "--verification-key-path") vk_path = val();
        else if (a == "-s" || a == "--proving-scheme") scheme_s = val();
        else if (a == "--entropy") { entropy = val(); have_entropy = true; }
        else if (a == "--device") device = atoi(val().c_str());
        else return usage();
    }
    if (scheme_s != "g16" && scheme_s != "gm17") return usage();
    try {
        const Mapped prog_bytes(input);
        const Program program(prog_bytes.data, prog_bytes.size);
        Hip hip(device);
        StdRng rng = have_entropy ? get_rng_from_entropy(entropy) : StdRng::from_os_entropy();
        const SetupKeypair kp = hip.setup(scheme_s == "gm17" ? Scheme::GM17 : Scheme::G16, program, rng);
        std::ofstream o(pk_path, std::ios::binary);
        o.write((const char*)kp.pk.data(), (std::streamsize)kp.pk.size());
        std::ofstream v(vk_path);
        v << kp.vk;
        o.close();                      // (the stream state is only final once the buffers have been flushed)
        v.close();
        if (!o || !v) throw Error(ZKHIP_ERR_BAD_ARG, "cannot write the key files");
        printf("setup (%s): %llu constraints, %llu variables; wrote %s, %s\n", scheme_s.c_str(), (unsigned long long)program.constraints(),
               (unsigned long long)program.variables(), pk_path.c_str(), vk_path.c_str());
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "zkhip-cli: %s\n", e.what());
        return 1;
    }
}

static std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(ZKHIP_ERR_BAD_ARG, "Could not open " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// zkhip-cli verify -v verification.key -j proof.json      (/root/reference/zokrates_cli/src/ops/verify.rs:49-195: curve and scheme
// come from the two files and must agree; "Performing verification..." then PASSED or FAILED, exit status 0 for both; no GPU)
int cmd_verify(int argc, char** argv) {
    std::string vk_path = "verification.key", proof_path = "proof.json";
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-v" || a == "--verification-key-path") vk_path = val();
        else if (a == "-j" || a == "--proof-path") proof_path = val();
        else if (a == "-b" || a == "--backend") val();          // accepted and ignored: there is one verifier here
        else return usage();
    }
    try {
        const VerificationKey vk = VerificationKey::from_json(slurp(vk_path));
        const Proof proof = Proof::from_json(slurp(proof_path));
        printf("Performing verification...\n");
        printf("%s\n", verify(vk, proof) ? "PASSED" : "FAILED");
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}

// zkhip-cli print-proof -j proof.json -f remix|json       (zokrates_cli/src/ops/print_proof.rs)
int cmd_print_proof(int argc, char** argv) {
    std::string proof_path = "proof.json", format = "remix";
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-j" || a == "--proof-path") proof_path = val();
        else if (a == "-f" || a == "--format") format = val();
        else return usage();
    }
    try {
        fputs(Proof::from_json(slurp(proof_path)).print(format).c_str(), stdout);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}

// zkhip-cli pairing-check <curve> <file>: the file holds lines "g1x g1y g2x0 g2x1 g2y0 g2y1" (hex as in proof.json); prints ONE or
// NOT-ONE for the product of the pairings (a test hook for the pairing itself: bilinearity on the reference's MPC fixture points)
int cmd_pairing_check(int argc, char** argv) {
    if (argc != 4) return usage();
    try {
        std::ifstream f(argv[3]);
        if (!f) throw Error(ZKHIP_ERR_BAD_ARG, std::string("Could not open ") + argv[3]);
        std::vector<std::pair<G1Affine, G2Affine>> pairs;
        std::string w[6];
        while (f >> w[0] >> w[1] >> w[2] >> w[3] >> w[4] >> w[5]) pairs.push_back({G1Affine{w[0], w[1]}, G2Affine{{w[2], w[3]}, {w[4], w[5]}}});
        printf("%s\n", pairing_product_is_one(argv[2], pairs) ? "ONE" : "NOT-ONE");
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "setup") == 0) return cmd_setup(argc, argv);
    if (argc >= 2 && strcmp(argv[1], "verify") == 0) return cmd_verify(argc, argv);
    if (argc >= 2 && strcmp(argv[1], "print-proof") == 0) return cmd_print_proof(argc, argv);
    if (argc >= 2 && strcmp(argv[1], "pairing-check") == 0) return cmd_pairing_check(argc, argv);
    if (argc < 2 || strcmp(argv[1], "generate-proof") != 0) return usage();
    std::string input = "out", witness_path = "witness", pk_path = "proving.key", proof_path = "proof.json", scheme_s = "g16", entropy, cache_dir;
    bool have_entropy = false, timings = false, self_check = false, resident_tables = false;
    int device = 0;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "-i" || a == "--input") input = val();
        else if (a == "--verify") self_check = true;
        else if (a == "--full-tables") resident_tables = true;      // (measurement: build the window-multiple tables a resident prover uses)
        else if (a == "-w" || a == "--witness") witness_path = val();
        else if (a == "-p" || a == "--proving-key-path") pk_path = val();
        else if (a == "-j" || a == "--proof-path") proof_path = val();
        else if (a == "-s" || a == "--proving-scheme") scheme_s = val();
        else if (a == "--entropy") { entropy = val(); have_entropy = true; }
        else if (a == "--key-cache") cache_dir = val();
        else if (a == "--device") device = atoi(val().c_str());
        else if (a == "--timings") timings = true;
        else return usage();
    }
    if (scheme_s != "g16" && scheme_s != "gm17") return usage();
    const Scheme scheme = scheme_s == "gm17" ? Scheme::GM17 : Scheme::G16;
    const auto t_start = std::chrono::steady_clock::now();
    try {
        // host side, beside the key upload: read + decode the program, read the witness
        std::unique_ptr<Program> program;
        Mapped witness;
        std::string host_error;
        int32_t host_code = 0;
        double ms_read = 0, ms_parse = 0;
        std::thread host([&] {
            try {
                auto t0 = std::chrono::steady_clock::now();
                const Mapped prog_bytes(input);
                witness = Mapped(witness_path);
                ms_read = ms_since(t0);
                t0 = std::chrono::steady_clock::now();
                program.reset(new Program(prog_bytes.data, prog_bytes.size));
                ms_parse = ms_since(t0);
            } catch (const Error& e) {
                host_error = e.what();
                host_code = e.code;
            } catch (const std::exception& e) {
                host_error = e.what();
                host_code = ZKHIP_ERR_NOMEM;
            }
        });
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } };
        Joiner joiner{host};
        // the curve of the key: bytes 8..12 of the program file (Field::id)
        int32_t curve = -1;
        {
            FILE* f = fopen(input.c_str(), "rb");
            uint8_t head[12] = {0};
            if (!f || fread(head, 1, 12, f) != 12) { if (f) fclose(f); throw Error(ZKHIP_ERR_PARSE, "Invalid header"); }
            fclose(f);
            static const uint8_t BN[4] = {0xb4, 0xf7, 0xb5, 0xbd}, BLS[4] = {0x40, 0xd8, 0xc1, 0xf9};
            curve = !memcmp(head + 8, BN, 4) ? ZKHIP_CURVE_BN128 : !memcmp(head + 8, BLS, 4) ? ZKHIP_CURVE_BLS12_381 : -1;
            if (memcmp(head, "ZOK\0", 4) != 0 || curve < 0) throw Error(ZKHIP_ERR_PARSE, "not a ZoKrates program for bn128 / bls12_381");
        }
        // which bytes hold the key (the image of an earlier run, if --key-cache has one): mapped on a second thread while the
        // HIP runtime starts
        std::string key_source = "proving.key", image_path;
        bool try_image = false;
        if (!cache_dir.empty()) {
            struct stat st;
            if (stat(pk_path.c_str(), &st) != 0) throw Error(ZKHIP_ERR_BAD_ARG, "cannot stat " + pk_path);
            char tag[64];
            snprintf(tag, sizeof(tag), "%016llx", (unsigned long long)fnv1a(pk_path + "|" + std::to_string((long long)st.st_size) + "|" +
                                                                           std::to_string((long long)st.st_mtim.tv_sec) + "." + std::to_string((long long)st.st_mtim.tv_nsec) + "|" +
                                                                           std::to_string((unsigned long long)st.st_ino) + "|" + scheme_s + "|" + std::to_string(curve) + (resident_tables ? "|tables" : "")));
            image_path = cache_dir + "/" + tag + ".zkhippk";
            struct stat ist;
            try_image = stat(image_path.c_str(), &ist) == 0;
        }
        Mapped key_bytes;
        std::string key_error;
        std::thread key_reader([&] {
            try {
                key_bytes = Mapped(try_image ? image_path : pk_path);
            } catch (const std::exception& e) {
                key_error = e.what();
            }
        });
        Joiner key_joiner{key_reader};
        auto t0 = std::chrono::steady_clock::now();
        Hip hip(device);
        if (!resident_tables) hip.one_shot();          // one proof, then the process ends: no window-multiple tables
        const double ms_init = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        key_reader.join();
        if (!key_error.empty()) throw Error(ZKHIP_ERR_BAD_ARG, key_error);
        Key key;
        if (try_image) {
            try {
                key = hip.import_key_image(key_bytes.data, key_bytes.size);
                key_source = "image";
            } catch (const Error&) {
                key_bytes = Mapped(pk_path);                  // a stale image of another library build: rewrite it below
            }
        }
        if (!key) {
            key = hip.load_proving_key(scheme, curve, key_bytes.data, key_bytes.size);
            if (!image_path.empty()) {
                mkdir(cache_dir.c_str(), 0777);
                const std::vector<uint8_t> img = hip.export_key_image(key);
                const std::string tmp = image_path + ".tmp" + std::to_string((long long)getpid());
                std::ofstream o(tmp, std::ios::binary);
                o.write((const char*)img.data(), (std::streamsize)img.size());
                o.close();
                // only a completely written image enters the cache (ENOSPC, I/O error: the run goes on without one)
                if (!o || rename(tmp.c_str(), image_path.c_str()) != 0) {
                    unlink(tmp.c_str());
                    fprintf(stderr, "warning: key image not cached (%s could not be written)\n", image_path.c_str());
                }
            }
        }
        const double ms_key = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        host.join();
        const double ms_wait = ms_since(t0);
        if (!program) throw Error(host_code ? host_code : ZKHIP_ERR_PARSE, host_error);
        StdRng rng = have_entropy ? get_rng_from_entropy(entropy) : StdRng::from_os_entropy();
        Timings tm;
        const Proof proof = hip.prove(scheme, *program, witness.data, witness.size, key, rng, &tm);
        t0 = std::chrono::steady_clock::now();
        {
            std::ofstream o(proof_path);
            if (!o) throw Error(ZKHIP_ERR_BAD_ARG, "cannot write " + proof_path);
            o << proof.to_json();
            o.close();
            if (!o) throw Error(ZKHIP_ERR_BAD_ARG, "error while writing " + proof_path);
        }
        const double ms_json = ms_since(t0);
        // --verify: the proof just written against the verification key at the head of the proving key, on the host CPU
        // (csrc/host/verify.cpp; ~40 ms on bn128) — a proof the pairing check refuses is an error, not an output
        double ms_verify = 0;
        if (self_check) {
            t0 = std::chrono::steady_clock::now();
            // only the head of the key file is read: the fixed points, the count, then that many G1 points
            const size_t fq = curve == ZKHIP_CURVE_BN128 ? 32 : 48, fixed = scheme == Scheme::GM17 ? 16 * fq : 14 * fq;
            std::vector<uint8_t> head(fixed + 8);
            std::ifstream kf(pk_path, std::ios::binary);
            if (!kf.read((char*)head.data(), (std::streamsize)head.size())) throw Error(ZKHIP_ERR_PARSE, "proving key too short for its verification key");
            uint64_t count;
            memcpy(&count, head.data() + fixed, 8);
            if (count > (1u << 24)) throw Error(ZKHIP_ERR_PARSE, "proving key: implausible number of public inputs");
            head.resize(fixed + 8 + count * 2 * fq);
            if (count && !kf.read((char*)head.data() + fixed + 8, (std::streamsize)(count * 2 * fq))) throw Error(ZKHIP_ERR_PARSE, "proving key too short for its verification key");
            const VerificationKey vk = VerificationKey::from_json(verification_key_json(scheme, curve, head.data(), head.size()));
            if (!verify(vk, proof)) {
                remove(proof_path.c_str());
                throw Error(ZKHIP_ERR_UNSATISFIED, "the proof does not verify against the key of " + pk_path + " (witness not satisfying the program, or a key for another program)");
            }
            ms_verify = ms_since(t0);
            printf("verified against the verification key of %s\n", pk_path.c_str());
        }
        printf("generate-proof (%s): wrote %s\n", scheme_s.c_str(), proof_path.c_str());
        const std::pair<int, int> where = cpu_and_node();
        if (timings)
            printf("timings {\"read_program_and_witness_ms\": %.3f, \"parse_program_ms\": %.3f, \"hip_init_ms\": %.3f, \"key_load_ms\": %.3f, "
                   "\"wait_for_host_side_ms\": %.3f, \"witness_to_assignment_ms\": %.3f, \"r1cs_upload_ms\": %.3f, \"prove_ms\": %.3f, "
                   "\"proof_json_ms\": %.3f, \"verify_ms\": %.3f, \"total_in_process_ms\": %.3f, \"key_source\": \"%s\", \"constraints\": %llu, "
                   "\"tables\": \"%s\", \"host_threads\": %u, \"cpu\": %d, \"numa_node\": %d}\n",
                   ms_read, ms_parse, ms_init, ms_key, ms_wait, tm.witness_to_assignment, tm.r1cs_upload, tm.prove, ms_json, ms_verify, ms_since(t_start),
                   key_source.c_str(), (unsigned long long)program->constraints(), resident_tables ? "every window multiple" : "none (one proof per process)",
                   std::thread::hardware_concurrency(), where.first, where.second);
        // one proof per process: the proof is on disk, so leave without tearing down 6 GiB of tables and the HIP runtime
        fflush(stdout);
        _exit(0);
    } catch (const Error& e) {
        fprintf(stderr, "zkhip-cli: %s\n", e.what());
        return 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "zkhip-cli: %s\n", e.what());
        return 1;
    }
}
