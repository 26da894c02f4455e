// TEST PROGRAM for the long-lived prover of the compiled host side (include/zkhip_backend.hpp: System, Hip::load_system, Hip::bind,
// Hip::prove over a System): one key and one constraint system resident, the same witness proved (1) before the key is bound, (2)
// bound, (3) bound again with fresh randomness from the same entropy, (4) through the one-call form (its own upload of the system:
// the key's own tables) — every proof.json text printed for the caller to compare with `generate-proof --entropy`.
// usage: resident_prover <out> <witness> <proving.key> <entropy> <g16|gm17>
#include <cstdio>
#include <fstream>
#include <iterator>

#include "../../include/zkhip_backend.hpp"

using namespace zokrates_hip;

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc != 6) return 2;
    try {
        const std::vector<uint8_t> out = slurp(argv[1]), wit = slurp(argv[2]), pkb = slurp(argv[3]);
        const std::string entropy = argv[4];
        const Scheme scheme = std::string(argv[5]) == "gm17" ? Scheme::GM17 : Scheme::G16;
        Hip hip(0);
        hip.separate_dispatchers();                           // (a stream plan is placement only: the same bytes on any layout)
        Program prog(out.data(), out.size());
        Key key = hip.load_proving_key(scheme, prog.curve(), pkb.data(), pkb.size());
        System sys = hip.load_system(prog);
        auto prove = [&](bool one_call) {
            StdRng rng = get_rng_from_entropy(entropy);
            return one_call ? hip.prove(scheme, prog, wit.data(), wit.size(), key, rng) : hip.prove(scheme, sys, wit.data(), wit.size(), key, rng);
        };
        const std::string before = prove(false).to_json();
        bool bound = false, refused = false;
        try {
            bound = hip.bind(key, sys);
        } catch (const Error&) {
            refused = true;                                   // (a key that does not match its system: not in this program)
        }
        const std::string after = prove(false).to_json(), again = prove(false).to_json(), one = prove(true).to_json();
        printf("bound=%d refused=%d is_bound=%d\n", (int)bound, (int)refused, (int)hip.is_bound(key, sys));
        printf("same=%d\n", (int)(before == after && after == again && again == one));
        printf("%s\n", after.c_str());
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "resident_prover: %s\n", e.what());
        return 1;
    }
}
