// Prints what the C++ host layer (zokrates_amd/csrc/host/backend.cpp) computes for the (r, s) chain, one "name hex" line
// each; tests/test_native_backend.py compares the lines with hashlib / zokrates_amd/rng.py and the published vectors.
#include <cstdio>
#include <string>

#include "../../include/zkhip_backend.hpp"

using namespace zokrates_hip;

static void hex(const char* name, const uint8_t* p, size_t n) {
    printf("%s ", name);
    for (size_t i = 0; i < n; ++i) printf("%02x", p[i]);
    printf("\n");
}

int main() {
    auto b = blake2b_512((const uint8_t*)"abc", 3);
    hex("blake2b_abc", b.data(), 64);
    b = blake2b_512(nullptr, 0);
    hex("blake2b_empty", b.data(), 64);
    std::string longmsg;
    for (int i = 0; i < 300; ++i) longmsg.push_back((char)(i * 7 + 1));
    b = blake2b_512((const uint8_t*)longmsg.data(), longmsg.size());
    hex("blake2b_300", b.data(), 64);
    b = blake2b_512((const uint8_t*)longmsg.data(), 128);
    hex("blake2b_128", b.data(), 64);
    b = blake2b_512((const uint8_t*)longmsg.data(), 256);
    hex("blake2b_256", b.data(), 64);
    std::array<uint8_t, 32> zero{};
    StdRng z(zero);
    uint32_t w[20];
    for (int i = 0; i < 15; ++i) w[i] = z.next_u32();
    hex("chacha12_zero_first15", (const uint8_t*)w, 60);
    uint64_t straddle = z.next_u64();
    hex("chacha12_zero_straddle", (const uint8_t*)&straddle, 8);
    for (int curve = 0; curve < 2; ++curve)
        for (const char* ent : {"bench", "golden vector 1", ""}) {
            StdRng g = get_rng_from_entropy(ent);
            for (int k = 0; k < 3; ++k) {
                auto v = fr_rand(g, curve);
                char name[64];
                snprintf(name, sizeof(name), "fr_rand_%d_%s_%d", curve, *ent ? (ent[0] == 'b' ? "bench" : "golden") : "empty", k);
                hex(name, v.data(), 32);
            }
        }
    return 0;
}
