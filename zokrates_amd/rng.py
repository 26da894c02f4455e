"""The reference's proof randomness, restated for the CLI shim (host-side product code).

`zokrates generate-proof --entropy TEXT` seeds rand 0.8's `StdRng` with the first 32 bytes of Blake2b-512(TEXT)
(/root/reference/zokrates_proof_systems/src/rng.rs:5-20, call site zokrates_cli/src/ops/generate_proof.rs:183-185) and
hands it to the backend, whose first draws are the blinding scalars: `Fr::rand(rng)` twice (r, s) in
[UPSTREAM] ark_groth16::create_random_proof, three times (d1, d2, r) in ark_gm17::create_random_proof.

[UPSTREAM] pieces restated here (none of them can be executed in this image, and the reference pins no vector for
them, so this is "parity unpinned" like the rest of the ark side; the ChaCha core below is checked against RFC 7539):
  * rand 0.8.5 `StdRng` = rand_chacha 0.3.1 `ChaCha12Rng`: key = seed, 64-bit block counter from 0 (state words 12-13),
    stream id 0 (words 14-15), output = the 16 words of each block in order; `next_u64` = two consecutive words, low first;
  * ark-ff 0.3.0 `impl Distribution<Fp256<P>> for Standard`: four `next_u64` limbs (little-endian), the top
    `REPR_SHAVE_BITS` bits of the last limb masked off, rejected unless below the modulus — and the accepted limbs ARE the
    Montgomery representation, so the field element is limbs * R^-1 mod p.
"""
import hashlib
import struct

MASK32 = 0xFFFFFFFF
FR = {
    0: (21888242871839275222246405745257275088548364400416034343698204186575808495617, 2),   # bn128: modulus, REPR_SHAVE_BITS
    1: (52435875175126190479447740508185965837690552500527637822603658699938581184513, 1),   # bls12_381
}


def _rotl(v, n):
    return ((v << n) & MASK32) | (v >> (32 - n))


def chacha_block(key_words, counter, stream, rounds):
    """One 64-byte ChaCha block as 16 little-endian words (counter: 64 bits in words 12-13, stream id in 14-15)."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & MASK32, (counter >> 32) & MASK32,
                                                                                     stream & MASK32, (stream >> 32) & MASK32]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & MASK32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & MASK32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(x, init)]


class StdRng:
    """rand 0.8 `StdRng::from_seed(seed)`."""

    def __init__(self, seed):
        assert len(seed) == 32
        self.key = struct.unpack("<8I", seed)
        self.counter = 0
        self.words = []

    def next_u32(self):
        if not self.words:
            self.words = chacha_block(self.key, self.counter, 0, 12)
            self.counter += 1
        return self.words.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


def rng_from_entropy(entropy):
    """get_rng_from_entropy (rng.rs:5-20)."""
    return StdRng(hashlib.blake2b(entropy.encode()).digest()[:32])


def fr_rand(rng, curve_id):
    """ark-ff `Fr::rand(rng)` as a canonical integer."""
    p, shave = FR[curve_id]
    r_inv = pow(1 << 256, -1, p)
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << (64 - shave)) - 1
        mont = sum(v << (64 * i) for i, v in enumerate(limbs))
        if mont < p:
            return mont * r_inv % p
