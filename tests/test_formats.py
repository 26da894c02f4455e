"""Formats either side of the hot path (zokrates_amd/formats.py) against the reference's own byte-level test vectors
(/root/reference/zokrates_circom/src/r1cs.rs:242-432, witness.rs:113-219) and the CLI shim end to end (on the TEST-ONLY
emulator build here; `-m gpu` runs the same flow on the real library)."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import formats as oformats
from oracle import pairing
from oracle.fields import BN254
from zokrates_amd import formats, synth

MOD = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")
ONE = (1).to_bytes(32, "little")


def lc(*terms):
    return struct.pack("<I", len(terms)) + b"".join(struct.pack("<I", w) + int(c).to_bytes(32, "little") for w, c in terms)


def r1cs_file(constraints, n_wires, n_out, n_in, n_prv):
    body = b"".join(lc(*a) + lc(*b) + lc(*c) for a, b, c in constraints)
    return (b"r1cs" + struct.pack("<II", 1, 3) + struct.pack("<IQ", 2, len(body)) + body + struct.pack("<IQ", 1, 64) + struct.pack("<I", 32) + MOD
            + struct.pack("<IIIIQI", n_wires, n_out, n_in, n_prv, n_wires, len(constraints)) + struct.pack("<IQ", 3, 8 * n_wires)
            + b"".join(struct.pack("<Q", i) for i in range(n_wires)))


# the three programs of the reference's tests, in the reference's expected bytes
R1CS_KATS = {
    "empty": (r1cs_file([], 1, 0, 0, 0), 1, 0, 0, 0),
    "return_one": (r1cs_file([([(0, 1)], [(0, 1)], [(1, 1)])], 2, 1, 0, 0), 2, 1, 0, 0),
    "with_inputs": (r1cs_file([([(3, 1)], [(3, 1)], [(3, 1)]), ([(0, 1)], [(3, 1), (2, 1)], [(1, 1)])], 4, 1, 1, 1), 4, 1, 1, 1),
}


def test_r1cs_kat_prefix_matches_reference_bytes():
    """Spot-check the helper against literal bytes quoted from r1cs.rs `empty()` (lines 248-277)."""
    expected = bytes([0x72, 0x31, 0x63, 0x73, 1, 0, 0, 0, 3, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0x40, 0, 0, 0, 0, 0, 0, 0,
                      0x20, 0, 0, 0]) + MOD + bytes([1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                      3, 0, 0, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert R1CS_KATS["empty"][0] == expected


@pytest.mark.parametrize("name", list(R1CS_KATS))
def test_r1cs_read_write_roundtrip(name):
    data, n_wires, n_out, n_in, n_prv = R1CS_KATS[name]
    r = formats.read_r1cs(data)
    assert (r.curve_id, r.n_wires, r.n_pub_out, r.n_pub_in, r.n_prv_in) == (0, n_wires, n_out, n_in, n_prv)
    assert r.l == 1 + n_out + n_in and r.w == n_wires - r.l
    assert formats.write_r1cs(0, n_wires, n_out, n_in, n_prv, r.mats) == data
    if name == "with_inputs":
        assert r.n == 2 and r.mats[1][1].tolist() == [3, 3, 2] and r.mats[1][0].tolist() == [0, 1, 3]


def test_r1cs_rejects_garbage():
    data = R1CS_KATS["with_inputs"][0]
    for bad in (b"xxxx" + data[4:], data[:-1][:60], data[:12] + struct.pack("<IQ", 2, 10) + data[24:]):
        with pytest.raises((formats.FormatError, struct.error, KeyError)):
            formats.read_r1cs(bad)


def test_wtns_kats():
    head = b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 0x28) + struct.pack("<I", 32) + MOD
    empty = head + struct.pack("<I", 0) + struct.pack("<IQ", 2, 0)
    one = head + struct.pack("<I", 1) + struct.pack("<IQ", 2, 32) + ONE
    four = head + struct.pack("<I", 4) + struct.pack("<IQ", 2, 128) + b"".join(int(v).to_bytes(32, "little") for v in (1, 42, 44, 43))
    assert formats.write_wtns(0, np.zeros(0, dtype=np.uint8)) == empty
    assert formats.write_wtns(0, np.frombuffer(ONE, dtype=np.uint8)) == one
    cid, vals = formats.read_wtns(four)           # witness.rs `one_and_pub_and_priv`: [one, ~out_0, _1, _0]
    assert cid == 0 and [int.from_bytes(vals[32 * i:32 * i + 32].tobytes(), "little") for i in range(4)] == [1, 42, 44, 43]
    assert formats.write_wtns(0, vals) == four


def test_zokrates_witness_file():
    # ir/witness.rs:44-71: usize count, then (isize id, 32-byte LE value); ids: 0 = ~one, 43 = _42, -9 = ~out_8
    data = struct.pack("<Q", 3) + struct.pack("<q", -9) + (8).to_bytes(32, "little") + struct.pack("<q", 0) + ONE + struct.pack("<q", 43) + (42).to_bytes(32, "little")
    assert formats.read_zokrates_witness(data) == {-9: 8, 0: 1, 43: 42}
    with pytest.raises(formats.FormatError):
        formats.read_zokrates_witness(data[:-1])


def test_proof_json_matches_oracle_writer():
    raw = bytes(range(1, 33)) * 8 + bytes([0, 0, 0])
    mine = formats.proof_json(0, raw, [5, 2 ** 200 + 7])
    assert mine == oformats.proof_json(BN254, oformats.proof_from_raw(BN254, raw), [5, 2 ** 200 + 7])
    doc = json.loads(mine)
    assert doc["scheme"] == "g16" and doc["curve"] == "bn128" and len(doc["proof"]["b"]) == 2 and doc["inputs"][0] == "0x" + "00" * 31 + "05"


def _cli_flow(tmp_path, lg=4):
    """synthetic circuit -> .r1cs/.wtns -> CLI setup + generate-proof -> parse JSON -> pairing check (oracle O3)."""
    from zokrates_amd import cli
    circ = synth.circuit(0, lg, seed=0x600D)
    z = circ.assignment(99)
    r1 = tmp_path / "c.r1cs"; wt = tmp_path / "c.wtns"; pkp = tmp_path / "proving.key"; vkp = tmp_path / "verification.key"; pj = tmp_path / "proof.json"
    r1.write_bytes(formats.write_r1cs(0, circ.m, 0, circ.l - 1, circ.w, circ.mats()))
    wt.write_bytes(formats.write_wtns(0, z))
    cli.main(["setup", "-i", str(r1), "-p", str(pkp), "-v", str(vkp), "--entropy", "unit test"])
    cli.main(["generate-proof", "-i", str(r1), "-w", str(wt), "-p", str(pkp), "-j", str(pj), "--entropy", "abc"])
    proof = json.loads(pj.read_text())
    vk = json.loads(vkp.read_text())
    h = lambda s: int(s, 16)
    g1 = lambda p: (h(p[0]), h(p[1]))
    g2 = lambda p: ((h(p[0][0]), h(p[0][1])), (h(p[1][0]), h(p[1][1])))
    ovk = dict(alpha_g1=g1(vk["alpha"]), beta_g2=g2(vk["beta"]), gamma_g2=g2(vk["gamma"]), delta_g2=g2(vk["delta"]),
               gamma_abc_g1=[g1(p) for p in vk["gamma_abc"]])
    pts = (g1(proof["proof"]["a"]), g2(proof["proof"]["b"]), g1(proof["proof"]["c"]))
    inputs = [h(x) for x in proof["inputs"]]
    assert len(inputs) == circ.l - 1 and inputs[0] == int.from_bytes(z[32:64].tobytes(), "little")
    assert pairing.groth16_verify(BN254, ovk, pts, inputs)
    assert not pairing.groth16_verify(BN254, ovk, pts, [inputs[0] + 1])
    # same entropy -> same artefacts (zokrates_js/tests/tests.js:189-203, 248-267)
    first = pj.read_text()
    cli.main(["generate-proof", "-i", str(r1), "-w", str(wt), "-p", str(pkp), "-j", str(pj), "--entropy", "abc"])
    assert pj.read_text() == first


def test_cli_flow_on_emulator(tmp_path, monkeypatch):
    from emu_util import emu_library
    from zokrates_amd import native
    monkeypatch.setattr(native, "_default", emu_library())
    _cli_flow(tmp_path)


@pytest.mark.gpu
def test_cli_flow_on_gpu(tmp_path):
    _cli_flow(tmp_path, lg=10)


def test_reference_rng_restatement():
    """zokrates_amd/rng.py: the ChaCha core against RFC 7539 §2.3.2, the seed derivation of
    /root/reference/zokrates_proof_systems/src/rng.rs:5-20, and the shape of ark's `Fr::rand`."""
    import hashlib
    import struct
    from zokrates_amd import rng
    key = struct.unpack("<8I", bytes(range(32)))
    blk = rng.chacha_block(key, 1 | (0x09000000 << 32), 0x4A000000, 20)
    assert struct.pack("<16I", *blk).hex().startswith("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e")
    g = rng.rng_from_entropy("some entropy")
    assert bytes(struct.pack("<8I", *g.key)) == hashlib.blake2b(b"some entropy").digest()[:32]
    first = rng.chacha_block(g.key, 0, 0, 12)
    assert g.next_u64() == first[0] | (first[1] << 32) and g.next_u64() == first[2] | (first[3] << 32)
    for cid, (p, shave) in rng.FR.items():
        a, b = rng.rng_from_entropy("x"), rng.rng_from_entropy("x")
        va = [rng.fr_rand(a, cid) for _ in range(50)]
        assert va == [rng.fr_rand(b, cid) for _ in range(50)] and all(0 <= v < p for v in va) and len(set(va)) == 50
        # the accepted limbs are the Montgomery form: value * 2^256 mod p fits the masked width and is below p
        for v in va:
            mont = v * (1 << 256) % p
            assert mont < p and mont >> (256 - shave) == 0
    # 17 blocks of 16 words: the block counter advances
    g = rng.rng_from_entropy("y")
    words = [g.next_u32() for _ in range(40)]
    assert words[:16] == rng.chacha_block(g.key, 0, 0, 12) and words[16:32] == rng.chacha_block(g.key, 1, 0, 12)


def test_rng_chain_published_known_answers():
    """The only link between `zokrates generate-proof --entropy` and this backend's bytes that the reference's tests do not
    pin is the r, s draw: Blake2b-512 -> 32-byte seed -> rand 0.8.5 `StdRng` (= rand_chacha 0.3.1 `ChaCha12Rng`) -> ark-ff
    0.3.0 `Fr::rand`.  Every primitive of that chain against the known answers its own specification publishes:
      * BLAKE2b-512("abc"): RFC 7693 Appendix A;
      * the ChaCha block function with a 256-bit all-zero key, zero IV, block 0 at 8 / 12 / 20 rounds:
        draft-strombergson-chacha-test-vectors-01, TC1 (the 20-round words are also the expectation of rand_chacha's own
        `test_chacha_true_values_a`: 0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653, ...);
      * RFC 7539 §2.3.2 (key 00..1f, nonce 00:00:00:09:00:00:00:4a:00:00:00:00, counter 1) for the state layout — rand_chacha
        keeps a 64-bit block counter in words 12-13 and the stream id in 14-15, the layout the RFC vector exercises when its
        nonce word 0 is read as the counter's high half;
      * `next_u64` = two consecutive words, low first, across block boundaries (rand_core `BlockRng::next_u64`);
      * ark-ff's sampler: four u64 limbs, the top REPR_SHAVE_BITS = 256 - modulus bits of the last one cleared, rejected
        unless below the modulus, and the limbs ARE the Montgomery representation."""
    import hashlib
    import struct
    from zokrates_amd import rng
    assert hashlib.blake2b(b"abc").hexdigest() == ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
                                                   "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    zero = [0] * 8
    tc1 = {8: "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42",
           12: "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be",
           20: "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586"}
    for rounds, want in tc1.items():
        assert struct.pack("<16I", *rng.chacha_block(zero, 0, 0, rounds)).hex() == want, rounds
    assert rng.chacha_block(zero, 0, 0, 20)[:4] == [0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653]
    # StdRng from the all-zero seed: the 12-round block, word by word, then the next block (counter 1)
    g = rng.StdRng(bytes(32))
    w0 = struct.unpack("<16I", bytes.fromhex(tc1[12]))
    assert [g.next_u32() for _ in range(15)] == list(w0[:15])
    nxt = rng.chacha_block(zero, 1, 0, 12)
    assert g.next_u64() == w0[15] | (nxt[0] << 32)          # a u64 that straddles two blocks
    # Fr::rand: the shave widths follow from the moduli; an out-of-range draw is rejected and the NEXT four limbs are used
    for cid, (p, shave) in rng.FR.items():
        assert shave == 256 - p.bit_length()

    class Fixed:
        def __init__(self, u64s):
            self.q = list(u64s)

        def next_u64(self):
            return self.q.pop(0)

    p, shave = rng.FR[0]
    too_big = [(1 << 64) - 1] * 4                             # masked to 2^254 - 1 >= p: rejected
    mont = 0x0123456789abcdef_0fedcba987654321_1111111122222222_0000000000000007
    limbs = [(mont >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]
    got = rng.fr_rand(Fixed(too_big + limbs), 0)
    assert got == mont * pow(1 << 256, -1, p) % p and got * (1 << 256) % p == mont
