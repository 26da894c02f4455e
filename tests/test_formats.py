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
