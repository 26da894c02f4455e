#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the read-only reference tree (run in the build container only;
/root/reference does not exist on the GPU box, so tests consume the committed JSON, never this script).

Sources (all in /root/reference):
  * zokrates_field/src/bn128.rs:44-240,273-292      Fr known-answer tests (decimal literals)
  * zokrates_proof_systems/src/scheme/groth16.rs:157 snark_scalar_field
  * zokrates_proof_systems/src/solidity.rs:24-26,430-441  FIELD_MODULUS, TWISTBX/Y, P1(), P2()
  * zokrates_book/src/toolbox/ir.md:15               curve id of bn128
  * zokrates_cli/examples/book/mpc_tutorial/phase1radix2m2   BN254 points (bellman uncompressed BE)
  * zokrates_ast/src/ir/witness.rs:99-156            witness binary layout (restated as a vector)
  * zokrates_core_test/tests/tests/snark/snark_verify_bls12_377_{1,2,5}.json   three more GM17 / BLS12-377 triples
                                                     (flattened decimal), expected to verify
  * zokrates_stdlib/tests/tests/snark/gm17.json      the reference's only golden (proof, vk, inputs) triple: GM17 over
                                                     BLS12-377, expected to verify (`"Ok": {"value": true}`)
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def must_contain(path, literal):
    txt = open(os.path.join(REF, path)).read()
    if literal not in txt:
        sys.exit(f"literal {literal!r} not found in {path}")
    return txt


def field_kats():
    p = "zokrates_field/src/bn128.rs"
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    kats = [
        # (op, a, b, expected)  -- negative operands as written in the reference (From<i32>)
        ("add", "65416358", "68135", "65484493"),
        ("add", "5", "-2", "3"),
        ("add", "65416358", "-68135", "65348223"),
        ("sub", "65416358", "68135", "65348223"),
        ("sub", "65416358", "-68135", "65484493"),
        ("sub", "68135", "65416358",
         "21888242871839275222246405745257275088548364400416034343698204186575743147394"),
        ("mul", "32", "421", "13472"),
        ("mul", "54", "-8912",
         "21888242871839275222246405745257275088548364400416034343698204186575808014369"),
        ("pow", "54", "12", "614787626176508399616"),
        ("div", "-1", "2",
         "10944121435919637611123202872628637544274182200208017171849102093287904247808"),
    ]
    for _, a, b, e in kats:
        for lit in (a.lstrip("-"), b.lstrip("-"), e):
            must_contain(p, lit)
    must_contain("zokrates_proof_systems/src/scheme/groth16.rs", str(R))
    return {"modulus": str(R), "source": p, "kats": kats}


def curve_consts():
    sol = open(os.path.join(REF, "zokrates_proof_systems/src/solidity.rs")).read()
    q = int(re.search(r"FIELD_MODULUS = (0x[0-9a-f]+)", sol).group(1), 16)
    tbx = int(re.search(r"TWISTBX = (0x[0-9a-f]+)", sol).group(1), 16)
    tby = int(re.search(r"TWISTBY = (0x[0-9a-f]+)", sol).group(1), 16)
    # P2(): the four literals inside function P2()
    p2 = sol[sol.index("function P2()"):]
    nums = [int(x) for x in re.findall(r"\b(\d{60,})\b", p2)[:4]]
    ir = must_contain("zokrates_book/src/toolbox/ir.md", "0xb4f7b5bd")
    return {
        "bn254_q": str(q), "twist_b_c0": str(tbx), "twist_b_c1": str(tby),
        # Solidity G2Point layout is X = [x.c1, x.c0], Y = [y.c1, y.c0] (solidity.rs:424 comment)
        "p2_literals": [str(n) for n in nums],
        "bn128_curve_id": "b4f7b5bd",
    }


def phase1_points():
    raw = open(os.path.join(REF, "zokrates_cli/examples/book/mpc_tutorial/phase1radix2m2"), "rb").read()
    assert len(raw) == 1728
    o = 0

    def g1():
        nonlocal o
        x = int.from_bytes(raw[o:o + 32], "big"); y = int.from_bytes(raw[o + 32:o + 64], "big"); o += 64
        return [str(x), str(y)]

    def g2():
        nonlocal o
        v = [int.from_bytes(raw[o + 32 * i:o + 32 * i + 32], "big") for i in range(4)]; o += 128
        # bellman uncompressed: x.c1 | x.c0 | y.c1 | y.c0
        return [[str(v[1]), str(v[0])], [str(v[3]), str(v[2])]]

    d = {"alpha_g1": g1(), "beta_g1": g1(), "beta_g2": g2()}
    d["coeffs_g1"] = [g1() for _ in range(4)]   # L_i(tau)*G1, Lagrange basis of the size-4 domain
    d["coeffs_g2"] = [g2() for _ in range(4)]   # L_i(tau)*G2
    d["alpha_coeffs_g1"] = [g1() for _ in range(4)]
    d["beta_coeffs_g1"] = [g1() for _ in range(4)]
    d["h_g1"] = [g1() for _ in range(3)]
    assert o == 1728
    return d


def witness_vector():
    # layout of Witness::write (witness.rs:44-53): usize LE count, then (isize LE id, 32-byte LE value)
    # in BTreeMap order; ids: ~out_8 -> -9, ~one -> 0, _42 -> 43 (variable.rs:6-35)
    must_contain("zokrates_ast/src/ir/witness.rs", '"~out_8": "8"')
    import struct
    entries = [(-9, 8), (0, 1), (43, 42)]
    b = struct.pack("<Q", len(entries))
    for vid, val in entries:
        b += struct.pack("<q", vid) + val.to_bytes(32, "little")
    return {"hex": b.hex(), "json": {"~out_8": "8", "~one": "1", "_42": "42"}}


def gm17_triple():
    doc = json.load(open(os.path.join(REF, "zokrates_stdlib/tests/tests/snark/gm17.json")))
    t = doc["tests"][0]
    proof, vk = t["input"]["values"]
    assert t["output"] == {"Ok": {"value": True}}
    return {"source": "zokrates_stdlib/tests/tests/snark/gm17.json", "curve": "bls12_377", "expected": True,
            "proof": proof["proof"], "inputs": proof["inputs"], "vk": vk}


def gm17_embed_triples():
    """zokrates_core_test/tests/tests/snark/snark_verify_bls12_377_{1,2,5}.json: (proof[8], inputs[k], vk[16 + 2(k+1)])
    flattened to decimal strings in proof.json / verification.key order (see the .zok files next to them); all three
    are expected to verify."""
    out = []
    for k in (1, 2, 5):
        rel = f"zokrates_core_test/tests/tests/snark/snark_verify_bls12_377_{k}.json"
        t = json.load(open(os.path.join(REF, rel)))["tests"][0]
        assert t["output"] == {"Ok": {"value": True}}
        proof, inputs, vk = t["input"]["values"]
        assert len(proof) == 8 and len(inputs) == k and len(vk) == 16 + 2 * (k + 1)
        out.append({"source": rel, "proof": proof, "inputs": inputs, "vk": vk})
    return {"curve": "bls12_377", "expected": True, "layout": "proof: a.x a.y b.x.c0 b.x.c1 b.y.c0 b.y.c1 c.x c.y; "
            "vk: h(4) g_alpha(2) h_beta(4) g_gamma(2) h_gamma(4) query(2 each)", "triples": out}


def ptau5_points():
    """zokrates_js/tests/powersOfTau5_0000.ptau: a snarkjs powers-of-tau file (power 5) the reference's JS tests load for the
    universal-setup flows.  Sections (id u32, size u64): 1 header (n8, q, power), 2 tauG1 (2^6 - 1 points), 3 tauG2 (2^5),
    4 alphaTauG1, 5 betaTauG1, 6 betaG2, 12..15 the same in the Lagrange bases of the domains 2^0 .. 2^6 / 2^5.  Coordinates are
    Montgomery form (R = 2^256), little-endian; G2 as x.c0 x.c1 y.c0 y.c1; infinity as all-zero.  `_0000` is the ceremony's
    starting file (tau = 1): every tauG1 entry IS the generator, and the Lagrange sections are mostly points at infinity plus
    the generator and one domain's worth of genuine distinct points — an unfriendly base set for an MSM (infinite bases,
    long runs of equal bases: the doubling and cancellation branches), and an independent one: none of it came out of this
    repository's setup code."""
    import struct
    raw = open(os.path.join(REF, "zokrates_js/tests/powersOfTau5_0000.ptau"), "rb").read()
    assert raw[:4] == b"ptau"
    off, secs = 12, {}
    while off < len(raw):
        sid, size = struct.unpack("<IQ", raw[off:off + 12])
        secs[sid] = (off + 12, size)
        off += 12 + size
    o, _ = secs[1]
    n8 = struct.unpack("<I", raw[o:o + 4])[0]
    q = int.from_bytes(raw[o + 4:o + 4 + n8], "little")
    power = struct.unpack("<I", raw[o + 4 + n8:o + 8 + n8])[0]
    assert n8 == 32 and power == 5
    rinv = pow(1 << 256, -1, q)
    fq = lambda b: str(int.from_bytes(b, "little") * rinv % q)

    def points(sid, ncoord):
        o, size = secs[sid]
        sz = 32 * ncoord
        return [[fq(raw[o + sz * i + 32 * k:o + sz * i + 32 * k + 32]) for k in range(ncoord)] for i in range(size // sz)]
    d = {"source": "zokrates_js/tests/powersOfTau5_0000.ptau", "q": str(q), "power": power,
         "tau_g1": points(2, 2), "tau_g2": points(3, 4), "lagrange_g1": points(12, 2), "lagrange_g2": points(13, 4)}
    assert len(d["tau_g1"]) == 63 and len(d["tau_g2"]) == 32 and len(d["lagrange_g1"]) == 127 and len(d["lagrange_g2"]) == 63
    return d


def sha256_packed_kat():
    """The reference's known answer for stdlib hashes/sha256/512bitPacked.zok (its own test of that program)."""
    rel = "zokrates_stdlib/tests/tests/hashes/sha256/512bitPacked.json"
    doc = json.load(open(os.path.join(REF, rel)))
    assert doc["entry_point"].endswith("hashes/sha256/512bitPacked.zok") and doc["curves"] == ["Bn128"]
    t = doc["tests"][0]
    return {"source": rel, "input": t["input"]["values"][0], "output": t["output"]["Ok"]["value"]}


def main():
    json.dump(sha256_packed_kat(), open(os.path.join(OUT, "sha256_packed_kat.json"), "w"), indent=1)
    json.dump(ptau5_points(), open(os.path.join(OUT, "ptau5_points.json"), "w"), indent=0)
    json.dump(gm17_embed_triples(), open(os.path.join(OUT, "gm17_bls12_377_embed_triples.json"), "w"), indent=1)
    json.dump(gm17_triple(), open(os.path.join(OUT, "gm17_bls12_377_triple.json"), "w"), indent=1)
    json.dump(field_kats(), open(os.path.join(OUT, "bn128_field_kats.json"), "w"), indent=1)
    json.dump(curve_consts(), open(os.path.join(OUT, "bn254_consts.json"), "w"), indent=1)
    json.dump(phase1_points(), open(os.path.join(OUT, "phase1radix2m2_points.json"), "w"), indent=1)
    json.dump(witness_vector(), open(os.path.join(OUT, "witness_kat.json"), "w"), indent=1)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
