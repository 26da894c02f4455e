"""`Backend::verify` of the compiled host layer (zokrates_amd/csrc/host/verify.cpp, `zkhip-cli verify`): the pairing check of
/root/reference/zokrates_ark/src/groth16.rs:55-87 and gm17.rs:69-110 on the host CPU.

What pins it:
  * the reference's own BN254 fixture points (zokrates_cli/examples/book/mpc_tutorial/phase1radix2m2 -> tests/golden/
    phase1radix2m2_points.json): bilinearity e(L_i G1, G2) = e(G1, L_i G2) holds for them, a mismatched pair does not;
  * proofs made by the product's own setup + generate-proof (emulator library here, the GPU in test_native_backend's `-m gpu`
    leg) PASS, the mutations the reference tests reject (to_token.rs:68-71: a.x overwritten; a valid but different point; a
    different public input) FAIL, and the oracle's independent pairing (oracle/pairing.py, oracle/gm17.py: flat degree-12
    arithmetic, Miller loop in Fq12) gives the same verdict on the same files;
  * THE REFERENCE'S OWN PROOFS: the four GM17 artefacts the reference ships — zokrates_stdlib/tests/tests/snark/gm17.json and
    zokrates_core_test/tests/tests/snark/snark_verify_bls12_377_{1,2,5}.json, made by `zokrates setup / generate-proof -b ark -s gm17`
    over BLS12-377 (tests/golden/gm17_bls12_377_*.json) — PASS, and fail once an input is changed: the compiled verifier knows
    that curve for exactly this purpose;
  * bilinearity on BLS12-381 with oracle-made multiples of the generators;
  * the CLI's messages and exit codes (ops/verify.rs:95-107,181-195) and `print-proof` (ops/print_proof.rs:85-114)."""
import json
import os
import subprocess

import pytest

from oracle import gm17 as ogm17
from oracle import ir, pairing
from oracle.curves import groups
from oracle.fields import BN254, BLS12_381

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "_emu", "zkhip-cli-emu")


def _env():
    from emu_util import EMU_LIB, emu_library
    emu_library()
    return dict(os.environ, ZKHIP_LIBRARY=EMU_LIB)


def _run(args, env=None):
    return subprocess.run([EXE] + args, capture_output=True, text=True, env=env or _env())


def _hex(v, nbytes):
    return "0x" + int(v).to_bytes(nbytes, "big").hex()


def _g1_hex(P, nb):
    return [_hex(P[0], nb), _hex(P[1], nb)]


def _g2_hex(Q, nb):
    return [[_hex(Q[0][0], nb), _hex(Q[0][1], nb)], [_hex(Q[1][0], nb), _hex(Q[1][1], nb)]]


def _g1_int(v):
    return (int(v[0], 16), int(v[1], 16))


def _g2_int(v):
    return ((int(v[0][0], 16), int(v[0][1], 16)), (int(v[1][0], 16), int(v[1][1], 16)))


def _pairs_file(path, pairs, nb):
    with open(path, "w") as f:
        for P, Q in pairs:
            f.write(" ".join(_g1_hex(P, nb) + _g2_hex(Q, nb)[0] + _g2_hex(Q, nb)[1]) + "\n")


def test_pairing_on_the_reference_mpc_fixture(tmp_path, golden_dir):
    d = json.load(open(os.path.join(golden_dir, "phase1radix2m2_points.json")))
    G1, _ = groups(BN254)
    p1 = lambda v: (int(v[0]), int(v[1]))
    p2 = lambda v: ((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1])))
    cg1, cg2 = [p1(v) for v in d["coeffs_g1"]], [p2(v) for v in d["coeffs_g2"]]
    ag1 = [p1(v) for v in d["alpha_coeffs_g1"]]
    cases = {
        "lagrange": ([(cg1[1], BN254.g2), (G1.aneg(BN254.g1), cg2[1])], "ONE"),
        "alpha": ([(ag1[2], BN254.g2), (G1.aneg(p1(d["alpha_g1"])), cg2[2])], "ONE"),
        "beta": ([(p1(d["beta_g1"]), BN254.g2), (G1.aneg(BN254.g1), p2(d["beta_g2"]))], "ONE"),
        "mismatch": ([(cg1[1], BN254.g2), (G1.aneg(BN254.g1), cg2[2])], "NOT-ONE"),
        "all_three": ([(cg1[1], BN254.g2), (G1.aneg(BN254.g1), cg2[1]), (ag1[2], BN254.g2), (G1.aneg(p1(d["alpha_g1"])), cg2[2]),
                       (p1(d["beta_g1"]), BN254.g2), (G1.aneg(BN254.g1), p2(d["beta_g2"]))], "ONE"),
        "empty": ([], "ONE"),
    }
    for name, (pairs, want) in cases.items():
        f = str(tmp_path / (name + ".txt"))
        _pairs_file(f, pairs, 32)
        r = _run(["pairing-check", "bn128", f])
        assert r.returncode == 0 and r.stdout.strip() == want, (name, r.stdout, r.stderr)


@pytest.mark.parametrize("curve", [BN254, BLS12_381], ids=lambda c: c.name)
def test_bilinearity_with_oracle_made_multiples(tmp_path, curve):
    G1, G2 = groups(curve)
    nb = 32 if curve is BN254 else 48
    a, b = 0x1234567890abcdef1234567890abcdef1234567 % curve.r, (curve.r - 0xfedcba0987654321)
    aP, bQ = G1.amul(G1.gen, a), G2.amul(G2.gen, b)
    abP = G1.amul(G1.gen, a * b % curve.r)
    good, bad = str(tmp_path / "good.txt"), str(tmp_path / "bad.txt")
    _pairs_file(good, [(aP, bQ), (G1.aneg(abP), G2.gen)], nb)                      # e(aP, bQ) = e(abP, Q)
    _pairs_file(bad, [(aP, bQ), (G1.aneg(G1.amul(G1.gen, (a * b + 1) % curve.r)), G2.gen)], nb)
    assert _run(["pairing-check", curve.name, good]).stdout.strip() == "ONE"
    assert _run(["pairing-check", curve.name, bad]).stdout.strip() == "NOT-ONE"
    # a point of the twist outside the r-torsion is refused (G2 of either curve has a cofactor)
    q = curve.q
    x = 1
    while True:
        y = _fq2_sqrt(((x * x * x + G2.b[0]) % q, G2.b[1] % q), q)
        if y is not None and G2.on_curve(((x, 0), y)) and G2.amul(((x, 0), y), curve.r) is not None:
            break
        x += 1
    f = str(tmp_path / "off.txt")
    _pairs_file(f, [(aP, ((x, 0), y))], nb)
    r = _run(["pairing-check", curve.name, f])
    assert r.returncode == 1 and "not in its group" in r.stderr


def _fq2_sqrt(a, q):
    """A square root in Fq[u]/(u^2 + 1), q = 3 mod 4, or None."""
    a0, a1 = a
    root = lambda t: (lambda r: r if r * r % q == t % q else None)(pow(t, (q + 1) // 4, q))
    if a1 == 0:
        r = root(a0)
        if r is not None:
            return (r, 0)
        r = root(-a0 % q)
        return None if r is None else (0, r)
    s = root((a0 * a0 + a1 * a1) % q)
    if s is None:
        return None
    for s_ in (s, q - s):
        x0 = root((a0 + s_) * pow(2, -1, q) % q)
        if x0:
            x1 = a1 * pow(2 * x0, -1, q) % q
            if (x0 * x0 - x1 * x1) % q == a0 % q:
                return (x0, x1)
    return None


def _program_files(d, curve):
    """def main(private field a, field b) -> (field, field): return a * b, a * b + b   with a = 7, b = 9"""
    prog = ir.Prog(curve, [ir.Parameter(1, True), ir.Parameter(2, False)], [
        ir.Other("Directive", {"span": None, "inputs": [], "outputs": [{"id": 3}], "solver": "ConditionEq"}),
        ir.Constraint([(1, 1)], [(2, 1)], [(3, 1)]),
        ir.Constraint([(0, 1)], [(3, 1)], [(-1, 1)]),
        ir.Constraint([(0, 1)], [(2, 1), (3, 1)], [(-2, 1)]),
    ], return_count=2)
    a, b = 7, 9
    open(os.path.join(d, "out"), "wb").write(ir.serialize_prog(prog))
    open(os.path.join(d, "witness"), "wb").write(ir.serialize_witness({0: 1, 1: a, 2: b, 3: a * b, -1: a * b, -2: a * b + b}))


def _oracle_verdict(curve, scheme, vk, proof):
    pts = (_g1_int(proof["proof"]["a"]), _g2_int(proof["proof"]["b"]), _g1_int(proof["proof"]["c"]))
    inputs = [int(s, 16) for s in proof["inputs"]]
    if scheme == "g16":
        ovk = dict(alpha_g1=_g1_int(vk["alpha"]), beta_g2=_g2_int(vk["beta"]), gamma_g2=_g2_int(vk["gamma"]), delta_g2=_g2_int(vk["delta"]),
                   gamma_abc_g1=[_g1_int(v) for v in vk["gamma_abc"]])
        return pairing.groth16_verify(curve, ovk, pts, inputs)
    ovk = dict(h_g2=_g2_int(vk["h"]), g_alpha_g1=_g1_int(vk["g_alpha"]), h_beta_g2=_g2_int(vk["h_beta"]), g_gamma_g1=_g1_int(vk["g_gamma"]),
               h_gamma_g2=_g2_int(vk["h_gamma"]), query=[_g1_int(v) for v in vk["query"]])
    return ogm17.verify(curve, ovk, pts, inputs)


@pytest.mark.parametrize("curve,scheme", [(BN254, "g16"), (BLS12_381, "g16"), (BN254, "gm17")],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_verify_accepts_own_proofs_and_rejects_mutations(tmp_path, curve, scheme):
    env = _env()
    d = str(tmp_path)
    nb = 32 if curve is BN254 else 48
    _program_files(d, curve)
    p = lambda name: os.path.join(d, name)
    r = _run(["setup", "-i", p("out"), "-p", p("proving.key"), "-v", p("verification.key"), "-s", scheme, "--entropy", "verify test"], env)
    assert r.returncode == 0, r.stderr
    r = _run(["generate-proof", "-i", p("out"), "-w", p("witness"), "-p", p("proving.key"), "-j", p("proof.json"), "-s", scheme, "--entropy", "x"], env)
    assert r.returncode == 0, r.stderr
    vk, proof = json.load(open(p("verification.key"))), json.load(open(p("proof.json")))

    def verdict(doc, vkdoc=None, name="case.json"):
        json.dump(doc, open(p(name), "w"))
        vkp = p("verification.key")
        if vkdoc is not None:
            vkp = p("vk_" + name)
            json.dump(vkdoc, open(vkp, "w"))
        r = _run(["verify", "-v", vkp, "-j", p(name)], env)
        return r

    r = verdict(proof)
    assert r.returncode == 0 and r.stdout.split("\n")[:2] == ["Performing verification...", "PASSED"], (r.stdout, r.stderr)
    assert _oracle_verdict(curve, scheme, vk, proof) is True
    G1, G2 = groups(curve)
    # the reference's own mutation (to_token.rs:68-71): a.x overwritten with 0xaa.. — not a point of the curve any more
    bad = json.loads(json.dumps(proof)); bad["proof"]["a"][0] = "0x" + "aa" * nb
    if int(bad["proof"]["a"][0], 16) < curve.q:
        assert verdict(bad).stdout.split()[-1] == "FAILED"
    # valid points, wrong proof: A + G, B + G2, C negated
    for which, mutated in (("a", _g1_hex(G1.aadd(_g1_int(proof["proof"]["a"]), G1.gen), nb)),
                           ("b", _g2_hex(G2.aadd(_g2_int(proof["proof"]["b"]), G2.gen), nb)),
                           ("c", _g1_hex(G1.aneg(_g1_int(proof["proof"]["c"])), nb))):
        bad = json.loads(json.dumps(proof)); bad["proof"][which] = mutated
        assert verdict(bad).stdout.split()[-1] == "FAILED", which
        if which == "a":
            assert _oracle_verdict(curve, scheme, vk, bad) is False
    # a different public input / a different output
    for i in (0, len(proof["inputs"]) - 1):
        bad = json.loads(json.dumps(proof)); bad["inputs"][i] = _hex(int(bad["inputs"][i], 16) + 1, 32)
        assert verdict(bad).stdout.split()[-1] == "FAILED", i
    # inputs may come without padding or prefix (T::try_from_str(s.trim_start_matches("0x"), 16))
    loose = json.loads(json.dumps(proof)); loose["inputs"] = [hex(int(s, 16))[2:] for s in proof["inputs"]]
    assert verdict(loose).stdout.split()[-1] == "PASSED"
    # a key for another statement
    other = json.loads(json.dumps(vk))
    q = "gamma_abc" if scheme == "g16" else "query"
    other[q][1], other[q][2] = other[q][2], other[q][1]
    assert verdict(proof, other).stdout.split()[-1] == "FAILED"
    # ---- failures the reference reports as errors (exit 1) ----
    r = verdict(dict(proof, curve="bls12_381" if curve is BN254 else "bn128"))
    assert r.returncode == 1 and "Expected the curve of the proof and the verification key to be equal" in r.stderr
    r = verdict(dict(proof, scheme="gm17" if scheme == "g16" else "g16"))
    assert r.returncode == 1 and "Expected the scheme of the proof and the verification key to be equal" in r.stderr
    r = verdict(dict(proof, inputs=proof["inputs"][:-1]))
    assert r.returncode == 1 and "public inputs" in r.stderr
    bad = json.loads(json.dumps(proof)); bad["proof"]["c"][1] = _hex(curve.q, nb)            # y = p: not canonical
    r = verdict(bad)
    assert r.returncode == 1 and "not below the field modulus" in r.stderr
    bad = json.loads(json.dumps(proof)); bad["inputs"][0] = _hex(curve.r, 32)
    r = verdict(bad)
    assert r.returncode == 1 and "not below the scalar field modulus" in r.stderr
    bad = json.loads(json.dumps(proof)); bad["proof"]["a"][0] = bad["proof"]["a"][0][2:]        # no 0x
    assert verdict(bad).returncode == 1
    no_curve = {k: v for k, v in proof.items() if k != "curve"}
    r = verdict(no_curve)
    assert r.returncode == 1 and "Field `curve` not found in proof" in r.stderr
    open(p("broken.json"), "w").write(json.dumps(proof)[:-5])
    r = _run(["verify", "-v", p("verification.key"), "-j", p("broken.json")], env)
    assert r.returncode == 1 and "Could not deserialize proof" in r.stderr
    r = _run(["verify", "-v", p("nowhere.key"), "-j", p("proof.json")], env)
    assert r.returncode == 1 and "Could not open" in r.stderr
    # ---- print-proof ----
    for fmt in ("json", "remix"):
        r = _run(["print-proof", "-j", p("proof.json"), "-f", fmt], env)
        if curve is not BN254:
            assert r.returncode == 1 and "only bn128 is supported" in r.stderr
            continue
        pts = proof["proof"]
        c = lambda v: json.dumps(v, separators=(",", ":"))
        if fmt == "json":
            want = c({"a": pts["a"], "b": pts["b"], "c": pts["c"]}) + "," + c(proof["inputs"]) + "\n"
        else:
            want = "[" + ", ".join(c(pts[k]) for k in ("a", "b", "c")) + "]," + c(proof["inputs"]) + "\n"
        assert r.returncode == 0 and r.stdout == want


def test_proof_json_round_trip_through_the_cpp_reader(tmp_path):
    """Proof::from_json reads what serde_json may legally write: any whitespace, any field order, escapes in strings."""
    env = _env()
    d = str(tmp_path)
    _program_files(d, BN254)
    p = lambda name: os.path.join(d, name)
    assert _run(["setup", "-i", p("out"), "-p", p("proving.key"), "-v", p("verification.key"), "--entropy", "rt"], env).returncode == 0
    assert _run(["generate-proof", "-i", p("out"), "-w", p("witness"), "-p", p("proving.key"), "-j", p("proof.json"), "--entropy", "rt"], env).returncode == 0
    proof, vk = json.load(open(p("proof.json"))), json.load(open(p("verification.key")))
    shuffled = {k: proof[k] for k in ("inputs", "proof", "curve", "scheme")}
    shuffled["proof"] = {k: proof["proof"][k] for k in ("c", "b", "a")}
    shuffled["extra"] = {"ignored": [1, 2.5e3, True, None, "é\n"]}
    open(p("compact.json"), "w").write(json.dumps(shuffled, separators=(",", ":")))
    open(p("spaced.json"), "w").write(json.dumps(shuffled, indent=7).replace("0x", "\\u0030x", 1))
    open(p("vk_compact.key"), "w").write(json.dumps({k: vk[k] for k in reversed(list(vk))}, separators=(",", ":")))
    for name in ("compact.json", "spaced.json"):
        r = _run(["verify", "-v", p("vk_compact.key"), "-j", p(name)], env)
        assert r.returncode == 0 and r.stdout.split()[-1] == "PASSED", (name, r.stderr)


def test_mutated_files_never_crash_the_verifier(tmp_path):
    """tests/host/verify_fuzz.cpp under ASan + UBSan: 120 byte-level mutations of a (verification key, proof) pair; a verdict or an
    Error every time, and PASSED only when the parsed values are the original ones."""
    env = _env()
    d = str(tmp_path)
    _program_files(d, BN254)
    p = lambda name: os.path.join(d, name)
    assert _run(["setup", "-i", p("out"), "-p", p("proving.key"), "-v", p("verification.key"), "--entropy", "fz"], env).returncode == 0
    assert _run(["generate-proof", "-i", p("out"), "-w", p("witness"), "-p", p("proving.key"), "-j", p("proof.json"), "--entropy", "fz"], env).returncode == 0
    exe = p("verify_fuzz")
    root = os.path.dirname(HERE)
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17",
                           os.path.join(HERE, "host", "verify_fuzz.cpp"), os.path.join(root, "zokrates_amd", "csrc", "host", "verify.cpp"), "-o", exe])
    r = subprocess.run([exe, p("verification.key"), p("proof.json"), "120", "11"], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.startswith("verified ") and "errors" in r.stdout


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_generate_proof_with_self_check(tmp_path, scheme):
    """`zkhip-cli generate-proof --verify`: the proof is checked against the verification key at the head of the proving key before
    the process reports success; an assignment that does not satisfy the program yields no proof.json and exit status 1."""
    env = _env()
    d = str(tmp_path)
    _program_files(d, BN254)
    p = lambda name: os.path.join(d, name)
    assert _run(["setup", "-i", p("out"), "-p", p("proving.key"), "-v", p("verification.key"), "-s", scheme, "--entropy", "sc"], env).returncode == 0
    r = _run(["generate-proof", "-i", p("out"), "-w", p("witness"), "-p", p("proving.key"), "-j", p("proof.json"), "-s", scheme, "--verify", "--timings"], env)
    assert r.returncode == 0 and "verified against the verification key" in r.stdout, (r.stdout, r.stderr)
    tm = json.loads([l for l in r.stdout.splitlines() if l.startswith("timings ")][0][8:])
    assert tm["verify_ms"] > 0
    # the key's own verification key is the one `setup` wrote
    assert _run(["verify", "-v", p("verification.key"), "-j", p("proof.json")], env).stdout.split()[-1] == "PASSED"
    a, b = 7, 9
    open(p("bad_witness"), "wb").write(ir.serialize_witness({0: 1, 1: a, 2: b, 3: a * b + 1, -1: a * b, -2: a * b + b}))
    r = _run(["generate-proof", "-i", p("out"), "-w", p("bad_witness"), "-p", p("proving.key"), "-j", p("bad_proof.json"), "-s", scheme, "--verify"], env)
    assert r.returncode == 1 and "does not verify" in r.stderr and not os.path.exists(p("bad_proof.json")), (r.stdout, r.stderr)
    # without the self check the same run writes a proof — which `verify` then refuses
    r = _run(["generate-proof", "-i", p("out"), "-w", p("bad_witness"), "-p", p("proving.key"), "-j", p("bad_proof.json"), "-s", scheme], env)
    assert r.returncode == 0
    assert _run(["verify", "-v", p("verification.key"), "-j", p("bad_proof.json")], env).stdout.split()[-1] == "FAILED"


def test_reference_gm17_artefacts_over_bls12_377(tmp_path, golden_dir):
    """Proofs the reference itself generated (ark backend, GM17, BLS12-377) through the compiled verifier: PASSED as they are,
    FAILED with a changed input, FAILED with the two halves of B's coordinates swapped ([c1, c0] is not the encoding)."""
    p = lambda name: os.path.join(str(tmp_path), name)
    cases = []
    d = json.load(open(os.path.join(golden_dir, "gm17_bls12_377_triple.json")))
    assert d["curve"] == "bls12_377" and d["expected"] is True
    cases.append(("stdlib_gm17", dict(d["vk"]), d["proof"], d["inputs"]))
    e = json.load(open(os.path.join(golden_dir, "gm17_bls12_377_embed_triples.json")))
    assert e["curve"] == "bls12_377" and len(e["triples"]) == 3
    h48 = lambda v: _hex(int(v), 48)
    g1 = lambda a, i: [h48(a[i]), h48(a[i + 1])]
    g2 = lambda a, i: [[h48(a[i]), h48(a[i + 1])], [h48(a[i + 2]), h48(a[i + 3])]]
    for k, t in enumerate(e["triples"]):
        pr, v = t["proof"], t["vk"]
        n_in = len(t["inputs"])
        vk = {"h": g2(v, 0), "g_alpha": g1(v, 4), "h_beta": g2(v, 6), "g_gamma": g1(v, 10), "h_gamma": g2(v, 12),
              "query": [g1(v, 16 + 2 * i) for i in range(n_in + 1)]}
        cases.append(("core_test_%d" % k, vk, {"a": g1(pr, 0), "b": g2(pr, 2), "c": g1(pr, 6)}, [_hex(int(x), 32) for x in t["inputs"]]))
    env = _env()
    for name, vk, pts, inputs in cases:
        vk = dict(vk, scheme="gm17", curve="bls12_377")
        json.dump(vk, open(p(name + ".key"), "w"))

        def verdict(points, ins):
            json.dump({"scheme": "gm17", "curve": "bls12_377", "proof": points, "inputs": ins}, open(p(name + ".json"), "w"))
            r = _run(["verify", "-v", p(name + ".key"), "-j", p(name + ".json")], env)
            assert r.returncode == 0, (name, r.stderr)
            return r.stdout.split()[-1]

        assert verdict(pts, inputs) == "PASSED", name
        changed = list(inputs)
        changed[-1] = _hex(int(changed[-1], 16) + 1, 32)
        assert verdict(pts, changed) == "FAILED", name
        swapped = dict(pts, b=[pts["b"][0][::-1], pts["b"][1][::-1]])
        assert verdict(swapped, inputs) == "FAILED", name
