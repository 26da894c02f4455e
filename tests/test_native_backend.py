"""The compiled host side (include/zkhip_backend.hpp, csrc/host/backend.cpp, the `zkhip-cli` executable): C++ above the C
ABI with the reference's names and behaviour (`Backend::generate_proof`, `Proof`, `get_rng_from_entropy`: DESIGN.md §1).
Checked against the Python host layer (zokrates_amd/cli.py, rng.py, formats.py) — two implementations of the same
reference files — and against the published known answers of the (r, s) chain."""
import hashlib
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import cpu, ir
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native, rng, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_rng_chain_of_the_cpp_host_layer(tmp_path):
    exe = str(tmp_path / "backend_kats")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(HERE, "host", "backend_kats.cpp"),
                           os.path.join(ROOT, "zokrates_amd", "csrc", "host", "backend.cpp"), "-L" + os.path.join(HERE, "_emu"), "-lzkhip_emu",
                           "-Wl,-rpath," + os.path.join(HERE, "_emu"), "-pthread", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    got = dict(line.split() for line in out.splitlines())
    long = bytes((i * 7 + 1) & 0xff for i in range(300))
    assert got["blake2b_abc"] == hashlib.blake2b(b"abc").hexdigest() == ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
                                                                         "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")   # RFC 7693 App. A
    assert got["blake2b_empty"] == hashlib.blake2b(b"").hexdigest()
    assert got["blake2b_300"] == hashlib.blake2b(long).hexdigest()
    assert got["blake2b_128"] == hashlib.blake2b(long[:128]).hexdigest()          # exactly one block: it is the LAST block
    assert got["blake2b_256"] == hashlib.blake2b(long[:256]).hexdigest()
    tc1_12 = "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be"
    assert got["chacha12_zero_first15"] == tc1_12[:120]                            # draft-strombergson-chacha-test-vectors-01, TC1, 12 rounds
    nxt = rng.chacha_block([0] * 8, 1, 0, 12)
    assert got["chacha12_zero_straddle"] == (bytes.fromhex(tc1_12)[60:64] + struct.pack("<I", nxt[0])).hex()
    for curve in (0, 1):
        for tag, ent in (("bench", "bench"), ("golden", "golden vector 1"), ("empty", "")):
            g = rng.rng_from_entropy(ent)
            for k in range(3):
                assert got["fr_rand_%d_%s_%d" % (curve, tag, k)] == rng.fr_rand(g, curve).to_bytes(32, "little").hex(), (curve, tag, k)


def _files(tmp_path, curve, lib, scheme):
    """A compiler-shaped program (def main(private field a, field b) -> (field, field): return a * b, a * b + b) with its
    witness and a key from the device setup."""
    prog = ir.Prog(curve, [ir.Parameter(1, True), ir.Parameter(2, False)], [
        ir.Other("Directive", {"span": None, "inputs": [], "outputs": [{"id": 3}], "solver": "ConditionEq"}),
        ir.Constraint([(1, 1)], [(2, 1)], [(3, 1)]),
        ir.Constraint([(0, 1)], [(3, 1)], [(-1, 1)]),
        ir.Constraint([(0, 1)], [(2, 1), (3, 1)], [(-2, 1)]),
    ], return_count=2)
    a, b = 7, 9
    paths = {k: str(tmp_path / k) for k in ("out", "witness", "proving.key", "proof_py.json", "proof_cpp.json", "cache")}
    open(paths["out"], "wb").write(ir.serialize_prog(prog))
    open(paths["witness"], "wb").write(ir.serialize_witness({0: 1, 1: a, 2: b, 3: a * b, -1: a * b, -2: a * b + b}))
    ctx = native.Context(0, lib)
    p = native.Program(open(paths["out"], "rb").read(), lib)
    cs = p.constraint_system(ctx)
    tox = synth.toxic_waste(curve.curve_id)
    pk = native.setup_gm17(ctx, cs, (tox[0], tox[1], tox[2], tox[4])) if scheme == "gm17" else native.setup_g16(ctx, cs, tox)
    pk.tofile(paths["proving.key"])
    ctx.close()
    return paths


def _both_clis(paths, scheme, exe, env):
    from zokrates_amd import cli
    common = ["-i", paths["out"], "-w", paths["witness"], "-p", paths["proving.key"], "-s", scheme, "--entropy", "same entropy"]
    py = subprocess.run([os.sys.executable, "-m", "zokrates_amd.cli", "generate-proof"] + common + ["-j", paths["proof_py.json"]],
                        capture_output=True, text=True, cwd=ROOT, env=env)
    assert py.returncode == 0, py.stderr
    for extra in ([], ["--key-cache", paths["cache"]], ["--key-cache", paths["cache"], "--timings"]):
        cpp = subprocess.run([exe, "generate-proof"] + common + ["-j", paths["proof_cpp.json"]] + extra, capture_output=True, text=True, env=env)
        assert cpp.returncode == 0, cpp.stderr
        assert open(paths["proof_cpp.json"]).read() == open(paths["proof_py.json"]).read(), extra
        if "--timings" in extra:
            tm = json.loads([l for l in cpp.stdout.splitlines() if l.startswith("timings ")][0][8:])
            assert tm["key_source"] == "image" and tm["constraints"] == 3
    # setup: both host layers draw the toxic waste the same way -> the same proving.key and verification.key, byte for byte
    outs = {}
    for who, cmd in (("py", [os.sys.executable, "-m", "zokrates_amd.cli", "setup"]), ("cpp", [exe, "setup"])):
        pkp, vkp = paths["proving.key"] + "." + who, paths["proving.key"] + ".vk." + who
        r = subprocess.run(cmd + ["-i", paths["out"], "-p", pkp, "-v", vkp, "-s", scheme, "--entropy", "setup entropy"], capture_output=True, text=True,
                           cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr
        outs[who] = (open(pkp, "rb").read(), open(vkp).read())
    assert outs["py"][0] == outs["cpp"][0] and outs["py"][1] == outs["cpp"][1]
    assert json.loads(outs["cpp"][1])["scheme"] == scheme
    # the whole trait in one chain: setup -> generate-proof -> verify (csrc/host/verify.cpp, no GPU in that step) says PASSED,
    # and FAILED once a public input is changed
    chain = paths["proof_cpp.json"] + ".chain"
    r = subprocess.run([exe, "generate-proof", "-i", paths["out"], "-w", paths["witness"], "-p", paths["proving.key"] + ".cpp", "-j", chain, "-s", scheme],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "verify", "-v", paths["proving.key"] + ".vk.cpp", "-j", chain], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.split()[-1] == "PASSED", (r.stdout, r.stderr)
    forged = json.load(open(chain))
    forged["inputs"][-1] = "0x" + (int(forged["inputs"][-1], 16) ^ 1).to_bytes(32, "big").hex()
    json.dump(forged, open(chain, "w"))
    r = subprocess.run([exe, "verify", "-v", paths["proving.key"] + ".vk.cpp", "-j", chain], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.split()[-1] == "FAILED", (r.stdout, r.stderr)
    doc = json.load(open(paths["proof_cpp.json"]))
    assert doc["scheme"] == scheme and len(doc["inputs"]) == 3 and doc["inputs"][0] == "0x" + (9).to_bytes(32, "big").hex()
    # failures: message on stderr, exit status 1 (the reference's panic hook + exit(1))
    bad = paths["witness"] + ".bad"
    open(bad, "wb").write(open(paths["witness"], "rb").read()[:-40])
    cpp = subprocess.run([exe, "generate-proof", "-i", paths["out"], "-w", bad, "-p", paths["proving.key"], "-j", paths["proof_cpp.json"], "-s", scheme],
                         capture_output=True, text=True, env=env)
    assert cpp.returncode == 1 and "zkhip-cli:" in cpp.stderr


@pytest.mark.parametrize("curve,scheme", [(BN254, "g16"), (BLS12_381, "g16"), (BN254, "gm17")], ids=lambda v: getattr(v, "name", str(v)))
def test_native_cli_equals_python_cli_on_emulator(tmp_path, curve, scheme):
    from emu_util import EMU_LIB, emu_library
    lib = emu_library()
    exe = os.path.join(HERE, "_emu", "zkhip-cli-emu")
    assert os.path.exists(exe)
    _both_clis(_files(tmp_path, curve, lib, scheme), scheme, exe, dict(os.environ, ZKHIP_LIBRARY=EMU_LIB))


@pytest.mark.parametrize("curve,scheme", [(BN254, "g16"), (BLS12_381, "g16"), (BN254, "gm17")], ids=lambda v: getattr(v, "name", v))
def test_long_lived_prover_of_the_cpp_host_layer_on_emulator(tmp_path, curve, scheme):
    """zokrates_js calls generate_proof many times per process (zokrates_js/src/lib.rs:380-452): the compiled host side keeps the
    constraint system resident (System) next to the key and binds the key (Groth16 or GM17) to it (Hip::bind = zkhip_pk_bind_r1cs).  Same
    proof.json text before and after the binding, through the one-call form, and from the `generate-proof` executable."""
    from emu_util import emu_library
    lib = emu_library()
    paths = _files(tmp_path, curve, lib, scheme)
    exe = str(tmp_path / "resident_prover")
    emu_dir = os.path.join(HERE, "_emu")
    host = os.path.join(ROOT, "zokrates_amd", "csrc", "host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(HERE, "host", "resident_prover.cpp"), os.path.join(host, "backend.cpp"),
                           os.path.join(host, "verify.cpp"), "-L" + emu_dir, "-lzkhip_emu", "-Wl,-rpath," + emu_dir, "-o", exe])
    r = subprocess.run([exe, paths["out"], paths["witness"], paths["proving.key"], "same entropy", scheme], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    head, same, text = r.stdout.split("\n", 2)
    assert same == "same=1"
    assert head == "bound=1 refused=0 is_bound=1"          # both schemes bind (GM17: two transforms per proof afterwards)
    cli = subprocess.run([os.path.join(emu_dir, "zkhip-cli-emu"), "generate-proof", "-i", paths["out"], "-w", paths["witness"], "-p", paths["proving.key"],
                          "-s", scheme, "--entropy", "same entropy", "-j", paths["proof_cpp.json"]], capture_output=True, text=True)
    assert cli.returncode == 0, cli.stderr
    assert json.loads(open(paths["proof_cpp.json"]).read()) == json.loads(text)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,scheme", [(BN254, "g16"), (BLS12_381, "gm17")], ids=lambda v: getattr(v, "name", str(v)))
def test_native_cli_equals_python_cli_on_gpu(tmp_path, curve, scheme):
    exe = os.path.join(ROOT, "zokrates_amd", "zkhip-cli")
    assert os.path.exists(exe), "python -m zokrates_amd.build builds it next to libzkhip.so"
    _both_clis(_files(tmp_path, curve, native.default_library(), scheme), scheme, exe, dict(os.environ))
