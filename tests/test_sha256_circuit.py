"""The SHA-256 workload of BASELINE.json configs[0] (zokrates_amd/sha256_circuit.py: stdlib `hashes/sha256/512bitPacked.zok` the way
the reference's uint optimizer, flattener and redefinition optimizer shape it).  The reference's own known answer for that
program (zokrates_stdlib/tests/tests/hashes/sha256/512bitPacked.json: [0, 0, 0, 5] -> two field elements) must come out of the
WITNESS of the generated constraint system, the system must be satisfied by it and by nothing with a wrong digest; `-m gpu`:
three hashes (rows of up to 7 000 terms in C) prove bit-identically to the oracle."""
import hashlib

import numpy as np
import pytest

from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import sha256_circuit as sha

import json
import os

_KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sha256_packed_kat.json")))   # tests/golden/make_golden.py
REFERENCE_KAT = ([int(v) for v in _KAT["input"]], [int(v) for v in _KAT["output"]])
assert REFERENCE_KAT == ([0, 0, 0, 5], [263561599766550617289250058199814760685, 65303172752238645975888084098459749904])


def _rows(mat, n):
    rp, col, val = mat
    rp, col = rp.tolist(), col.tolist()
    vals = [int.from_bytes(val[32 * q:32 * q + 32].tobytes(), "little") for q in range(len(col))]
    return [[(col[q], vals[q]) for q in range(rp[i], rp[i + 1])] for i in range(n)]


def _ints(z, m):
    return [int.from_bytes(z[32 * j:32 * j + 32].tobytes(), "little") for j in range(m)]


def test_the_function_is_sha256():
    assert sha.sha256_packed(REFERENCE_KAT[0]) == REFERENCE_KAT[1]
    d = hashlib.sha256(bytes(63) + b"\5").hexdigest()       # the comment in the reference's 512bitPacked.zok test
    assert d == "c6481e22c5ff4164af680b8cfaa5e8ed3120eeff89c4f307c4a6faaae059ce10"


def test_shape_of_one_hash():
    """What the reference's rules give for one call: the count is a consequence, pinned here so that a change of the rules shows."""
    rows, tape, nvar, _ = sha.template()
    assert (len(rows), nvar) == (48972, 48654)
    # 4 x (128 bit checks + 1 sum), 2 outputs; per compression: 64 x (32 ch + 64 maj + 4 x 32 xor) and the decompositions
    bool_rows = sum(1 for a, b, c in rows if a == b == c and len(a) == 1)
    widths = [op[2] for op in tape if op[0] == "bits"]
    assert bool_rows == 4 * 128 + sum(widths)             # every bit of every decomposition is checked, nothing else is
    # the maxima of the lazy sums grow like a Fibonacci sequence with a lag of four rounds (e' = d + h + ..., a' = h + ...: both
    # operands are unreduced values of earlier rounds), from 34 bits in the message schedule to 58 at the end of the second block
    assert min(widths) == 33 and max(widths) == 58 and widths[0] == 34
    assert max(len(c) for _, _, c in rows) > 5000           # the inlined lazy sums: the widest rows of any workload here


def test_generated_bytes_are_pinned():
    """The matrices of one call over bn128 and the assignment for seed 1, as digests: bench lines and proofs made from this
    generator stay comparable across rounds only while it emits the same bytes (a deliberate change of the rules updates these)."""
    c = sha.circuit(0, 1)
    h = hashlib.sha256()
    for rp, col, val in c.mats():
        for a in (rp, col, val):
            h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == "23e20149c76f28f719889e135d2728701953bef7d1c5c7816a970c6ca8eeed6b"
    assert hashlib.sha256(c.assignment(1).tobytes()).hexdigest() == "dba1e8c2fa5911b4e24b468cddbe7093610995d80ae7723265e31c93c672ad92"


@pytest.mark.parametrize("curve", [BN254, BLS12_381], ids=lambda c: c.name)
def test_r1cs_and_witness(curve):
    H = 2 if curve is BN254 else 1          # (the second hash is the first one shifted: once is enough)
    c = sha.circuit(curve.curve_id, H)
    rows, _, nvar, _ = sha.template()
    assert (c.l, c.w, c.n, c.N) == (1 + 6 * H, H * nvar, H * len(rows), 1 << (16 + H - 1))
    cs = g16.R1CS(l=c.l, w=c.w)
    cs.A, cs.B, cs.C = (_rows(m, c.n) for m in c.mats())
    pre = [REFERENCE_KAT[0], [(1 << 128) - 1, 0x0123456789abcdef << 60, 1, 0]][:H]
    z = c.values(pre)
    zi = _ints(z, c.m)
    assert cs.is_satisfied(zi, curve.r)
    assert zi[:c.l] == [1] + [v for p_ in pre for v in p_] + [v for p_ in pre for v in sha.sha256_packed(p_)]     # ark order: ONE, inputs, outputs
    assert zi[1 + 4 * H:3 + 4 * H] == REFERENCE_KAT[1]
    # every wire behind the public part is a bit
    assert set(np.unique(z.reshape(-1, 32)[c.l:, 0]).tolist()) <= {0, 1} and not z.reshape(-1, 32)[c.l:, 1:].any()
    for j in (1 + 4 * H, c.l - 1, c.l, c.l + 128 * 4 + 40, c.m - 1):      # a wrong digest, a flipped wire: not a witness
        bad = list(zi)
        bad[j] = (bad[j] + 1) % curve.r
        assert not cs.is_satisfied(bad, curve.r), j
    # the assignment() of the bench: seeded preimages
    zs = _ints(c.assignment(0x5EED), c.m)
    assert cs.is_satisfied(zs, curve.r) and zs[1 + 4 * H:3 + 4 * H] == sha.sha256_packed(zs[1:5])


def test_columns_are_in_generate_constraints_order():
    """The system written as a ZoKrates `out` program and read back by the product's reader (which allocates columns the way
    ark's generate_constraints does: arguments, then every variable when a constraint first names it) is the system itself —
    same columns, same matrices: a key made for one fits the other (the CLI-shaped legs of bench.py rely on it)."""
    from emu_util import emu_library
    from zokrates_amd import native
    lib = emu_library()
    c = sha.circuit(0, 1)
    ids = np.arange(c.m, dtype=np.int64)
    out = native.write_program(0, c.n, c.m, c.mats(), ids=ids, args=[(j, False) for j in range(1, c.l)], library=lib)
    prog = native.Program(out, lib)
    assert (prog.n, prog.l, prog.w) == (c.n, c.l, c.w)
    assert (prog.variable_order() == ids).all()
    for (rp, col, val), (rp2, col2, val2) in zip(c.mats(), prog.mats()):
        assert (rp == rp2).all() and (col == col2).all() and (np.asarray(val).reshape(-1) == np.asarray(val2).reshape(-1)).all()


def test_witness_map_and_proof_on_the_emulator():
    """One hash through the product's sparse mat-vec and prover on the TEST-ONLY emulator (domain 2^16, rows of thousands of
    terms next to one-term rows), against the C++ oracle."""
    from emu_util import emu_library
    from oracle import cpu
    from zokrates_amd import native
    c = sha.circuit(0, 1)
    ctx = native.Context(0, emu_library())
    try:
        z = c.assignment(11)
        cs = native.ConstraintSystem(ctx, 0, c.n, c.l, c.w, c.mats())
        oc = cpu.Circuit.from_csr(0, c.n, c.l, c.w, c.mats())
        assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_sha256_three_hashes():
    from oracle import cpu
    from zokrates_amd import native, synth
    ctx = native.Context(0)
    c = sha.circuit(0, 3)
    assert c.N == 1 << 18
    z = c.assignment(0x5EED)
    cs = native.ConstraintSystem(ctx, 0, c.n, c.l, c.w, c.mats())
    oc = cpu.Circuit.from_csr(0, c.n, c.l, c.w, c.mats())
    assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
    tox = synth.toxic_waste(0)
    raw = native.setup_g16(ctx, cs, tox)
    pk = native.ProvingKey(ctx, 0, raw)
    got = native.prove_g16(ctx, pk, cs, z, 1234567, 7654321)
    assert got == cpu.trapdoor(oc, b"".join(int(v).to_bytes(32, "little") for v in tox), z, 1234567, 7654321)
    opk = cpu.ProvingKey.parse(0, raw)
    want, _ = cpu.prove(oc, opk, z, 1234567, 7654321)
    assert got == want
    pk.close()
    ctx.close()
