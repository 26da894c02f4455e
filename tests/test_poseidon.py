"""The Poseidon hash-chain workload of BASELINE.json configs[3] (zokrates_amd/poseidon.py): parameters regenerated with
the Grain LFSR reproduce the reference's hash known answers; the generated R1CS is satisfied by the generated witness and
agrees with a term-by-term construction; `-m gpu`: the depth-1024 BLS12-381 chain proves bit-identically to the oracle."""
import numpy as np
import pytest

from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import poseidon


def test_reference_hash_kats():
    """/root/reference/zokrates_stdlib/tests/tests/hashes/poseidon/poseidon_{1,2,3}.json; the first round constant and
    MDS entry of the reference's constants.zok (stdlib/hashes/poseidon/constants.zok:3, :1848)."""
    assert poseidon.poseidon([1]) == 18586133768512220936620570745912940619677854269274689475585506675881198879027
    assert poseidon.poseidon([42]) == 12326503012965816391338144612242952408728683609716147019497703475006801258307
    assert poseidon.poseidon([1, 2]) == 7853200120776062878684798364095072458815029376092732009249414926327459813530
    assert poseidon.poseidon([1, 2, 3]) == 6542985608222806190361240322586112750744169038454362455181422643027100751666
    consts, mds = poseidon.parameters(2)
    assert consts[0] == 4417881134626180770308697923359573201005643519861877412381846989312604493735
    assert mds[0][0] == 2910766817845651019878574839501801340070030115151021261302834310722729507541
    assert len(poseidon.parameters(3)[0]) == 3 * 65


def _rows(mat, n):
    rp, col, val = mat
    return [[(int(col[q]), int.from_bytes(val[32 * q:32 * q + 32].tobytes(), "little")) for q in range(int(rp[i]), int(rp[i + 1]))] for i in range(n)]


@pytest.mark.parametrize("curve", [BN254, BLS12_381], ids=lambda c: c.name)
@pytest.mark.parametrize("depth", [1, 2, 5])
def test_chain_r1cs(curve, depth):
    ch = poseidon.chain(curve.curve_id, depth)
    assert (ch.l, ch.w, ch.n) == (3, 243 * depth, 243 * depth + 1)
    cs = g16.R1CS(l=ch.l, w=ch.w)
    cs.A, cs.B, cs.C = (_rows(m, ch.n) for m in ch.mats())
    for seed in (1, 2):
        z = ch.assignment(seed)
        zi = [int.from_bytes(z[32 * j:32 * j + 32].tobytes(), "little") for j in range(ch.m)]
        assert cs.is_satisfied(zi, curve.r)
        # the public output is the chain of hashes
        s = zi[1]
        for h in range(depth):
            s = poseidon.poseidon([s, h], curve.r)
        assert zi[2] == s
        bad = list(zi)
        bad[2] = (bad[2] + 1) % curve.r
        assert not cs.is_satisfied(bad, curve.r)
    if depth == 5:      # (hashes >= 2 are hash 1 shifted: the satisfiability check above covers the replication)
        assert ch.mats()[0][1].max() < ch.m and ch.mats()[1][1].max() < ch.m
        assert max(len(r) for r in cs.A) > 50            # the partial rounds' wide combinations are there


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_emu_wide_rows(scheme):
    """The grouped mat-vec (several work-items per row + LDS tree) on the TEST-ONLY emulator: a depth-2 chain has rows
    of up to ~60 terms, so A and B run with G > 1; witness map and proofs must equal the oracle's."""
    from emu_util import emu_library
    from oracle import cpu, gm17
    from zokrates_amd import native
    curve = BN254
    ch = poseidon.chain(0, 2)
    assert int(ch.mats()[1][0][-1]) // ch.n >= 8           # average row length of B: G = 2 at least
    ctx = native.Context(0, emu_library())
    z = ch.assignment(3)
    cs = native.ConstraintSystem(ctx, 0, ch.n, ch.l, ch.w, ch.mats())
    oc = cpu.Circuit.from_csr(0, ch.n, ch.l, ch.w, ch.mats())
    if scheme == "g16":
        assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
        tox = g16.Toxic.from_seed(curve)
        tb = cpu.toxic_bytes(tox)
        opk = cpu.ProvingKey.setup(oc, tb)
        pk = native.ProvingKey(ctx, 0, opk.serialize())
        assert native.prove_g16(ctx, pk, cs, z, 5, 6) == cpu.trapdoor(oc, tb, z, 5, 6)
    else:
        tox = gm17.Toxic.from_seed(curve)
        tb = cpu.gm17_toxic_bytes(tox)
        cpk = cpu.Gm17ProvingKey.setup(oc, tb)
        pk = native.ProvingKey(ctx, 0, cpk.serialize(), scheme="gm17")
        assert native.prove_gm17(ctx, pk, cs, z, 5, 6, 7) == cpu.gm17_trapdoor(oc, tb, z, 5, 7)
    ctx.close()


@pytest.mark.gpu
def test_gpu_poseidon_chain_bls12_381():
    from oracle import cpu
    from zokrates_amd import native, synth
    ctx = native.Context(0)
    ch = poseidon.chain(1, 1024)
    assert ch.N == 1 << 18
    z = ch.assignment(0x5EED)
    cs = native.ConstraintSystem(ctx, 1, ch.n, ch.l, ch.w, ch.mats())
    tox = synth.toxic_waste(1)
    raw = native.setup_g16(ctx, cs, tox)
    pk = native.ProvingKey(ctx, 1, raw)
    got = native.prove_g16(ctx, pk, cs, z, 1234567, 7654321)
    oc = cpu.Circuit.from_csr(1, ch.n, ch.l, ch.w, ch.mats())
    opk = cpu.ProvingKey.parse(1, raw)
    want, _ = cpu.prove(oc, opk, z, 1234567, 7654321)
    assert got == want
    assert got == cpu.trapdoor(oc, b"".join(int(v).to_bytes(32, "little") for v in tox), z, 1234567, 7654321)
    ctx.close()
