"""Pins the C++ oracle (oracle/c, the CPU baseline) bit-for-bit against the python big-int oracle."""
import json
import os
import random

import numpy as np
import pytest

from oracle import cpu, formats
from oracle import groth16 as g16
from oracle.curves import groups
from oracle.fields import BN254, BLS12_381, inv

CURVES = [BN254, BLS12_381]


def le32(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_field_ops(curve, golden_dir):
    rnd = random.Random(1)
    for field, p in ((0, curve.r), (1, curve.q)):
        edge = [0, 1, 2, p - 1, p - 2, (1 << (p.bit_length() - 1)), pow(2, 64 * ((p.bit_length() + 63) // 64), p)]
        vals = edge + [rnd.randrange(p) for _ in range(20)]
        for a in vals:
            for b in vals[:10]:
                assert cpu.field_op(curve.curve_id, field, "add", a, b) == (a + b) % p
                assert cpu.field_op(curve.curve_id, field, "sub", a, b) == (a - b) % p
                assert cpu.field_op(curve.curve_id, field, "mul", a, b) == a * b % p
            if a:
                assert cpu.field_op(curve.curve_id, field, "inv", a) == inv(a, p)
    if curve is BN254:
        d = json.load(open(os.path.join(golden_dir, "bn128_field_kats.json")))
        r = curve.r
        for op, a, b, e in d["kats"]:
            if op in ("add", "sub", "mul"):
                assert cpu.field_op(0, 0, op, int(a) % r, int(b) % r) == int(e)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_generators_and_synth_stream(curve):
    g1, g2 = cpu.generators(curve.curve_id)
    assert g1 == formats.ser_g1(curve, curve.g1) and g2 == formats.ser_g2(curve, curve.g2)
    for kind in ("dense", "sha"):
        cs, z = g16.synthetic_chain(curve, 37, 0x5EED0001, kind)
        c = cpu.Circuit.synth(curve.curve_id, 37, 0x5EED0001, kind)
        assert (c.n, c.l, c.w, c.N) == (cs.n, cs.l, cs.w, cs.domain_size())
        assert c.assignment().tobytes() == le32(z).tobytes()
        for which, M in enumerate((cs.A, cs.B, cs.C)):
            rp, col, val = c.csr(which)
            flat = [e for row in M for e in row]
            assert list(rp) == list(np.cumsum([0] + [len(row) for row in M]))
            assert list(col) == [j for j, _ in flat]
            assert val.tobytes() == le32([v for _, v in flat]).tobytes()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_ntt_matches_python(curve):
    rnd = random.Random(2)
    for logn in (0, 1, 3, 6):
        N = 1 << logn
        a = [rnd.randrange(curve.r) for _ in range(N)]
        dom = g16.Domain(curve, N)
        for name, fn in (("fft", dom.fft), ("ifft", dom.ifft), ("coset_fft", dom.coset_fft), ("coset_ifft", dom.coset_ifft)):
            got = cpu.ntt(curve.curve_id, le32(a), name)
            assert got.tobytes() == le32(fn(a)).tobytes(), (logn, name)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_matches_python(curve):
    rnd = random.Random(3)
    G1, G2 = groups(curve)
    for n in (1, 5, 40):   # 40 >= 32 takes the ln-heuristic window size
        ks = [rnd.randrange(curve.r) for _ in range(n)]
        ks[0] = 0
        if n > 2:
            ks[1] = 1; ks[2] = curve.r - 1
        p1 = [G1.amul(G1.gen, rnd.randrange(1, curve.r)) for _ in range(n)]
        p2 = [G2.amul(G2.gen, rnd.randrange(1, curve.r)) for _ in range(n)]
        if n > 4:
            p1[3] = None; p2[4] = None; p1[4] = p1[2]; ks[4] = ks[2]  # infinity base, repeated base
        e1 = G1.to_affine(G1.msm(p1, ks)); e2 = G2.to_affine(G2.msm(p2, ks))
        b1 = np.frombuffer(b"".join(formats.ser_g1(curve, P) for P in p1), dtype=np.uint8)
        b2 = np.frombuffer(b"".join(formats.ser_g2(curve, P) for P in p2), dtype=np.uint8)
        nb = curve.fq_bytes
        got1 = cpu.msm(curve.curve_id, 1, b1, le32(ks))
        got2 = cpu.msm(curve.curve_id, 2, b2, le32(ks))
        enc1 = lambda P: bytes(2 * nb) + b"\1" if P is None else b"".join(int(v).to_bytes(nb, "little") for v in P) + b"\0"
        enc2 = lambda P: bytes(4 * nb) + b"\1" if P is None else b"".join(
            int(v).to_bytes(nb, "little") for v in (P[0][0], P[0][1], P[1][0], P[1][1])) + b"\0"
        assert got1 == enc1(e1)
        assert got2 == enc2(e2)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("kind", ["dense", "sha"])
def test_setup_prove_trapdoor_match_python(curve, kind):
    n, seed = 13, 0x5EED0002
    cs, z = g16.synthetic_chain(curve, n, seed, kind)
    tox = g16.Toxic.from_seed(curve)
    pk_py, _ = g16.setup(curve, cs, tox)
    c = cpu.Circuit.synth(curve.curve_id, n, seed, kind)
    pk = cpu.ProvingKey.setup(c, cpu.toxic_bytes(tox), threads=3)
    raw = pk.serialize().tobytes()
    assert raw == formats.ark_pk_serialize(curve, pk_py)
    pk2 = cpu.ProvingKey.parse(curve.curve_id, raw)
    assert pk2.serialize().tobytes() == raw
    with pytest.raises(ValueError):
        cpu.ProvingKey.parse(curve.curve_id, raw[:-3])
    r_, s_ = 0xabcdef0123456789abcdef % curve.r, 0x1122334455667788990011223344 % curve.r
    want = formats.proof_raw(curve, g16.prove(curve, cs, pk_py, z, r_, s_))
    for threads in (1, 4):
        got, tm = cpu.prove(c, pk2, c.assignment(), r_, s_, threads=threads)
        assert got == want
    assert cpu.trapdoor(c, cpu.toxic_bytes(tox), c.assignment(), r_, s_) == want
    assert cpu.witness_map(c, c.assignment()).tobytes() == le32(g16.witness_map(curve, cs, z)).tobytes()
    # r = 0: B1 skipped
    got0, _ = cpu.prove(c, pk, c.assignment(), 0, s_)
    assert got0 == formats.proof_raw(curve, g16.trapdoor_prove(curve, cs, tox, z, 0, s_))


@pytest.mark.slow
def test_medium_size_prove_equals_trapdoor():
    """2^12 constraints: algorithmic prover (FFT + Pippenger with real windows) == closed form."""
    curve = BN254
    c = cpu.Circuit.synth(0, (1 << 12) - 2, 0x5EED0003)
    assert c.N == 1 << 12
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    pk = cpu.ProvingKey.setup(c, tox)
    z = c.assignment()
    got, tm = cpu.prove(c, pk, z, 123456789, 987654321)
    assert got == cpu.trapdoor(c, tox, z, 123456789, 987654321)
