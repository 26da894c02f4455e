"""`-m gpu` parity tests: libzkhip.so (hand-written HIP, through the C ABI) vs the CPU oracle on the same
seeded inputs.  Bit-exact everywhere (integer/modular arithmetic only — no tolerance).  Sizes are chosen
so the oracle finishes in seconds; BASELINE.json's full sizes are covered by the size-independent
property test at the end (closed-form trapdoor proof + linearity of the MSM)."""
import os
import random

import numpy as np
import pytest

from oracle import cpu, formats
from oracle import groth16 as g16
from oracle.curves import groups
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

pytestmark = pytest.mark.gpu
CURVES = [BN254, BLS12_381]


def le(vals, nb=32):
    return np.frombuffer(b"".join(int(v).to_bytes(nb, "little") for v in vals), dtype=np.uint8)


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)              # raises if libzkhip.so or the GPU is missing: no fallback
    d = c.describe()
    assert "EMULATOR" not in d and "gfx950" in d, d
    yield c
    c.close()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_field_ops(ctx, curve):
    rnd = random.Random(11)
    for field, p, nb in ((0, curve.r, 32), (1, curve.q, curve.fq_bytes)):
        n = 5000
        a = [0, 1, p - 1, p - 1, 2, (1 << (8 * nb)) % p] + [rnd.randrange(p) for _ in range(n)]
        b = [0, p - 1, p - 1, 1, p - 2, (1 << (8 * nb)) % p] + [rnd.randrange(p) for _ in range(n)]
        for op, fn in (("add", lambda x, y: (x + y) % p), ("sub", lambda x, y: (x - y) % p), ("mul", lambda x, y: x * y % p)):
            got = ctx.field_op(curve.curve_id, field, op, le(a, nb), le(b, nb))
            assert got.tobytes() == le([fn(x, y) for x, y in zip(a, b)], nb).tobytes(), (field, op)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("logn", [0, 1, 4, 10, 11, 13, 16])
def test_ntt(ctx, curve, logn):
    rnd = np.random.default_rng(logn)
    a = rnd.integers(0, 256, size=(1 << logn) * 32, dtype=np.uint8)
    a.reshape(-1, 32)[:, 31] &= 0x0f        # < r
    for d in ("fft", "ifft", "coset_fft", "coset_ifft"):
        assert ctx.ntt(curve.curve_id, a, d).tobytes() == cpu.ntt(curve.curve_id, a, d).tobytes(), d
    # round trip
    assert ctx.ntt(curve.curve_id, ctx.ntt(curve.curve_id, a, "coset_fft"), "coset_ifft").tobytes() == a.tobytes()


def _bases(curve, n, seed):
    """n pseudo-random points = oracle fixed-base multiples (via its setup machinery: a_query of a circuit)."""
    oc = cpu.Circuit.synth(curve.curve_id, n, seed)
    raw = cpu.ProvingKey.setup(oc, cpu.toxic_bytes(g16.Toxic.from_seed(curve, seed))).serialize()
    pk = formats.ark_pk_deserialize(curve, raw.tobytes()) if n <= 64 else None
    nb = curve.fq_bytes
    off = 2 * nb + 3 * 4 * nb + 8 + oc.l * 2 * nb + 2 * 2 * nb + 8
    g1 = raw[off:off + oc.m * 2 * nb]
    off2 = off + oc.m * 2 * nb + 8 + oc.m * 2 * nb + 8
    g2 = raw[off2:off2 + oc.m * 4 * nb]
    return oc.m, g1, g2


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 33, 1000, 20000])
def test_msm(ctx, curve, n):
    m, g1, g2 = _bases(curve, max(n - 4, 1), 100 + n)
    n = m
    rnd = np.random.default_rng(n)
    ks = rnd.integers(0, 256, size=n * 32, dtype=np.uint8)
    ks.reshape(-1, 32)[:, 31] &= 0x0f
    k2 = ks.reshape(-1, 32)
    k2[0] = 0
    if n > 3:
        k2[1] = 0; k2[1, 0] = 1                     # scalar one
        k2[2] = le([curve.r - 1])                   # -1
    assert ctx.msm(curve.curve_id, 1, g1, ks) == cpu.msm(curve.curve_id, 1, g1, ks)
    if n <= 1100:
        assert ctx.msm(curve.curve_id, 2, g2, ks) == cpu.msm(curve.curve_id, 2, g2, ks)


def test_msm_skewed_scalars(ctx):
    """Hot buckets at a realistic size: 60 % ones, 10 % zeros, 10 % copies of one value, 5 % of -1 (the
    'sha-like' wire statistics); exercises the ones bucket, the heavy-bucket workgroup reduction and buckets
    straddling work-item boundaries for several slice lengths."""
    curve = BN254
    m, g1, g2 = _bases(curve, 30000, 4242)
    rnd = np.random.default_rng(17)
    ks = rnd.integers(0, 256, size=(m, 32), dtype=np.uint8)
    ks[:, 31] &= 0x0f
    cls = rnd.random(m)
    one = np.zeros(32, dtype=np.uint8); one[0] = 1
    ks[cls < 0.6] = one
    ks[(cls >= 0.6) & (cls < 0.7)] = 0
    ks[(cls >= 0.7) & (cls < 0.8)] = ks[-1]
    ks[(cls >= 0.8) & (cls < 0.85)] = le([curve.r - 1])
    ks = ks.reshape(-1)
    want1 = cpu.msm(0, 1, g1, ks)
    assert ctx.msm(0, 1, g1, ks) == want1
    assert ctx.msm(0, 2, g2[:4000 * 128], ks[:4000 * 32]) == cpu.msm(0, 2, g2[:4000 * 128], ks[:4000 * 32])
    try:
        for P, lanes in ((1, 0), (5, 0), (64, 0), (1000, 0), (1, 1000), (1, 77777)):
            ctx.tune("msm_min_slice", P)
            ctx.tune("msm_lanes", lanes)
            assert ctx.msm(0, 1, g1, ks) == want1, (P, lanes)
    finally:
        ctx.tune("msm_min_slice", 8)
        ctx.tune("msm_lanes", 0)


def test_msm_edge_points(ctx):
    curve = BN254
    G1, G2 = groups(curve)
    rnd = random.Random(3)
    n = 64
    p1 = [G1.amul(G1.gen, rnd.randrange(1, curve.r)) for _ in range(n)]
    ks = [rnd.randrange(curve.r) for _ in range(n)]
    p1[3] = None
    p1[5] = p1[6]; ks[5] = ks[6]
    p1[7] = G1.aneg(p1[8]); ks[7] = ks[8]
    b1 = np.frombuffer(b"".join(formats.ser_g1(curve, P) for P in p1), dtype=np.uint8)
    assert ctx.msm(0, 1, b1, le(ks)) == cpu.msm(0, 1, b1, le(ks))
    assert ctx.msm(0, 1, b1, np.zeros(n * 32, dtype=np.uint8))[-1] == 1
    for c in (2, 7, 11, 16):
        ctx.tune("msm_c", c)
        try:
            assert ctx.msm(0, 1, b1, le(ks)) == cpu.msm(0, 1, b1, le(ks)), c
        finally:
            ctx.tune("msm_c", 0)


@pytest.mark.parametrize("curve,logn,kind", [
    (BN254, 4, "dense"), (BN254, 10, "dense"), (BN254, 11, "sha"), (BN254, 14, "dense"), (BN254, 16, "dense"),
    (BLS12_381, 5, "dense"), (BLS12_381, 12, "sha"),
], ids=lambda v: getattr(v, "name", str(v)))
def test_prove_matches_oracle(ctx, curve, logn, kind):
    n = (1 << logn) - 2 if kind == "dense" else (1 << logn) - 2
    oc = cpu.Circuit.synth(curve.curve_id, n, 0x5EED0000 + logn, kind)
    assert oc.N == 1 << logn
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    opk = cpu.ProvingKey.setup(oc, tox)
    z = oc.assignment()
    r_, s_ = 0x1234567890abcdef % curve.r, 0xfedcba9876543210fedcba % curve.r
    want, _ = cpu.prove(oc, opk, z, r_, s_)                       # O2: algorithmic restatement of ark
    assert want == cpu.trapdoor(oc, tox, z, r_, s_)               # O1: closed form
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, curve.curve_id, opk.serialize())
    assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
    got, tm = native.prove_g16(ctx, pk, cs, z, r_, s_, want_timings=True)
    assert got == want
    assert native.prove_g16(ctx, pk, cs, z, 0, 0) == cpu.trapdoor(oc, tox, z, 0, 0)
    za = native.Assignment(ctx, cs, z)
    assert native.prove_g16_resident(ctx, pk, cs, za, r_, s_) == want
    assert native.prove_g16_resident(ctx, pk, cs, za, 3, 4) == cpu.trapdoor(oc, tox, z, 3, 4)
    # pipelined batch (two proofs in flight): distinct witnesses and blinding factors, each equal to its single proof
    rs = [(r_, s_), (3, 4), (0, 9), (r_ ^ 5, s_ ^ 7), (1, 1)]
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za] * len(rs), rs)
    assert proofs[0] == want
    for (a, b), pr in zip(rs, proofs):
        assert pr == cpu.trapdoor(oc, tox, z, a, b)
    proofs, _ = native.prove_g16_batch(ctx, pk, cs, np.concatenate([z, z, z]), rs[:3])
    assert proofs == [cpu.trapdoor(oc, tox, z, a, b) for a, b in rs[:3]]
    # determinism: same (pk, z, r, s) -> same bytes (cf. zokrates_js/tests/tests.js:248-267)
    assert native.prove_g16(ctx, pk, cs, z, r_, s_) == got
    # the key bound to this system (zkhip_pk_bind_r1cs: four transforms per proof, c folded into the bases): the same bytes from
    # every entry point, and the key as loaded again after unbind
    pk.bind(cs)
    assert pk.is_bound(cs)
    assert native.prove_g16(ctx, pk, cs, z, r_, s_) == want
    assert native.prove_g16(ctx, pk, cs, z, 0, 0) == cpu.trapdoor(oc, tox, z, 0, 0)
    assert native.prove_g16_resident(ctx, pk, cs, za, 3, 4) == cpu.trapdoor(oc, tox, z, 3, 4)
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za] * len(rs), rs)
    assert proofs == [cpu.trapdoor(oc, tox, z, a, b) for a, b in rs]
    zbad = np.array(z, copy=True)
    zbad[32 * (oc.l + 1)] ^= 1                                    # an assignment that does not satisfy the system: what ark computes for it
    if logn <= 14:
        assert native.prove_g16(ctx, pk, cs, zbad, 5, 6) == cpu.prove(oc, opk, zbad, 5, 6)[0]
    pk.unbind()
    assert not pk.is_bound(cs)
    assert native.prove_g16(ctx, pk, cs, z, r_, s_) == want


def test_schedule_does_not_change_proofs(ctx):
    """Gate / fused-launch / slices-per-launch settings (zkhip_ctx_tune) on the device: same proof bytes, single and
    pipelined, Groth16 and GM17; dense and boolean-heavy witnesses (heavy buckets inside the fused row fold)."""
    from schedule_checks import schedule_invariance
    schedule_invariance(ctx, logn=12)


def test_stream_plan_does_not_change_proofs():
    """The resident prover's stream plan (ZKHIP_TUNE_PIPE_PLAN) on the device: a context of its own (the plan is fixed at the first proof)."""
    from schedule_checks import stream_plan_invariance
    stream_plan_invariance(lambda: native.Context(0), logn=12)


def test_pairing_accepts_device_proof(ctx):
    """O3: the verification equation of zokrates_proof_systems/src/scheme/groth16.rs:156-172 accepts the
    device proof and rejects a mutated one (to_token.rs:68-71)."""
    from oracle import pairing
    curve = BN254
    cs_py, z_py = g16.synthetic_chain(curve, 30, 0x5EED0042)
    tox = g16.Toxic.from_seed(curve)
    pk_py, vk = g16.setup(curve, cs_py, tox)
    oc = cpu.Circuit.synth(0, 30, 0x5EED0042)
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, 0, formats.ark_pk_serialize(curve, pk_py))
    raw = native.prove_g16(ctx, pk, cs, oc.assignment(), 77, 99)
    proof = formats.proof_from_raw(curve, raw)
    assert pairing.groth16_verify(curve, vk, proof, z_py[1:cs_py.l])
    G1, _ = groups(curve)
    bad = (G1.aadd(proof[0], G1.gen), proof[1], proof[2])
    assert not pairing.groth16_verify(curve, vk, bad, z_py[1:cs_py.l])


def test_error_paths(ctx):
    oc = cpu.Circuit.synth(0, 5, 1)
    raw = cpu.ProvingKey.setup(oc, cpu.toxic_bytes(g16.Toxic.from_seed(BN254))).serialize()
    with pytest.raises(native.ZkhipError) as e:
        native.ProvingKey(ctx, 0, raw[:-1])
    assert e.value.code == -2
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, 0, raw)
    z = oc.assignment().copy(); z[0] = 2
    with pytest.raises(native.ZkhipError) as e:
        native.prove_g16(ctx, pk, cs, z, 1, 2)
    assert e.value.code == -1


@pytest.mark.parametrize("curve,logn,kind", [(BN254, 6, "dense"), (BN254, 12, "sha"), (BLS12_381, 9, "dense")],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_setup_matches_oracle(ctx, curve, logn, kind):
    """N3: zkhip_setup_g16 writes the same `proving.key` bytes as the oracle's restatement of
    ark_groth16::generate_random_parameters (App. A.6) for the same toxic waste and generators; the key then
    round-trips through zkhip_pk_load_g16 and proves."""
    from zokrates_amd import synth
    circ = synth.circuit(curve.curve_id, logn, kind=kind, seed=0x77 + logn)
    cs = native.ConstraintSystem(ctx, curve.curve_id, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(curve.curve_id, 0xBEEF)
    raw = native.setup_g16(ctx, cs, tox)
    oc = cpu.Circuit.from_csr(curve.curve_id, circ.n, circ.l, circ.w, circ.mats())
    tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
    assert raw.tobytes() == cpu.ProvingKey.setup(oc, tb).serialize().tobytes()
    z = circ.assignment(99)
    pk = native.ProvingKey(ctx, curve.curve_id, raw)
    assert native.prove_g16(ctx, pk, cs, z, 17, 19) == cpu.trapdoor(oc, tb, z, 17, 19)
    # non-standard generators (ark samples random ones): 5*G1, 7*G2 from the oracle's MSM
    g1, g2 = cpu.generators(curve.curve_id)
    g1b = cpu.msm(curve.curve_id, 1, np.frombuffer(g1, dtype=np.uint8), le([5]))[:-1]
    g2b = cpu.msm(curve.curve_id, 2, np.frombuffer(g2, dtype=np.uint8), le([7]))[:-1]
    raw2 = native.setup_g16(ctx, cs, tox, np.frombuffer(g1b, dtype=np.uint8), np.frombuffer(g2b, dtype=np.uint8))
    assert raw2.tobytes() != raw.tobytes()
    pk2 = native.ProvingKey(ctx, curve.curve_id, raw2)
    opk2 = cpu.ProvingKey.parse(curve.curve_id, raw2.tobytes())
    want, _ = cpu.prove(oc, opk2, z, 17, 19)
    assert native.prove_g16(ctx, pk2, cs, z, 17, 19) == want


def test_full_size_properties(ctx):
    """BASELINE.json configs[1] (2^20 constraints, BN254) is too large for the Python oracle; check it through
    size-independent properties: (1) the device proof for a key with known toxic waste equals the closed-form
    trapdoor proof computed by the C++ oracle with pure Fr arithmetic + 3 scalar multiplications, (2) MSM
    linearity: MSM(P, k) + MSM(P, k') = MSM(P, k + k') over 2^20 points, (3) NTT round trip on 2^20 points."""
    from zokrates_amd import synth
    lg = int(os.environ.get("ZKHIP_TEST_FULL_LOG", "20"))
    circ = synth.circuit(0, lg)
    cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(0)
    raw = native.setup_g16(ctx, cs, tox)
    pk = native.ProvingKey(ctx, 0, raw)
    z = circ.assignment(0x5EED0001)
    oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
    tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
    r_, s_ = 0xDEADBEEF12345678, 0xCAFEBABE87654321
    want = cpu.trapdoor(oc, tb, z, r_, s_)
    assert native.prove_g16(ctx, pk, cs, z, r_, s_) == want
    # (1a) and the ALGORITHMIC oracle at this size (O2: the C++ restatement of ark's create_proof over the key bytes the device made —
    # seven transforms and five Pippenger MSMs on all host threads, seconds at 2^20): the suite itself holds the device to it, not
    # only bench.py's cpu_baseline leg
    if lg <= int(os.environ.get("ZKHIP_TEST_FULL_O2_MAX_LOG", "20")):
        want2, _ = cpu.prove(oc, cpu.ProvingKey.parse(0, raw.tobytes()), z, r_, s_)
        assert want2 == want
    # (1b) the same with the key bound to the system (two 2^20-point transforms over G1 at bind time, H' and L' in the MSMs)
    pk.bind(cs)
    assert native.prove_g16(ctx, pk, cs, z, r_, s_) == want
    z2 = circ.assignment(0x5EED0002)
    assert native.prove_g16(ctx, pk, cs, z2, 7, 0) == cpu.trapdoor(oc, tb, z2, 7, 0)
    pk.unbind()
    # (2) linearity on the key's own a_query bases
    nb = 32
    off = 2 * nb + 3 * 4 * nb + 8 + circ.l * 2 * nb + 2 * 2 * nb + 8
    g1 = raw[off:off + circ.m * 2 * nb]
    rnd = np.random.default_rng(5)
    k1 = rnd.integers(0, 256, size=circ.m * 32, dtype=np.uint8); k1.reshape(-1, 32)[:, 31] &= 0x07
    k2 = rnd.integers(0, 256, size=circ.m * 32, dtype=np.uint8); k2.reshape(-1, 32)[:, 31] &= 0x07
    a1 = k1.view(np.uint64).reshape(-1, 4); a2 = k2.view(np.uint64).reshape(-1, 4)
    ks = np.zeros_like(a1)
    carry = np.zeros(circ.m, dtype=np.uint64)
    for i in range(4):                      # 252-bit + 252-bit < r: plain multi-limb addition
        t = a1[:, i] + a2[:, i]
        c1 = t < a1[:, i]
        t2 = t + carry
        c2 = t2 < t
        ks[:, i] = t2
        carry = (c1 | c2).astype(np.uint64)
    p1, p2, p12 = (ctx.msm(0, 1, g1, k) for k in (k1, k2, ks.view(np.uint8).reshape(-1)))
    two = np.concatenate([np.frombuffer(p1[:-1], dtype=np.uint8), np.frombuffer(p2[:-1], dtype=np.uint8)])
    assert ctx.msm(0, 1, two, le([1, 1])) == p12
    # (3) NTT round trip
    a = z[: (1 << lg) * 32]
    assert ctx.ntt(0, ctx.ntt(0, a, "coset_fft"), "coset_ifft").tobytes() == a.tobytes()
    assert ctx.ntt(0, ctx.ntt(0, a, "fft"), "ifft").tobytes() == a.tobytes()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_proof_virtual_ranks(ctx, world):
    """one 2^14 proof split over `world` ranks (virtual ranks: one after the other on this GPU) is
    bit-identical to the unsharded proof and to the oracle."""
    from zokrates_amd import synth
    circ = synth.circuit(0, 14, seed=0x5A4D)
    cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(0, 0xFACE)
    raw = native.setup_g16(ctx, cs, tox)
    z = circ.assignment(77)
    r_, s_ = 0x1234567890abcdef1234, 0xfedcba0987654321
    whole = native.ProvingKey(ctx, 0, raw)
    want = native.prove_g16(ctx, whole, cs, z, r_, s_)
    oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
    assert want == cpu.trapdoor(oc, b"".join(int(v).to_bytes(32, "little") for v in tox), z, r_, s_)
    shards = [native.ProvingKey(ctx, 0, raw, rank=k, world=world) for k in range(world)]
    za = native.Assignment(ctx, cs, z)
    parts = [native.prove_g16_partial(ctx, shards[k], cs, za, r_, s_) for k in range(world)]
    assert native.combine_g16(ctx, shards[0], parts, r_, s_) == want


@pytest.mark.parametrize("setting", ["0", "1", "2"])
def test_lone_proofs_hold_their_g1_lanes_for_the_g2_accumulation(setting):
    """ZKHIP_G2_HEAD_START (0 never, 1 over a bound key — the default —, 2 always): on BLS12-381, whose G2 accumulation runs one wave
    per SIMD, a lone proof's G1 lanes also wait for the end of that accumulation.  A scheduling rule on real streams: the same proof
    bytes at every setting, bound and as loaded, single and pipelined."""
    os.environ["ZKHIP_G2_HEAD_START"] = setting
    try:
        c2 = native.Context(0)
    finally:
        os.environ.pop("ZKHIP_G2_HEAD_START")
    try:
        for curve, logn in ((BLS12_381, 12), (BN254, 10)):
            oc = cpu.Circuit.synth(curve.curve_id, (1 << logn) - 2, 0x5EED0090 + logn, "sha")
            tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            raw = cpu.ProvingKey.setup(oc, tox).serialize()
            cs = native.ConstraintSystem(c2, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
            z = oc.assignment()
            want = cpu.trapdoor(oc, tox, z, 41, 42)
            pk = native.ProvingKey(c2, curve.curve_id, raw)
            za = native.Assignment(c2, cs, z)
            for bound in (False, True):
                if bound:
                    pk.bind(cs)
                for _ in range(3):
                    assert native.prove_g16(c2, pk, cs, z, 41, 42) == want and native.prove_g16_resident(c2, pk, cs, za, 41, 42) == want
                proofs, _ = native.prove_g16_resident_batch(c2, pk, cs, [za] * 5, [(41, 42)] * 5)
                assert proofs == [want] * 5
    finally:
        c2.close()
