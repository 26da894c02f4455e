"""Differential test on random constraint systems (not the regular chain circuits of the other tests): random sparse rows
with empty combinations, variables that appear in no row of a matrix (their key entries are the point at infinity),
repeated and small witness values, several public inputs — both schemes, device (emulator on CPU, HIP on the GPU) against
the C++ oracle's closed form and algorithmic prover."""
import random

import numpy as np
import pytest

from oracle import cpu
from oracle import gm17
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native


def le(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8)


def random_system(curve, rnd, n, l, extra):
    """n rows: A_k, B_k random combinations over the variables allocated so far, C_k = one fresh variable (or, now and
    then, an empty combination made true by forcing A_k to evaluate to 0).  `extra` unused witness variables at the end."""
    r = curve.r
    z = [1] + [rnd.choice([0, 1, 2, rnd.randrange(r)]) for _ in range(l - 1)]
    A, B, C = [], [], []
    small = lambda: rnd.choice([1, r - 1, 2, rnd.randrange(1 << 64), rnd.randrange(r)])

    def lc(maxlen):
        k = rnd.randrange(0, maxlen + 1)
        cols = rnd.sample(range(len(z)), min(k, len(z)))
        return [(c, small()) for c in cols]

    ev = lambda row: sum(c * z[j] for j, c in row) % r
    for _ in range(n):
        a, b = lc(4), lc(3)
        if rnd.random() < 0.15:                         # 0 * b = 0 with an empty C row
            a = []
            A.append(a); B.append(b); C.append([])
            continue
        v = ev(a) * ev(b) % r
        z.append(v)
        A.append(a); B.append(b); C.append([(len(z) - 1, 1)])
    for _ in range(extra):
        z.append(rnd.choice([0, 1, 7, rnd.randrange(r)]))   # variables no constraint mentions
    cs = g16.R1CS(l=l, w=len(z) - l)
    cs.A, cs.B, cs.C = A, B, C
    assert cs.is_satisfied(z, r)
    return cs, z


def csr(rows):
    rp, col, val = [0], [], []
    for row in rows:
        for j, v in sorted(row):
            col.append(j); val.append(v)
        rp.append(len(col))
    return np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint32), le(val) if val else np.zeros(0, dtype=np.uint8)


def run(ctx, curve, seed, n, l, extra):
    rnd = random.Random(seed)
    cs, z = random_system(curve, rnd, n, l, extra)
    mats = [csr(cs.A), csr(cs.B), csr(cs.C)]
    zb = le(z)
    dcs = native.ConstraintSystem(ctx, curve.curve_id, cs.n, cs.l, cs.w, mats)
    oc = cpu.Circuit.from_csr(curve.curve_id, cs.n, cs.l, cs.w, mats)
    # Groth16
    tox = g16.Toxic.from_seed(curve, seed)
    tb = cpu.toxic_bytes(tox)
    raw = native.setup_g16(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))
    assert raw.tobytes() == cpu.ProvingKey.setup(oc, tb).serialize().tobytes()
    pk = native.ProvingKey(ctx, curve.curve_id, raw)
    r_, s_ = rnd.randrange(curve.r), rnd.randrange(curve.r)
    got = native.prove_g16(ctx, pk, dcs, zb, r_, s_)
    assert got == cpu.trapdoor(oc, tb, zb, r_, s_)
    assert got == cpu.prove(oc, cpu.ProvingKey.parse(curve.curve_id, raw), zb, r_, s_)[0]
    assert dcs.witness_map(zb).tobytes() == cpu.witness_map(oc, zb).tobytes()
    pk.bind(dcs)                                            # the key bound to the system (zkhip_pk_bind_r1cs): the same bytes
    assert native.prove_g16(ctx, pk, dcs, zb, r_, s_) == got
    assert native.prove_g16(ctx, pk, dcs, zb, 0, s_) == cpu.trapdoor(oc, tb, zb, 0, s_)
    pk.unbind()
    # GM17
    t17 = gm17.Toxic.from_seed(curve, seed)
    tb17 = cpu.gm17_toxic_bytes(t17)
    raw17 = native.setup_gm17(ctx, dcs, (t17.alpha, t17.beta, t17.gamma, t17.t))
    opk17 = cpu.Gm17ProvingKey.setup(oc, tb17)
    assert raw17.tobytes() == opk17.serialize().tobytes()
    pk17 = native.ProvingKey(ctx, curve.curve_id, raw17, scheme="gm17")
    d1, d2 = rnd.randrange(curve.r), rnd.randrange(curve.r)
    got17 = native.prove_gm17(ctx, pk17, dcs, zb, d1, d2, r_)
    assert got17 == cpu.gm17_trapdoor(oc, tb17, zb, d1, r_)
    assert got17 == cpu.gm17_prove(oc, opk17, zb, d1, d2, r_)[0]


CASES = [(BN254, 1, 9, 1, 0), (BN254, 2, 17, 3, 4), (BLS12_381, 3, 12, 2, 2), (BN254, 4, 40, 5, 7), (BLS12_381, 5, 1, 4, 1)]


@pytest.mark.parametrize("curve,seed,n,l,extra", CASES, ids=lambda v: getattr(v, "name", str(v)))
def test_random_systems_on_emulator(curve, seed, n, l, extra):
    from emu_util import emu_library
    ctx = native.Context(0, emu_library())
    run(ctx, curve, seed, n, l, extra)
    ctx.close()


@pytest.mark.gpu
def test_random_systems_on_gpu():
    ctx = native.Context(0)
    for curve, seed, n, l, extra in CASES + [(BN254, 6, 3000, 9, 100), (BLS12_381, 7, 1500, 4, 33), (BN254, 8, 20000, 2, 5)]:
        run(ctx, curve, seed, n, l, extra)
    ctx.close()


def _rows_of_every_length(curve, rnd, lengths, l=3):
    """A system whose k-th constraint has lengths[k] terms in A, lengths[-1-k] in B and lengths[(k + 1) % n] in C — each matrix
    meets every length, rows on either side of the mat-vec's 32-term boundary sit next to each other.  z is chosen first; A and B
    are random, C is random up to one coefficient solved for (its variable's value is invertible)."""
    r = curve.r
    m = max(lengths) + 8
    z = [1] + [rnd.randrange(1, r) for _ in range(m - 1)]
    ev = lambda row: sum(c * z[j] for j, c in row) % r
    A, B, C = [], [], []
    n = len(lengths)
    for k in range(n):
        def lc(length):
            cols = rnd.sample(range(m), length)
            return [(c, rnd.choice([1, r - 1, rnd.randrange(1 << 40), rnd.randrange(r)])) for c in cols]
        a, b, c = lc(lengths[k]), lc(lengths[n - 1 - k]), lc(lengths[(k + 1) % n])
        if c:
            j, _ = c[-1]
            rest = ev(c[:-1])
            c[-1] = (j, (ev(a) * ev(b) - rest) * pow(z[j], r - 2, r) % r)
        else:
            a = []
        A.append(a); B.append(b); C.append(c)
    cs = g16.R1CS(l=l, w=m - l)
    cs.A, cs.B, cs.C = A, B, C
    assert cs.is_satisfied(z, r)
    return cs, z


def _check_row_lengths(ctx, curve):
    lengths = [0, 1, 2, 31, 32, 33, 34, 63, 64, 65, 127, 128, 129, 300, 1000, 32, 33, 1, 0, 64, 500, 33, 511, 512, 513, 1100, 2]
    cs, z = _rows_of_every_length(curve, random.Random(77 + curve.curve_id), lengths)
    mats = [csr(cs.A), csr(cs.B), csr(cs.C)]
    zb = le(z)
    dcs = native.ConstraintSystem(ctx, curve.curve_id, len(lengths), cs.l, cs.w, mats)
    oc = cpu.Circuit.from_csr(curve.curve_id, len(lengths), cs.l, cs.w, mats)
    assert dcs.witness_map(zb).tobytes() == cpu.witness_map(oc, zb).tobytes()
    # a bound key over the same system: long rows in C (whose mat-vec a bound proof skips, k_matvec_long's share included) and
    # columns of C with hundreds of entries (the per-variable sums of zkhip_pk_bind_r1cs)
    tox = g16.Toxic.from_seed(curve, 5)
    tb = cpu.toxic_bytes(tox)
    pk = native.ProvingKey(ctx, curve.curve_id, native.setup_g16(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau)))
    want = cpu.trapdoor(oc, tb, zb, 12, 34)
    assert native.prove_g16(ctx, pk, dcs, zb, 12, 34) == want
    pk.bind(dcs)
    assert native.prove_g16(ctx, pk, dcs, zb, 12, 34) == want


@pytest.mark.parametrize("curve", [BN254, BLS12_381], ids=lambda c: c.name)
def test_rows_around_the_long_row_boundary_on_emulator(curve):
    """k_matvec leaves rows of more than 32 terms to k_matvec_long (a wavefront per row) and rows of more than 512 to k_matvec_huge (a
    workgroup per row): every length around both boundaries, in every matrix, next to empty and one-term rows."""
    from emu_util import emu_library
    ctx = native.Context(0, emu_library())
    try:
        _check_row_lengths(ctx, curve)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_rows_around_the_long_row_boundary_on_gpu():
    ctx = native.Context(0)
    try:
        for curve in (BN254, BLS12_381):
            _check_row_lengths(ctx, curve)
    finally:
        ctx.close()
