"""Kernel-logic tests on the TEST-ONLY CPU emulator (tests/_emu): same kernel sources as libzkhip.so,
checked bit-for-bit against the oracle on tiny inputs.  These do not replace the `-m gpu` parity tests."""
import os
import random

import numpy as np
import pytest

from oracle import cpu, formats
from oracle import groth16 as g16
from oracle.curves import groups
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

from emu_util import emu_library
from schedule_checks import schedule_invariance, stream_plan_invariance

CURVES = [BN254, BLS12_381]


def le(vals, nb=32):
    return np.frombuffer(b"".join(int(v).to_bytes(nb, "little") for v in vals), dtype=np.uint8)


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0, emu_library())
    assert "EMULATOR" in c.describe()
    yield c
    c.close()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_field_ops(ctx, curve):
    rnd = random.Random(5)
    for field, p, nb in ((0, curve.r, 32), (1, curve.q, curve.fq_bytes)):
        a = [0, 1, p - 1, p - 1, 2, (1 << (8 * nb)) % p] + [rnd.randrange(p) for _ in range(70)]
        b = [0, p - 1, p - 1, 1, p - 2, (1 << (8 * nb)) % p] + [rnd.randrange(p) for _ in range(70)]
        for op, fn in (("add", lambda x, y: (x + y) % p), ("sub", lambda x, y: (x - y) % p), ("mul", lambda x, y: x * y % p)):
            got = ctx.field_op(curve.curve_id, field, op, le(a, nb), le(b, nb))
            assert got.tobytes() == le([fn(x, y) for x, y in zip(a, b)], nb).tobytes(), (field, op)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("single_max", [10, 1])
def test_ntt(ctx, curve, single_max):
    """single_max = 1 forces the two-pass (cols + rows, sigma-order) path on small domains."""
    c2 = native.Context(0, emu_library())   # fresh plan cache
    c2.tune("ntt_single_max_log", single_max)
    rnd = random.Random(6)
    try:
        for logn in ((0, 1, 2, 5, 7, 10) if single_max == 10 else (2, 3, 5, 6, 7, 9)):     # even / odd sub-lengths, N1 != N2
            a = le([rnd.randrange(curve.r) for _ in range(1 << logn)])
            for d in ("fft", "ifft", "coset_fft", "coset_ifft"):
                want = cpu.ntt(curve.curve_id, a, d).tobytes()
                for fuse in (1, 0):      # the first round of a pass on the way in (sub-transforms of 16 points and more) / through LDS
                    c2.tune("ntt_fuse_first", fuse)
                    assert c2.ntt(curve.curve_id, a, d).tobytes() == want, (logn, d, fuse)
    finally:
        c2.close()


def _rand_points(curve, n, rnd, with_edge=True):
    G1, G2 = groups(curve)
    p1 = [G1.amul(G1.gen, rnd.randrange(1, curve.r)) for _ in range(n)]
    p2 = [G2.amul(G2.gen, rnd.randrange(1, curve.r)) for _ in range(n)]
    ks = [rnd.randrange(curve.r) for _ in range(n)]
    if with_edge and n >= 8:
        ks[0] = 0; ks[1] = 1; ks[2] = curve.r - 1
        p1[3] = None; p2[4] = None
        p1[5] = p1[6]; ks[5] = ks[6]                      # equal points in one bucket -> doubling branch
        p2[5] = p2[6]
        p1[7] = G1.aneg(p1[6]); ks[7] = ks[6]             # P + (-P) in one bucket -> infinity branch
    b1 = np.frombuffer(b"".join(formats.ser_g1(curve, P) for P in p1), dtype=np.uint8)
    b2 = np.frombuffer(b"".join(formats.ser_g2(curve, P) for P in p2), dtype=np.uint8)
    return b1, b2, le(ks)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm(ctx, curve):
    rnd = random.Random(7)
    for n in (1, 3, 20, 70):
        b1, b2, ks = _rand_points(curve, n, rnd)
        assert ctx.msm(curve.curve_id, 1, b1, ks) == cpu.msm(curve.curve_id, 1, b1, ks), n
        if n <= 20:
            assert ctx.msm(curve.curve_id, 2, b2, ks) == cpu.msm(curve.curve_id, 2, b2, ks), n
    # empty and all-zero
    assert ctx.msm(curve.curve_id, 1, b1[:0], ks[:0])[-1] == 1
    assert ctx.msm(curve.curve_id, 1, b1, np.zeros_like(ks))[-1] == 1


def test_msm_window_sizes(ctx):
    """Same answer for every window width (exercises digit recoding / carries / fold geometry)."""
    rnd = random.Random(8)
    b1, _, ks = _rand_points(BN254, 40, rnd)
    want = cpu.msm(0, 1, b1, ks)
    try:
        for c in (2, 3, 5, 8, 13, 16, 17):      # (17: more buckets than one sort workgroup's histogram holds: two halves)
            ctx.tune("msm_c", c)
            assert ctx.msm(0, 1, b1, ks) == want, c
    finally:
        ctx.tune("msm_c", 0)


def test_msm_fold_fallback(ctx):
    """The double-and-add form of the last fold step (used when Lw + H points do not fit in LDS) gives the same sums
    as the scan form that normally runs."""
    rnd = random.Random(10)
    b1, b2, ks = _rand_points(BN254, 60, rnd)
    want1, want2 = cpu.msm(0, 1, b1, ks), cpu.msm(0, 2, b2[:20 * 128], ks[:20 * 32])
    try:
        for c in (3, 9, 12):
            ctx.tune("msm_c", c)
            for scan in (1, 0):
                ctx.tune("fold_scan", scan)
                assert ctx.msm(0, 1, b1, ks) == want1, (c, scan)
                assert ctx.msm(0, 2, b2[:20 * 128], ks[:20 * 32]) == want2, (c, scan)
    finally:
        ctx.tune("msm_c", 0)
        ctx.tune("fold_scan", 1)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_fold_meets_equal_and_opposite_sums(ctx, curve):
    """The fold adds bucket SUMS, and two of them can be the same point or opposite ones: the same base under two digits (buckets of one
    row and of one column: the line sums and the scan both pair them), a bucket whose slices hold equal partial sums, sums that cancel.
    xyzz_add_from's doubling takes the addition's own tail with other inputs (ec.cuh); both fold layouts must agree with the oracle."""
    rnd = random.Random(12)
    G1, G2 = groups(curve)
    P1, Q1 = (G1.amul(G1.gen, rnd.randrange(1, curve.r)) for _ in range(2))
    P2, Q2 = (G2.amul(G2.gen, rnd.randrange(1, curve.r)) for _ in range(2))
    try:
        for c, ks in ((3, [1, 3, 2, 4]),            # K = 4: buckets 0 and 2 hold P (the tree's first level pairs them), 1 and 3 hold Q and -Q
                      (3, [1, 2, 3, 4]),
                      (5, [1, 1 + 4, 9, 9 + 4]),     # K = 16: a 4 x 4 ... (Lw = K: one row) neighbours at every tree distance
                      (10, [1, 1 + 256, 2, 2 + 256]),   # K = 512, Lw = 256, H = 2: P in one COLUMN (rows 0 and 1), Q / -Q in the next
                      (4, [5, 5, 5, 5, 5, 5, 5, 5])):   # one bucket, eight entries P Q P Q ...: slices of two hold equal partial sums
            n = len(ks)
            if n == 4:
                p1, p2 = [P1, P1, Q1, G1.aneg(Q1)], [P2, P2, Q2, G2.aneg(Q2)]
            else:
                p1, p2 = [P1, Q1] * 4, [P2, Q2] * 4
            b1 = np.frombuffer(b"".join(formats.ser_g1(curve, P) for P in p1), dtype=np.uint8)
            b2 = np.frombuffer(b"".join(formats.ser_g2(curve, P) for P in p2), dtype=np.uint8)
            want1, want2 = cpu.msm(curve.curve_id, 1, b1, le(ks)), cpu.msm(curve.curve_id, 2, b2, le(ks))
            ctx.tune("msm_c", c)
            for lines in (1, 0):
                ctx.tune("fold_lines", lines)
                for min_slice in (1, 2):
                    ctx.tune("msm_min_slice", min_slice)
                    assert ctx.msm(curve.curve_id, 1, b1, le(ks)) == want1, (c, ks, lines, min_slice)
                    assert ctx.msm(curve.curve_id, 2, b2, le(ks)) == want2, (c, ks, lines, min_slice)
    finally:
        ctx.tune("msm_c", 0)
        ctx.tune("fold_lines", 0)
        ctx.tune("msm_min_slice", 8)


def test_msm_skewed_scalars(ctx):
    """Hot buckets: many scalars equal to 1, many copies of one full-width value and of -1 (a bucket spread over more
    than MSM_HEAVY slices -> workgroup reduction), zeros; several cuts of the sorted list (number of slices, finest
    slice) so that buckets straddle slice boundaries in every way."""
    curve = BN254
    rnd = random.Random(9)
    n = 150
    G1, G2 = groups(curve)
    p1 = [G1.amul(G1.gen, rnd.randrange(1, curve.r)) for _ in range(n)]
    p2 = [G2.amul(G2.gen, rnd.randrange(1, curve.r)) for _ in range(24)]
    hot = rnd.randrange(curve.r)
    ks = [1] * 50 + [hot] * 40 + [curve.r - 1] * 30 + [0] * 10 + [rnd.randrange(curve.r) for _ in range(20)]
    rnd.shuffle(ks)
    b1 = np.frombuffer(b"".join(formats.ser_g1(curve, P) for P in p1), dtype=np.uint8)
    b2 = np.frombuffer(b"".join(formats.ser_g2(curve, P) for P in p2), dtype=np.uint8)
    want1 = cpu.msm(0, 1, b1, le(ks))
    want2 = cpu.msm(0, 2, b2, le(ks[:24]))
    try:
        for P, lanes, c in ((1, 0, 4), (2, 0, 5), (3, 0, 3), (7, 0, 6), (32, 0, 4), (1, 11, 5), (1, 3, 7), (1, 1, 4)):
            ctx.tune("msm_min_slice", P)
            ctx.tune("msm_lanes", lanes)
            ctx.tune("msm_c", c)
            assert ctx.msm(0, 1, b1, le(ks)) == want1, (P, lanes, c)
            if P in (2, 32) or lanes == 3:
                assert ctx.msm(0, 2, b2, le(ks[:24])) == want2, (P, lanes, c)
        # a bucket over hundreds of slices (the ones of a witness of bits): k_msm_heavy_reduce sums its partials run by run
        # before the fold (runs of 2, 3 and 5 slices here), or the row's workgroup sums them alone — the same point either way
        ks2 = [1] * 140 + [hot] * 6 + [0] * 4
        rnd.shuffle(ks2)
        p1b = (p1 * 3)[:150 * 2 + 20]
        ks3 = ([1] * 300 + ks2)[:len(p1b)]
        b1b = np.frombuffer(b"".join(formats.ser_g1(curve, P) for P in p1b), dtype=np.uint8)
        for bases, scalars in ((b1, ks2), (b1b, ks3)):
            want = cpu.msm(0, 1, bases, le(scalars))
            for runs in (1, 0):
                ctx.tune("heavy_runs", runs)
                for P, lanes, c in ((1, 0, 4), (2, 0, 6)):
                    ctx.tune("msm_min_slice", P)
                    ctx.tune("msm_lanes", lanes)
                    ctx.tune("msm_c", c)
                    assert ctx.msm(0, 1, bases, le(scalars)) == want, (runs, P, lanes, c)
        ctx.tune("heavy_runs", 1)
        ctx.tune("msm_min_slice", 1)
        ctx.tune("msm_c", 4)
        k24 = ([1] * 22 + [hot, 0])
        assert ctx.msm(0, 2, b2, le(k24)) == cpu.msm(0, 2, b2, le(k24))
    finally:
        ctx.tune("heavy_runs", 1)
        ctx.tune("msm_min_slice", 8)
        ctx.tune("msm_lanes", 0)
        ctx.tune("msm_c", 0)


@pytest.mark.parametrize("curve,scheme", [(BN254, "g16"), (BN254, "gm17"), (BLS12_381, "g16")], ids=lambda v: getattr(v, "name", v))
def test_prove_with_the_ones_bucket_over_hundreds_of_slices(ctx, curve, scheme):
    """A witness of bits at the finest slicing: the bucket of the ones is spread over hundreds of slices in every table of the
    fused launch (window-multiple tables, A / B1 / L on one sorted list, G2 on the thinned one), so k_msm_heavy_reduce leaves run
    sums in three tables at once; the proof must be the oracle's with and without it."""
    from oracle import gm17
    oc = cpu.Circuit.synth(curve.curve_id, 400, 0x5EED0011, "sha")
    z = oc.assignment()
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    if scheme == "g16":
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        pk = native.ProvingKey(ctx, curve.curve_id, cpu.ProvingKey.setup(oc, tox).serialize())
        want = cpu.trapdoor(oc, tox, z, 5, 6)
        prove = lambda: native.prove_g16(ctx, pk, cs, z, 5, 6)
    else:
        tox = cpu.gm17_toxic_bytes(gm17.Toxic.from_seed(curve))
        pk = native.ProvingKey(ctx, curve.curve_id, cpu.Gm17ProvingKey.setup(oc, tox).serialize(), scheme="gm17")
        want = cpu.gm17_trapdoor(oc, tox, z, 5, 7)
        prove = lambda: native.prove_gm17(ctx, pk, cs, z, 5, 6, 7)
    try:
        ctx.tune("msm_min_slice", 1)
        for runs in (1, 0):
            ctx.tune("heavy_runs", runs)
            assert prove() == want, runs
    finally:
        ctx.tune("heavy_runs", 1)
        ctx.tune("msm_min_slice", 8)
        pk.close()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("kind", ["dense", "sha"])
def test_prove_matches_oracle(ctx, curve, kind):
    n = 13 if kind == "dense" else 29
    oc = cpu.Circuit.synth(curve.curve_id, n, 0x5EED0007, kind)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    opk = cpu.ProvingKey.setup(oc, tox)
    z = oc.assignment()
    r_, s_ = 0x1234567 % curve.r, 0x89abcdef0123 % curve.r
    want, _ = cpu.prove(oc, opk, z, r_, s_)
    assert want == cpu.trapdoor(oc, tox, z, r_, s_)
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, curve.curve_id, opk.serialize())
    assert (pk.m, pk.hlen, pk.w, pk.l) == (oc.m, oc.N - 1, oc.w, oc.l)
    assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
    got, tm = native.prove_g16(ctx, pk, cs, z, r_, s_, want_timings=True)
    assert got == want
    # r = 0 / s = 0 corner (ark skips B1 when r == 0; the result must not change)
    assert native.prove_g16(ctx, pk, cs, z, 0, s_) == cpu.trapdoor(oc, tox, z, 0, s_)
    assert native.prove_g16(ctx, pk, cs, z, r_, 0) == cpu.trapdoor(oc, tox, z, r_, 0)
    # resident assignment == host assignment, reusable across (r, s)
    za = native.Assignment(ctx, cs, z)
    assert native.prove_g16_resident(ctx, pk, cs, za, r_, s_) == want
    assert native.prove_g16_resident(ctx, pk, cs, za, 5, 6) == cpu.trapdoor(oc, tox, z, 5, 6)
    za.close()
    # batch (two proofs in flight) == singles; resident batch with a repeated assignment
    proofs, _ = native.prove_g16_batch(ctx, pk, cs, np.concatenate([z, z, z]), [(r_, s_), (5, 6), (7, 8)])
    assert proofs[0] == want and proofs[1] == cpu.trapdoor(oc, tox, z, 5, 6) and proofs[2] == cpu.trapdoor(oc, tox, z, 7, 8)
    za = native.Assignment(ctx, cs, z)
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za] * 4, [(r_, s_), (5, 6), (7, 8), (0, 0)])
    assert proofs[0] == want and proofs[1] == cpu.trapdoor(oc, tox, z, 5, 6) and proofs[3] == cpu.trapdoor(oc, tox, z, 0, 0)
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za], [(r_, s_)])
    assert proofs == [want]


@pytest.mark.parametrize("curve,world", [(BN254, 2), (BN254, 5), (BLS12_381, 3)], ids=lambda v: getattr(v, "name", str(v)))
def test_sharded_proof_virtual_ranks(ctx, curve, world):
    """One proof split over `world` ranks (here: one after the other on the same device): every rank loads its share of
    the key, computes partial sums, the records are combined — bit-identical to the unsharded proof."""
    oc = cpu.Circuit.synth(curve.curve_id, 21, 0x5EED0011)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    raw = cpu.ProvingKey.setup(oc, tox).serialize()
    z = oc.assignment()
    r_, s_ = 0xabcdef12345 % curve.r, 0x13579bdf2468 % curve.r
    want = cpu.trapdoor(oc, tox, z, r_, s_)
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    shards = [native.ProvingKey(ctx, curve.curve_id, raw, rank=k, world=world) for k in range(world)]
    parts = [native.prove_g16_partial(ctx, shards[k], cs, z, r_, s_) for k in range(world)]
    assert native.combine_g16(ctx, shards[0], parts, r_, s_) == want
    assert native.combine_g16(ctx, shards[-1], parts[::-1], r_, s_) == want          # order and combining rank do not matter
    za = native.Assignment(ctx, cs, z)
    parts = [native.prove_g16_partial(ctx, shards[k], cs, za, 0, s_) for k in range(world)]
    assert native.combine_g16(ctx, shards[0], parts, 0, s_) == cpu.trapdoor(oc, tox, z, 0, s_)
    with pytest.raises(native.ZkhipError):                                           # a shard cannot prove alone
        native.prove_g16(ctx, shards[0], cs, z, r_, s_)
    whole = native.ProvingKey(ctx, curve.curve_id, raw)
    assert native.combine_g16(ctx, whole, [native.prove_g16_partial(ctx, whole, cs, z, r_, s_)], r_, s_) == want


@pytest.mark.parametrize("curve", [BN254], ids=lambda c: c.name)      # (the sets / levels pairing is curve-independent host code)
def test_prove_with_thinned_tables(curve):
    """Keys too large for every window multiple of every base (domains above 2^24: the tables of a 2^26 key would take 384 GiB)
    keep every 2nd / 4th / ... multiple and fold as many bucket sets (ZKHIP_TUNE_MSM_SETS; automatic by device memory otherwise);
    sets >= windows is the table-less limit.  Same proofs — Groth16, GM17, a sharded key, a key image — at every setting."""
    c2 = native.Context(0, emu_library())
    try:
        oc = cpu.Circuit.synth(curve.curve_id, 40, 7)
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        tb17 = tox[:96] + tox[128:160]
        raw17 = cpu.Gm17ProvingKey.setup(oc, tb17).serialize()
        cs = native.ConstraintSystem(c2, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
        z = oc.assignment()
        want, want17 = cpu.trapdoor(oc, tox, z, 11, 13), cpu.gm17_trapdoor(oc, tb17, z, 21, 23)
        for sets, c in ((2, 0), (3, 0), (4, 5), (64, 0)):
            c2.tune("msm_sets", sets)
            c2.tune("msm_c", c)
            pk = native.ProvingKey(c2, curve.curve_id, raw)
            assert native.prove_g16(c2, pk, cs, z, 11, 13) == want, sets
            assert native.prove_g16(c2, native.ProvingKey.from_image(c2, curve.curve_id, pk.export_image()), cs, z, 11, 13) == want, sets
            pk17 = native.ProvingKey(c2, curve.curve_id, raw17, scheme="gm17")
            assert native.prove_gm17(c2, pk17, cs, z, 21, 7, 23) == want17, sets
            shards = [native.ProvingKey(c2, curve.curve_id, raw, rank=k, world=3) for k in range(3)]
            parts = [native.prove_g16_partial(c2, sh, cs, z, 11, 13) for sh in shards]
            assert native.combine_g16(c2, shards[0], parts, 11, 13) == want, sets
        # an image written with thinned tables names them in its header: imported under other defaults it keeps its own shape
        c2.tune("msm_sets", 2)
        img = native.ProvingKey(c2, curve.curve_id, raw).export_image()
        c2.tune("msm_sets", 0)
        assert native.prove_g16(c2, native.ProvingKey.from_image(c2, curve.curve_id, img), cs, z, 11, 13) == want
    finally:
        c2.close()


def test_prove_with_17_bit_windows():
    """254-bit scalars take 15 windows of 17 bits instead of 16 of 16 (what a key of 2^16 points or more gets by itself): 2^16
    buckets per MSM, which the sort's workgroups take in two halves (their LDS histogram holds 2^15 counters), a fold over 256 rows
    of 256 buckets, tables of 15 levels.  Forced here on a toy key; same proof, also through a key image."""
    c2 = native.Context(0, emu_library())
    try:
        oc = cpu.Circuit.synth(0, 40, 7)
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(BN254))
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        cs = native.ConstraintSystem(c2, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
        z = oc.assignment()
        want = cpu.trapdoor(oc, tox, z, 11, 13)
        c2.tune("msm_c", 17)
        pk = native.ProvingKey(c2, 0, raw)
        assert native.prove_g16(c2, pk, cs, z, 11, 13) == want
        img = pk.export_image()
        c2.tune("msm_c", 0)
        assert native.prove_g16(c2, native.ProvingKey.from_image(c2, 0, img), cs, z, 11, 13) == want
    finally:
        c2.close()


@pytest.mark.parametrize("mode", [1, 2], ids=["lanes-sit-out", "wavefront-vote"])
def test_both_ways_of_meeting_bases_at_infinity(mode):
    """The accumulation kernel exists twice: tables with many points at infinity (real circuits: b_query) let a lane sit such a
    base out, tables with next to none send the rare one through the general code by a wavefront vote; the key load picks per table
    from a count.  Both kernels over both kinds of data — a dense circuit, a sha-like one whose B matrix leaves most variables out,
    an ad-hoc MSM with infinite and repeated bases — must give the oracle's results."""
    c2 = native.Context(0, emu_library())
    c2.tune("skip_inf", mode)
    try:
        for curve, kind, n in ((BN254, "dense", 21), (BN254, "sha", 45), (BLS12_381, "sha", 30)):
            oc = cpu.Circuit.synth(curve.curve_id, n, 0x5EED0040 + n, kind)
            tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            opk = cpu.ProvingKey.setup(oc, tox)
            z = oc.assignment()
            cs = native.ConstraintSystem(c2, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
            pk = native.ProvingKey(c2, curve.curve_id, opk.serialize())
            assert native.prove_g16(c2, pk, cs, z, 11, 13) == cpu.trapdoor(oc, tox, z, 11, 13), (curve.name, kind)
        rnd = random.Random(77)
        b1, b2, ks = _rand_points(BN254, 40, rnd)
        assert c2.msm(0, 1, b1, ks) == cpu.msm(0, 1, b1, ks) and c2.msm(0, 2, b2, ks) == cpu.msm(0, 2, b2, ks)
    finally:
        c2.close()


@pytest.mark.parametrize("mode", [1, 2], ids=["own-list", "common-list"])
def test_b_family_on_a_list_of_its_own(mode):
    """Variables that do not occur in the B matrix have the point at infinity in b_g1_query AND b_g2_query (a third of the Poseidon
    chain's); a GM17 key holds it for the same half of its variables in a_query, c_query_2 and b_query.  The tables with many such
    bases form a family whose MSMs pair with a sorted list of their own without the variables at infinity in all of them
    (zkhip_pk::thin_mask: one more counting sort, the most expensive MSMs shrink by that share).  Forced on and off here
    (ZKHIP_TUNE_B_SORT) over circuits with and without such variables: Groth16 on both curves (b1 and b2 on the thinned list, A and
    L fused on the common one), GM17 (a, c2 and b on it, c1 alone on the common one), the unfused launches, a sharded key (the
    bitmap of a shard covers its own index range), a key image."""
    c2 = native.Context(0, emu_library())
    c2.tune("b_sort", mode)
    try:
        for curve, kind, n in ((BN254, "sha", 45), (BN254, "dense", 21), (BLS12_381, "sha", 30)):
            oc = cpu.Circuit.synth(curve.curve_id, n, 0x5EED0050 + n, kind)
            tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            opk = cpu.ProvingKey.setup(oc, tox)
            z = oc.assignment()
            cs = native.ConstraintSystem(c2, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
            raw = opk.serialize()
            pk = native.ProvingKey(c2, curve.curve_id, raw)
            want = cpu.trapdoor(oc, tox, z, 11, 13)
            assert native.prove_g16(c2, pk, cs, z, 11, 13) == want, (curve.name, kind)
            proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z, z, z]), [(11, 13), (5, 6), (11, 13)])
            assert proofs[0] == want and proofs[2] == want
            assert native.prove_g16(c2, native.ProvingKey.from_image(c2, curve.curve_id, pk.export_image()), cs, z, 11, 13) == want
            shards = [native.ProvingKey(c2, curve.curve_id, raw, rank=k, world=3) for k in range(3)]
            parts = [native.prove_g16_partial(c2, sh, cs, z, 11, 13) for sh in shards]
            assert native.combine_g16(c2, shards[0], parts, 11, 13) == want
            c2.tune("fuse_z", 0)
            assert native.prove_g16(c2, pk, cs, z, 11, 13) == want
            c2.tune("fuse_z", 1)
            if curve is BN254:
                tb17 = tox[:96] + tox[128:160]
                pk17 = native.ProvingKey(c2, 0, cpu.Gm17ProvingKey.setup(oc, tb17).serialize(), scheme="gm17")
                assert native.prove_gm17(c2, pk17, cs, z, 21, 7, 23) == cpu.gm17_trapdoor(oc, tb17, z, 21, 23), kind
        # a circuit the flattener makes: one Poseidon hash, a third of whose variables never occur in B
        from zokrates_amd import poseidon
        ch = poseidon.chain(0, 1)
        absent = ch.m - len(np.unique(ch.mats()[1][1]))
        assert absent * 10 > ch.m
        oc = cpu.Circuit.from_csr(0, ch.n, ch.l, ch.w, ch.mats())
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(BN254))
        z = ch.assignment(3)
        cs = native.ConstraintSystem(c2, 0, ch.n, ch.l, ch.w, ch.mats())
        pk = native.ProvingKey(c2, 0, cpu.ProvingKey.setup(oc, tox).serialize())
        assert native.prove_g16(c2, pk, cs, z, 5, 6) == cpu.trapdoor(oc, tox, z, 5, 6)
    finally:
        c2.close()


def test_schedule_does_not_change_proofs(ctx):
    schedule_invariance(ctx, logn=5, kinds=("dense",))


def test_stream_plan_does_not_change_proofs():
    stream_plan_invariance(lambda: native.Context(0, emu_library()), logn=5)


def test_two_pass_prove(ctx):
    """Force the two-pass NTT (sigma order, permuted h_query) inside the full prover."""
    os.environ["ZKHIP_NTT_SINGLE_MAX_LOG"] = "2"
    c2 = native.Context(0, emu_library())
    try:
        curve = BN254
        oc = cpu.Circuit.synth(0, 30, 0x5EED0008)   # N = 32 -> N1 = 4, N2 = 8
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        opk = cpu.ProvingKey.setup(oc, tox)
        z = oc.assignment()
        cs = native.ConstraintSystem(c2, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
        pk = native.ProvingKey(c2, 0, opk.serialize())
        assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
        assert native.prove_g16(c2, pk, cs, z, 11, 13) == cpu.trapdoor(oc, tox, z, 11, 13)
    finally:
        os.environ.pop("ZKHIP_NTT_SINGLE_MAX_LOG")
        c2.close()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("sub", [2, 3])
def test_ntt_three_passes(curve, sub):
    """Domains above 2^22 take three passes (N = N1 * N2 * N3, sigma order with three digits).  ZKHIP_TUNE_NTT_MAX_SUBLOG pulls the
    threshold down so that the same launchers and kernels run here on 2^5 .. 2^9 points: every split shape (equal factors, a short
    first one, odd sub-lengths) against the oracle's radix-2 transform."""
    c2 = native.Context(0, emu_library())
    c2.tune("ntt_max_sublog", sub)
    rnd = random.Random(60 + sub)
    try:
        for logn in range(2 * sub + 1, 3 * sub + 1):
            a = le([rnd.randrange(curve.r) for _ in range(1 << logn)])
            for d in ("fft", "ifft", "coset_fft", "coset_ifft"):
                assert c2.ntt(curve.curve_id, a, d).tobytes() == cpu.ntt(curve.curve_id, a, d).tobytes(), (logn, d)
        with pytest.raises(native.ZkhipError):                  # beyond three sub-transforms of the (test-sized) length
            c2.ntt(curve.curve_id, le([1] * (1 << (3 * sub + 1))), "fft")
    finally:
        c2.close()


@pytest.mark.parametrize("logn,sub", [(5, 2), (6, 2), (7, 3), (8, 3)])
def test_three_pass_prove(logn, sub):
    """The whole prover over a three-pass domain: the witness map's kind-b transforms (sigma order in, natural out), h left in the
    three-digit sigma order and paired with the h bases the key load permuted the same way — Groth16 and GM17."""
    c2 = native.Context(0, emu_library())
    c2.tune("ntt_max_sublog", sub)
    try:
        curve = BN254
        oc = cpu.Circuit.synth(0, (1 << logn) - 2, 0x5EED0030 + logn)
        assert oc.N == 1 << logn
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        opk = cpu.ProvingKey.setup(oc, tox)
        z = oc.assignment()
        cs = native.ConstraintSystem(c2, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
        pk = native.ProvingKey(c2, 0, opk.serialize())
        assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()
        assert native.prove_g16(c2, pk, cs, z, 11, 13) == cpu.trapdoor(oc, tox, z, 11, 13)
        proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z, z]), [(3, 4), (5, 6)])
        assert proofs == [cpu.trapdoor(oc, tox, z, 3, 4), cpu.trapdoor(oc, tox, z, 5, 6)]
        if logn + 1 <= 3 * sub:                                 # GM17: SAP domain 2^(logn + 1) must fit three (test-sized) passes
            t4 = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            tb17 = t4[:96] + t4[128:160]                        # alpha, beta, gamma, t
            opk17 = cpu.Gm17ProvingKey.setup(oc, tb17)
            pk17 = native.ProvingKey(c2, 0, opk17.serialize(), scheme="gm17")
            assert native.prove_gm17(c2, pk17, cs, z, 21, 7, 23) == cpu.gm17_trapdoor(oc, tb17, z, 21, 23)
        # a key ordered for this split is refused once the split changes (its h bases would pair with the wrong coefficients)
        c2.tune("ntt_max_sublog", 11)
        with pytest.raises(native.ZkhipError):
            native.prove_g16(c2, pk, cs, z, 11, 13)
    finally:
        c2.close()


def test_witness_map_with_paired_tiles(ctx):
    """A two-pass domain whose passes run 8 tiles per vector: the launches over a and b take the XCD-paired order (kernels_ntt.cuh
    ntt_tile_of: workgroups L and L + 8 share a tile of the factor table) — an affinity only, the result is the oracle's h."""
    oc = cpu.Circuit.synth(0, (1 << 13) - 2, 0x5EED0123)
    assert oc.N == 1 << 13
    z = oc.assignment()
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    assert cs.witness_map(z).tobytes() == cpu.witness_map(oc, z).tobytes()


def test_error_paths(ctx):
    curve = BN254
    oc = cpu.Circuit.synth(0, 5, 1)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    raw = cpu.ProvingKey.setup(oc, tox).serialize()
    with pytest.raises(native.ZkhipError) as e:
        native.ProvingKey(ctx, 0, raw[:-1])
    assert e.value.code == -2
    with pytest.raises(native.ZkhipError) as e:
        native.ProvingKey(ctx, 7, raw)
    assert e.value.code == -1
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, 0, raw)
    z = oc.assignment().copy()
    z[0] = 2                                       # z[0] != 1
    with pytest.raises(native.ZkhipError) as e:
        native.prove_g16(ctx, pk, cs, z, 1, 2)
    assert e.value.code == -1
    for bad in (BN254.r, (1 << 256) - 1):         # an entry >= r: ark's FromBytes rejects it, and so do all three entry points
        z = oc.assignment().copy()
        z[3 * 32:4 * 32] = np.frombuffer(int(bad).to_bytes(32, "little"), dtype=np.uint8)
        with pytest.raises(native.ZkhipError) as e:
            native.prove_g16(ctx, pk, cs, z, 1, 2)
        assert e.value.code == -1 and "canonical" in str(e.value)
        with pytest.raises(native.ZkhipError) as e:
            native.Assignment(ctx, cs, z)
        assert e.value.code == -1
        with pytest.raises(native.ZkhipError):
            native.prove_g16_partial(ctx, pk, cs, z, 1, 2)
    assert native.prove_g16(ctx, pk, cs, oc.assignment(), 1, 2) == cpu.trapdoor(oc, tox, oc.assignment(), 1, 2)   # the context stays usable
    oc2 = cpu.Circuit.synth(0, 9, 1)               # key / circuit mismatch
    cs2 = native.ConstraintSystem(ctx, 0, oc2.n, oc2.l, oc2.w, [oc2.csr(k) for k in range(3)])
    with pytest.raises(native.ZkhipError):
        native.prove_g16(ctx, pk, cs2, oc2.assignment(), 1, 2)
    with pytest.raises(native.ZkhipError):         # bad column index
        rp, col, val = oc.csr(0)
        col = col.copy(); col[0] = 10 ** 6
        native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [(rp, col, val), oc.csr(1), oc.csr(2)])
