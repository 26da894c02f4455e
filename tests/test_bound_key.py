"""A proving key bound to its constraint system (zkhip_pk_bind_r1cs, csrc/bind.cuh): the transforms that lead from the quotient's
evaluations to h's coefficients are applied to the key's bases once, a proof takes four transforms instead of six and never touches
the C matrix — and the proof bytes must not move.  Here on the TEST-ONLY emulator (same kernel sources) against the oracle; the
`-m gpu` copies of these checks are in test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle import cpu
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

from emu_util import emu_library

CURVES = [BN254, BLS12_381]


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0, emu_library())
    assert "EMULATOR" in c.describe()
    yield c
    c.close()


def bound_key_checks(ctx, curve, oc, seeds=((0x1234567, 0x89abcdef0123), (0, 77), (5, 0))):
    """Every entry point of the prover, unbound and bound, against the closed-form trapdoor proof."""
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    opk = cpu.ProvingKey.setup(oc, tox)
    z = oc.assignment()
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, curve.curve_id, opk.serialize())
    want = [cpu.trapdoor(oc, tox, z, r % curve.r, s % curve.r) for r, s in seeds]
    rs = [(r % curve.r, s % curve.r) for r, s in seeds]
    assert not pk.is_bound(cs)
    assert [native.prove_g16(ctx, pk, cs, z, r, s) for r, s in rs] == want
    pk.bind(cs)
    assert pk.is_bound(cs)
    assert [native.prove_g16(ctx, pk, cs, z, r, s) for r, s in rs] == want
    za = native.Assignment(ctx, cs, z)
    assert native.prove_g16_resident(ctx, pk, cs, za, *rs[0]) == want[0]
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za] * len(rs), rs)
    assert proofs == want
    proofs, _ = native.prove_g16_batch(ctx, pk, cs, np.concatenate([z] * len(rs)), rs)
    assert proofs == want
    # another copy of the same system is another system: the key's own tables serve it (same bytes, the unbound way)
    cs2 = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    assert not pk.is_bound(cs2)
    assert native.prove_g16(ctx, pk, cs2, z, *rs[0]) == want[0]
    # binding again (to the copy), unbinding, and the key is as it was loaded
    pk.bind(cs2)
    assert pk.is_bound(cs2) and not pk.is_bound(cs)
    assert native.prove_g16(ctx, pk, cs2, z, *rs[0]) == want[0]
    assert native.prove_g16(ctx, pk, cs, z, *rs[0]) == want[0]
    pk.unbind()
    assert not pk.is_bound(cs2)
    assert native.prove_g16(ctx, pk, cs2, z, *rs[0]) == want[0]
    return pk, cs, z, tox


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("kind", ["dense", "sha"])
def test_bound_key_proves_the_same_bytes(ctx, curve, kind):
    n = 14 if kind == "dense" else 29
    oc = cpu.Circuit.synth(curve.curve_id, n, 0x5EED0040, kind)
    bound_key_checks(ctx, curve, oc)


def test_bound_key_with_an_unsatisfying_assignment(ctx):
    """What the reference computes for an assignment that does NOT satisfy the system (ark's witness_map divides anyway and the MSM
    ignores the top coefficient) is what the bound key computes: the algorithmic oracle on a corrupted assignment."""
    curve = BN254
    oc = cpu.Circuit.synth(0, 30, 0x5EED0041)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    opk = cpu.ProvingKey.setup(oc, tox)
    z = np.array(oc.assignment(), copy=True)
    z[32 * 7] ^= 1                                            # one bit of one witness value
    want, _ = cpu.prove(oc, opk, z, 11, 13)
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, 0, opk.serialize())
    assert native.prove_g16(ctx, pk, cs, z, 11, 13) == want
    pk.bind(cs)
    assert native.prove_g16(ctx, pk, cs, z, 11, 13) == want


def test_bound_key_over_two_and_three_pass_domains():
    """The bases come out of the key's sigma order (whatever the split of the context's plan is) and H' pairs with NATURAL-order
    evaluations: two passes (sigma order of two digits) and three."""
    for sub, logn in ((2, 4), (2, 6), (3, 7)):
        c2 = native.Context(0, emu_library())
        c2.tune("ntt_max_sublog", sub)
        if logn <= 2 * sub:
            c2.tune("ntt_single_max_log", 1)
        try:
            oc = cpu.Circuit.synth(0, (1 << logn) - 2, 0x5EED0050 + logn)
            assert oc.N == 1 << logn
            bound_key_checks(c2, BN254, oc, seeds=((3, 4),))
        finally:
            c2.close()


def test_bound_key_with_heavy_columns_and_general_coefficients(ctx):
    """C with a variable that occurs in most rows (its products meet in the workgroup's tree), coefficients other than +-1 (the
    scalar multiplication of the per-entry products), a public variable in C (L' is finite where l_query is padding) and an
    empty column."""
    curve = BN254
    r = curve.r
    l, w, n = 3, 9, 70                                         # ONE, two public inputs; N = 128
    m = l + w
    import random
    rnd = random.Random(99)
    zv = [1] + [rnd.randrange(r) for _ in range(m - 1)]
    rows = [[], [], []]
    for k in range(n):
        a = {rnd.randrange(m): rnd.randrange(1, r) for _ in range(2)}
        b = {rnd.randrange(m): rnd.randrange(1, r) for _ in range(2)}
        av = sum(c * zv[v] for v, c in a.items()) % r
        bv = sum(c * zv[v] for v, c in b.items()) % r
        # c: the heavy column 0 (ONE) in every row, variable 1 (public) with a large coefficient in some, a +-1 entry, never variable m-1
        c = {0: 0}
        if k % 3 == 0:
            c[1] = rnd.randrange(2, r)
        v = 3 + rnd.randrange(m - 4)
        c[v] = 1 if k % 2 else r - 1
        rest = (av * bv - sum(cf * zv[u] for u, cf in c.items())) % r
        c[0] = rest                                            # ONE carries whatever is missing: the row holds
        rows[0].append(a); rows[1].append(b); rows[2].append(c)
    mats = []
    for mat in rows:
        rp, col, val = [0], [], []
        for row in mat:
            for v in sorted(row):
                col.append(v); val.append(row[v])
            rp.append(len(col))
        mats.append((np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint32),
                     np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in val), dtype=np.uint8)))
    oc = cpu.Circuit.from_csr(0, n, l, w, mats)
    z = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in zv), dtype=np.uint8)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    opk = cpu.ProvingKey.setup(oc, tox)
    want = cpu.trapdoor(oc, tox, z, 21, 22)
    assert cpu.prove(oc, opk, z, 21, 22)[0] == want
    cs = native.ConstraintSystem(ctx, 0, n, l, w, mats)
    pk = native.ProvingKey(ctx, 0, opk.serialize())
    assert native.prove_g16(ctx, pk, cs, z, 21, 22) == want
    pk.bind(cs)
    assert native.prove_g16(ctx, pk, cs, z, 21, 22) == want


def test_what_does_not_bind(ctx):
    curve = BN254
    oc = cpu.Circuit.synth(0, 13, 0x5EED0042)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    raw = cpu.ProvingKey.setup(oc, tox).serialize()
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    shard = native.ProvingKey(ctx, 0, raw, rank=0, world=2)
    with pytest.raises(native.ZkhipError):
        shard.bind(cs)                                         # a shard holds only its ranges of the bases: bind_shard takes the key file
    other = cpu.Circuit.synth(0, 40, 0x5EED0043)
    cs_other = native.ConstraintSystem(ctx, 0, other.n, other.l, other.w, [other.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, 0, raw)
    with pytest.raises(native.ZkhipError):
        pk.bind(cs_other)                                      # another domain
    assert not pk.is_bound(cs_other) and not pk.is_bound(cs)
    # a refused call leaves an earlier binding as it was (the argument checks come before anything of the key is touched)
    pk.bind(cs)
    with pytest.raises(native.ZkhipError):
        pk.bind(cs_other)
    assert pk.is_bound(cs)
    z = oc.assignment()
    assert native.prove_g16(ctx, pk, cs, z, 1, 2) == cpu.trapdoor(oc, tox, z, 1, 2)      # the refusals left the context usable
    t4 = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    tb17 = t4[:96] + t4[128:160]
    pk17 = native.ProvingKey(ctx, 0, cpu.Gm17ProvingKey.setup(oc, tb17).serialize(), scheme="gm17")
    with pytest.raises(native.ZkhipError):
        pk17.bind(cs_other)                                    # a GM17 key binds to ITS system only
    with pytest.raises(native.ZkhipError):
        pk.bind_shard(cs, cpu.Gm17ProvingKey.setup(oc, tb17).serialize())      # the key file of another scheme


def _gm17_case(curve, n, kind="dense", seed=0x5EED0090):
    oc = cpu.Circuit.synth(curve.curve_id, n, seed, kind)
    t5 = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    tb17 = t5[:96] + t5[128:160]
    return oc, tb17, cpu.Gm17ProvingKey.setup(oc, tb17), oc.assignment()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("kind,n", [("dense", 14), ("sha", 27), ("dense", 1)])
def test_gm17_key_bound_to_its_system_proves_the_same_bytes(ctx, curve, kind, n):
    """GM17 (/root/reference/zokrates_ark/src/gm17.rs:63): W's transform and its share of the quotient ride on the c_query_1 bases, the
    last transform on g_gamma2_z_t — two transforms per proof instead of four, the same three group elements: against the oracle's
    term-by-term restatement of ark-gm17 and its closed form, single and batched, and unbound again afterwards."""
    oc, tb17, opk, z = _gm17_case(curve, n, kind)
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    pk = native.ProvingKey(ctx, curve.curve_id, opk.serialize(), scheme="gm17")
    rnds = [(3, 5, 7), (0, 9, 0), (curve.r - 1, 1, 2)]
    want = [cpu.gm17_prove(oc, opk, z, *rnd)[0] for rnd in rnds]
    assert want[0] == cpu.gm17_trapdoor(oc, tb17, z, 3, 7)
    assert [native.prove_gm17(ctx, pk, cs, z, *rnd) for rnd in rnds] == want
    pk.bind(cs)
    assert pk.is_bound(cs)
    assert [native.prove_gm17(ctx, pk, cs, z, *rnd) for rnd in rnds] == want
    za = native.Assignment(ctx, cs, z)
    proofs, _ = native.prove_gm17_resident_batch(ctx, pk, cs, [za] * 3, rnds)
    assert proofs == want
    # an assignment that does NOT satisfy the system: the binding is linear, so bound and unbound agree there as well (what ark-gm17
    # makes of such an assignment is another matter — it folds the blinding into the quotient before a division that is no longer
    # exact, the device adds the blinding terms as group elements: INTEGRATION.md §7)
    zbad = np.array(z, copy=True)
    zbad[32 * (oc.l + 1)] ^= 1
    bad_bound = native.prove_gm17(ctx, pk, cs, zbad, 3, 5, 7)
    pk.unbind()
    assert not pk.is_bound(cs) and native.prove_gm17(ctx, pk, cs, z, *rnds[0]) == want[0]
    assert native.prove_gm17(ctx, pk, cs, zbad, 3, 5, 7) == bad_bound


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_shards_bind_from_the_key_file(ctx, world, scheme):
    """A shard holds its index ranges of the bases; the binding's transforms need all of them once: zkhip_pk_bind_r1cs_shard takes the
    key file, every shard keeps ITS ranges of H' / L' — the partial records of bound shards combine to the unsharded proof."""
    curve = BN254
    if scheme == "g16":
        oc = cpu.Circuit.synth(0, 29, 0x5EED00A0, "sha")
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        z = oc.assignment()
        want = cpu.trapdoor(oc, tox, z, 51, 52)
    else:
        oc, tb17, opk, z = _gm17_case(curve, 13)
        raw = opk.serialize()
        want = cpu.gm17_trapdoor(oc, tb17, z, 51, 52)
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    shards = [native.ProvingKey(ctx, 0, raw, rank=k, world=world, scheme=scheme) for k in range(world)]
    for bound in (False, True):
        if bound:
            for sh in shards:
                sh.bind_shard(cs, raw)
                assert sh.is_bound(cs)
        if scheme == "g16":
            parts = [native.prove_g16_partial(ctx, sh, cs, z, 51, 52) for sh in shards]
            assert native.combine_g16(ctx, shards[0], parts, 51, 52) == want, bound
        else:
            parts = [native.prove_gm17_partial(ctx, sh, cs, z, 51, 0, 52) for sh in shards]
            assert native.combine_gm17(ctx, shards[0], parts, 51, 0, 52) == want, bound
    # a whole key binds through the same entry point, and an image of a bound shard comes back bound-able without the transforms
    whole = native.ProvingKey(ctx, 0, raw, scheme=scheme)
    whole.bind_shard(cs, raw)
    prove = (lambda k: native.prove_g16(ctx, k, cs, z, 51, 52)) if scheme == "g16" else (lambda k: native.prove_gm17(ctx, k, cs, z, 51, 0, 52))
    assert whole.is_bound(cs) and prove(whole) == want


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_key_image_carries_the_bound_tables(ctx, scheme):
    """zkhip_pk_export of a bound key holds level 0 of H' / L' and the system's fingerprint; zkhip_pk_bind_r1cs on the imported key
    attaches them when the fingerprint agrees (no transforms), and recomputes for any other system."""
    curve = BN254
    if scheme == "g16":
        oc = cpu.Circuit.synth(0, 30, 0x5EED00B0)
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        z = oc.assignment()
        want = cpu.trapdoor(oc, tox, z, 61, 62)
        prove = lambda k, c: native.prove_g16(ctx, k, c, z, 61, 62)
    else:
        oc, tb17, opk, z = _gm17_case(curve, 14)
        raw = opk.serialize()
        want = cpu.gm17_trapdoor(oc, tb17, z, 61, 62)
        prove = lambda k, c: native.prove_gm17(ctx, k, c, z, 61, 0, 62)
    mats = [oc.csr(k) for k in range(3)]
    cs = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, mats)
    pk = native.ProvingKey(ctx, 0, raw, scheme=scheme)
    plain = pk.export_image()
    pk.bind(cs)
    image = pk.export_image()
    assert image.size > plain.size
    pk.close()
    back = native.ProvingKey.from_image(ctx, 0, image, scheme=scheme)
    assert not back.is_bound(cs)                                   # tables present, not attached to a system yet
    assert prove(back, cs) == want                                 # (the key's own tables)
    cs_again = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, mats)     # the "restarted process": the same system loaded again
    assert cs_again.fingerprint() == cs.fingerprint() and cs.fingerprint() != (0, 0)
    back.bind(cs_again)
    assert back.is_bound(cs_again) and prove(back, cs_again) == want
    # the same dimensions, other coefficients: another fingerprint — the binding is recomputed, not trusted
    rp, col, val = mats[2]
    val2 = np.array(val, copy=True)
    if val2.size:
        val2[0] ^= 1
    cs_other = native.ConstraintSystem(ctx, 0, oc.n, oc.l, oc.w, [mats[0], mats[1], (rp, col, val2)])
    assert cs_other.fingerprint() != cs.fingerprint()
    back.bind(cs_other)
    assert back.is_bound(cs_other) and not back.is_bound(cs_again)
    back.bind(cs_again)                                            # ... and back: recomputed, the same bytes
    assert prove(back, cs_again) == want
    # a truncated / inconsistent image is refused
    with pytest.raises(native.ZkhipError):
        native.ProvingKey.from_image(ctx, 0, image[:-5], scheme=scheme)


def test_multi_members_bind_together(ctx):
    """zkhip_multi_bind: one member computes the bound bases from the key file, every member installs its ranges; the members'
    proof is the unsharded one, for both schemes, with the host exchange and the gathered one."""
    multi_bind_checks(emu_library(), gathered=True)


def multi_bind_checks(lib, gathered):
    curve = BN254
    for scheme in ("g16", "gm17"):
        if scheme == "g16":
            oc = cpu.Circuit.synth(0, 37, 0x5EED00C0)
            tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            raw = cpu.ProvingKey.setup(oc, tox).serialize()
            z = oc.assignment()
            want = cpu.trapdoor(oc, tox, z, 71, 72)
        else:
            oc, tb17, opk, z = _gm17_case(curve, 12)
            raw = opk.serialize()
            want = cpu.gm17_trapdoor(oc, tb17, z, 71, 72)
        for members, rccl in ((3, False), (2, True)) if gathered else ((3, False), (8, False)):
            multi = native.Multi([0] * members, lib)
            try:
                if rccl:
                    multi.use_rccl(True)
                multi.load_constraint_system(0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
                multi.load_proving_key(0, raw, scheme=scheme)
                for bound in (False, True):
                    if bound:
                        multi.bind(raw)
                    got = multi.prove_g16(z, 71, 72) if scheme == "g16" else multi.prove_gm17(z, 71, 0, 72)
                    assert got == want, (scheme, members, rccl, bound)
                multi.unbind()
                got = multi.prove_g16(z, 71, 72) if scheme == "g16" else multi.prove_gm17(z, 71, 0, 72)
                assert got == want
            finally:
                multi.close()


def _bound_proof(c2, curve, oc, raw, cs, z, r_, s_, via_image=False):
    pk = native.ProvingKey(c2, curve.curve_id, raw)
    if via_image:
        pk = native.ProvingKey.from_image(c2, curve.curve_id, pk.export_image())
    pk.bind(cs)
    assert pk.is_bound(cs)
    return pk, native.prove_g16(c2, pk, cs, z, r_, s_)


def test_bound_key_under_every_table_and_list_setting():
    """The binding builds its two tables with the key's own shape and rides on the prover's other choices: thinned tables (every
    2nd / 3rd multiple, several bucket sets), 17-bit windows, a key that came from an image, the b family on a list of its own
    (L' leaves that list whatever l's family was: its public entries are finite), both ways of meeting bases at infinity, the
    unfused launches, one stream — same bytes everywhere."""
    c2 = native.Context(0, emu_library())
    try:
        curve = BN254
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        cases = []
        for kind, n in (("dense", 40), ("sha", 45)):
            oc = cpu.Circuit.synth(0, n, 0x5EED0060 + n, kind)
            raw = cpu.ProvingKey.setup(oc, tox).serialize()
            cs = native.ConstraintSystem(c2, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
            z = oc.assignment()
            cases.append((oc, raw, cs, z, cpu.trapdoor(oc, tox, z, 11, 13)))
        from zokrates_amd import poseidon
        ch = poseidon.chain(0, 1)                                 # a compiler-shaped circuit: a third of its variables never occur in B
        oc = cpu.Circuit.from_csr(0, ch.n, ch.l, ch.w, ch.mats())
        z = ch.assignment(3)
        cases.append((oc, cpu.ProvingKey.setup(oc, tox).serialize(), native.ConstraintSystem(c2, 0, ch.n, ch.l, ch.w, ch.mats()), z,
                      cpu.trapdoor(oc, tox, z, 11, 13)))
        # (which circuits a setting is run over: the table shapes over the dense and the boolean one, the lists and the infinities
        # over the two with a sparse B, the scheduling knobs over the dense one)
        settings = [({}, (0, 1, 2)), ({"msm_sets": 2}, (0, 1)), ({"msm_sets": 64}, (0,)), ({"msm_c": 17}, (0,)), ({"b_sort": 1}, (1, 2)), ({"b_sort": 2}, (1, 2)),
                    ({"skip_inf": 1}, (1, 2)), ({"skip_inf": 2}, (1,)), ({"fuse_z": 0}, (0, 1)), ({"serial": 1}, (0,)), ({"z_gate": 2}, (0,)), ({"slots": 1}, (0,))]
        defaults = {"msm_sets": 0, "msm_c": 0, "b_sort": 0, "skip_inf": 0, "fuse_z": 1, "serial": 0, "z_gate": 1, "slots": 3}
        for st, which in settings:
            for k, v in st.items():
                c2.tune(k, v)
            for i in which:
                oc, raw, cs, z, want = cases[i]
                pk, got = _bound_proof(c2, curve, oc, raw, cs, z, 11, 13, via_image=bool(st.get("msm_sets") == 2 or not st))
                assert got == want, (st, i)
                if not st or "slots" in st or "fuse_z" in st:
                    proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z, z]), [(11, 13), (11, 13)])
                    assert proofs == [want, want], (st, i)
                pk.close()
            for k in st:
                c2.tune(k, defaults[k])
    finally:
        c2.close()


def test_bound_key_on_bls12_381_with_a_sparse_b(ctx):
    """BLS12-381 (14-limb base field, 255-bit scalars: 16 windows) over a circuit most of whose variables never occur in B."""
    curve = BLS12_381
    oc = cpu.Circuit.synth(curve.curve_id, 30, 0x5EED0070, "sha")
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    raw = cpu.ProvingKey.setup(oc, tox).serialize()
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    z = oc.assignment()
    pk, got = _bound_proof(ctx, curve, oc, raw, cs, z, 21, 22)
    assert got == cpu.trapdoor(oc, tox, z, 21, 22) == cpu.prove(oc, cpu.ProvingKey.parse(curve.curve_id, raw), z, 21, 22)[0]


def _csr_from_rows(rows):
    rp, col, val = [0], [], []
    for row in rows:
        for v, c in row:                                          # (kept in the order given: duplicates and zeros stay what they are)
            col.append(v); val.append(c)
        rp.append(len(col))
    return (np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint32),
            np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in val), dtype=np.uint8) if val else np.zeros(0, dtype=np.uint8))


@pytest.mark.parametrize("shape", ["empty_c", "zero_and_duplicate_entries", "one_constraint", "c_on_one_only"])
def test_bound_key_edge_shapes_of_c(ctx, shape):
    """What the per-variable sums of the binding meet at the edges: a C matrix with no entry at all (every product a * b is 0), entries
    with coefficient 0 and the same (row, variable) twice, a single constraint (domain 2), C on the constant ONE only."""
    curve = BN254
    r = curve.r
    import random
    rnd = random.Random({"empty_c": 1, "zero_and_duplicate_entries": 2, "one_constraint": 3, "c_on_one_only": 4}[shape])
    l = 2
    if shape == "one_constraint":
        zv = [1, 5, 7, 35]
        A, B, C = [[(1, 1)]], [[(2, 1)]], [[(3, 1)]]
    else:
        n, w = 11, 6
        zv = [1] + [rnd.randrange(1, r) for _ in range(l - 1 + w)]
        A, B, C = [], [], []
        for k in range(n):
            a = [(rnd.randrange(len(zv)), rnd.randrange(1, r))]
            av = a[0][1] * zv[a[0][0]] % r
            if shape == "empty_c":
                b, c = [], []                                     # b = 0: a * 0 = 0
            elif shape == "c_on_one_only":
                b = [(rnd.randrange(len(zv)), rnd.randrange(1, r))]
                c = [(0, av * (b[0][1] * zv[b[0][0]] % r) % r)]
            else:
                b = [(rnd.randrange(len(zv)), rnd.randrange(1, r))]
                prod = av * (b[0][1] * zv[b[0][0]] % r) % r
                v = 2 + rnd.randrange(len(zv) - 2)
                half = rnd.randrange(r)
                inv = pow(zv[v], r - 2, r)
                # prod = (half + rest) * z_v, the variable listed twice, next to an entry with coefficient 0
                c = [(v, half), (1, 0), (v, (prod * inv - half) % r)]
            A.append(a); B.append(b); C.append(c)
    n = len(A)
    w = len(zv) - l
    mats = [_csr_from_rows(A), _csr_from_rows(B), _csr_from_rows(C)]
    oc = cpu.Circuit.from_csr(0, n, l, w, mats)
    z = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in zv), dtype=np.uint8)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    opk = cpu.ProvingKey.setup(oc, tox)
    want = cpu.prove(oc, opk, z, 31, 32)[0]
    assert want == cpu.trapdoor(oc, tox, z, 31, 32)
    cs = native.ConstraintSystem(ctx, 0, n, l, w, mats)
    pk = native.ProvingKey(ctx, 0, opk.serialize())
    assert native.prove_g16(ctx, pk, cs, z, 31, 32) == want
    pk.bind(cs)
    assert pk.is_bound(cs) and native.prove_g16(ctx, pk, cs, z, 31, 32) == want


@pytest.mark.parametrize("setting", ["0", "1", "2"])
def test_lone_proofs_hold_their_g1_lanes_for_the_g2_accumulation(setting):
    """ZKHIP_G2_HEAD_START (0 never, 1 over a bound key, 2 always): where the G2 accumulation runs one wave per SIMD (BLS12-381) a LONE
    proof's G1 lanes also wait for the end of that accumulation.  A scheduling rule: the same proofs at every setting, bound and as
    loaded, single and batched (the emulator runs the streams in order — the GPU copy of this check is in test_gpu_parity.py)."""
    os.environ["ZKHIP_G2_HEAD_START"] = setting
    try:
        c2 = native.Context(0, emu_library())
    finally:
        os.environ.pop("ZKHIP_G2_HEAD_START")
    try:
        for curve in (BLS12_381, BN254):
            oc = cpu.Circuit.synth(curve.curve_id, 21, 0x5EED0080, "sha")
            tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            raw = cpu.ProvingKey.setup(oc, tox).serialize()
            cs = native.ConstraintSystem(c2, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
            z = oc.assignment()
            want = cpu.trapdoor(oc, tox, z, 41, 42)
            pk = native.ProvingKey(c2, curve.curve_id, raw)
            za = native.Assignment(c2, cs, z)
            assert native.prove_g16(c2, pk, cs, z, 41, 42) == want and native.prove_g16_resident(c2, pk, cs, za, 41, 42) == want
            pk.bind(cs)
            assert native.prove_g16(c2, pk, cs, z, 41, 42) == want and native.prove_g16_resident(c2, pk, cs, za, 41, 42) == want
            proofs, _ = native.prove_g16_resident_batch(c2, pk, cs, [za] * 3, [(41, 42)] * 3)
            assert proofs == [want] * 3
    finally:
        c2.close()


def test_members_split_the_witness_map(monkeypatch):
    """Bound members of a multi-GPU proof split its witness map — even members transform a, odd members b, partners copy each other's
    vector (zkhip_multi_transform_split; north_star's "NTT domain shard") — and the proof is the unsharded one: 2, 3 (an odd member
    without a partner of its own) and 8 members, the host exchange and the gathered one, split off again, domains of one, two and
    three passes."""
    monkeypatch.setenv("ZKHIP_SPLIT_MIN_LOG", "0")
    split_checks(emu_library(), ((37, None), (60, 3), (200, 3)), gathered=True)


def split_checks(lib, sizes, gathered):
    curve = BN254
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    for n_con, sub in sizes:
        oc = cpu.Circuit.synth(0, n_con, 0x5EED00E0 + n_con, "sha" if n_con == 60 else "dense")
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        z = oc.assignment()
        want = cpu.trapdoor(oc, tox, z, 81, 82)
        for members, rccl in (((2, False), (3, False), (8, False)) + (((2, True),) if gathered else ())) if sub is None else ((3, False),):
            multi = native.Multi([0] * members, lib)
            try:
                if sub:
                    for k in range(members):
                        c = multi.member_context(k)
                        c.tune("ntt_max_sublog", sub)
                        c.tune("ntt_single_max_log", 1)
                if rccl:
                    multi.use_rccl(True)
                multi.load_constraint_system(0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
                multi.load_proving_key(0, raw)
                assert multi.prove_g16(z, 81, 82) == want and not multi.last_split()          # unbound: every member runs the whole map
                multi.bind(raw)
                assert multi.prove_g16(z, 81, 82) == want and multi.last_split(), (n_con, members, rccl)
                assert multi.prove_g16(z, 81, 82) == want                                      # ... again: the slots are reusable
                assert multi.transform_split(False) is True
                assert multi.prove_g16(z, 81, 82) == want and not multi.last_split()
                multi.transform_split(True)
                multi.unbind()
                assert multi.prove_g16(z, 81, 82) == want and not multi.last_split()
            finally:
                multi.close()


def test_split_entry_points_for_ranks_in_separate_processes(monkeypatch):
    """zkhip_prove_g16_split_begin / _end: what a rank of a multi-process prover calls around its exchange — here two contexts in one
    process stand for two ranks: each transforms its half, they swap, both finish, the records combine to the unsharded proof."""
    split_entry_point_checks(emu_library())


def split_entry_point_checks(lib):
    curve = BN254
    oc = cpu.Circuit.synth(0, 29, 0x5EED00F0)
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
    raw = cpu.ProvingKey.setup(oc, tox).serialize()
    z = oc.assignment()
    want = cpu.trapdoor(oc, tox, z, 91, 92)
    ranks = []
    for k in range(2):
        c = native.Context(0, lib)
        cs = native.ConstraintSystem(c, 0, oc.n, oc.l, oc.w, [oc.csr(q) for q in range(3)])
        sh = native.ProvingKey(c, 0, raw, rank=k, world=2)
        ranks.append((c, cs, sh))
    with pytest.raises(native.ZkhipError):
        native.prove_g16_split_begin(ranks[0][0], ranks[0][2], ranks[0][1], z, 91, 92, 0)      # not bound: nothing to split
    for c, cs, sh in ranks:
        sh.bind_shard(cs, raw)
    halves = [native.prove_g16_split_begin(c, sh, cs, z, 91, 92, k) for k, (c, cs, sh) in enumerate(ranks)]
    assert halves[0].tobytes() != halves[1].tobytes()
    parts = [native.prove_g16_split_end(c, sh, cs, halves[1 - k]) for k, (c, cs, sh) in enumerate(ranks)]
    assert native.combine_g16(ranks[0][0], ranks[0][2], parts, 91, 92) == want
    # the same ranks prove the ordinary way afterwards
    parts = [native.prove_g16_partial(c, sh, cs, z, 91, 92) for c, cs, sh in ranks]
    assert native.combine_g16(ranks[0][0], ranks[0][2], parts, 91, 92) == want
    for c, cs, sh in ranks:
        sh.close(); c.close()
