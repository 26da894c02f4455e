"""`-m gpu` copies of tests/test_bound_key.py's round-6 checks — GM17 keys, shards and members of a multi-GPU prover bound to their
constraint system, key images that carry the bound tables — on the real device through the C ABI, against the CPU oracle."""
import pytest

from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

import test_bound_key as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gctx():
    c = native.Context(0)
    assert "gfx950" in c.describe() and "EMULATOR" not in c.describe()
    yield c
    c.close()


@pytest.mark.parametrize("curve", [BN254, BLS12_381], ids=lambda c: c.name)
def test_gpu_gm17_bound(gctx, curve):
    T.test_gm17_key_bound_to_its_system_proves_the_same_bytes(gctx, curve, "sha", 27)
    T.test_gm17_key_bound_to_its_system_proves_the_same_bytes(gctx, curve, "dense", 1)


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_gpu_shards_bind_from_the_key_file(gctx, scheme):
    T.test_shards_bind_from_the_key_file(gctx, 3, scheme)


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_gpu_key_image_carries_the_bound_tables(gctx, scheme):
    T.test_key_image_carries_the_bound_tables(gctx, scheme)


def test_gpu_multi_members_bind_together():
    T.multi_bind_checks(native.default_library(), gathered=False)      # members share the one GPU of the box: the host exchange


def test_gpu_what_does_not_bind(gctx):
    T.test_what_does_not_bind(gctx)


@pytest.mark.parametrize("sched", [1, 2, 3])
def test_gpu_lone_proof_layouts(sched):
    """ZKHIP_TUNE_LONE_SCHED: a lone proof's G2 accumulation at one workgroup per CU, its G1 lanes behind the h sort — scheduling
    only: the same bytes at every setting, bound and as loaded, both curves, and batches untouched."""
    import numpy as np
    from oracle import cpu
    from oracle import groth16 as g16
    c2 = native.Context(0)
    try:
        c2.tune("lone_sched", sched)
        for curve in (BN254, BLS12_381):
            oc = cpu.Circuit.synth(curve.curve_id, 3000, 0x5EED00D0, "sha")
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


def test_gpu_members_split_the_witness_map(monkeypatch):
    """Even members transform a, odd members b, partners copy each other's vector on the device they share here (between GPUs: an xGMI
    peer copy) — small domains forced to split, and one at the default threshold (2^18) with 8 members."""
    monkeypatch.setenv("ZKHIP_SPLIT_MIN_LOG", "0")
    T.split_checks(native.default_library(), ((37, None), (3000, None)), gathered=False)
    monkeypatch.delenv("ZKHIP_SPLIT_MIN_LOG")
    T.split_checks(native.default_library(), (((1 << 18) - 2, 11),), gathered=False)      # (sub = 11: the default plan) three members, 2^18


def test_gpu_split_entry_points():
    T.split_entry_point_checks(native.default_library())
