"""Shared by tests/test_emu_kernels.py (emulator) and tests/test_gpu_parity.py (MI355X): the scheduling knobs of a context
do not change a proof."""
import numpy as np

from oracle import cpu
from oracle import groth16 as g16
from oracle.fields import BN254
from zokrates_amd import native


def schedule_invariance(c2, logn=5, kinds=("dense", "sha")):
    """The order in which a proof's kernels are released (`z_gate`), A / B1 / L as one launch or three (`fuse_z`) and the
    slices per launch (`msm_fused_waves`) are scheduling only: the proof bytes do not move, single or pipelined,
    Groth16 or GM17.  """
    from oracle import gm17
    for kind in kinds:
        oc = cpu.Circuit.synth(0, (1 << logn) - 2, 0x5C4ED + logn, kind)
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(BN254))
        cs = native.ConstraintSystem(c2, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
        pk = native.ProvingKey(c2, 0, cpu.ProvingKey.setup(oc, tox).serialize())
        z = oc.assignment()
        rs = [(11, 13), (0, 5), (7, 0), (1 << 200, 3)]
        want = [cpu.trapdoor(oc, tox, z, a, b) for a, b in rs]
        gtox = gm17.Toxic.from_seed(BN254)
        gpk = native.ProvingKey(c2, 0, cpu.Gm17ProvingKey.setup(oc, cpu.gm17_toxic_bytes(gtox)).serialize(), scheme="gm17")
        gwant = None
        try:
            for gate in (0, 1, 2):
                for fuse, waves in ((1, 0), (1, 3), (0, 0)):
                    c2.tune("z_gate", gate); c2.tune("fuse_z", fuse); c2.tune("msm_fused_waves", waves)
                    assert native.prove_g16(c2, pk, cs, z, *rs[0]) == want[0], (kind, gate, fuse, waves)
                    proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z] * len(rs)), rs)
                    assert proofs == want, (kind, gate, fuse, waves)
                    g = native.prove_gm17(c2, gpk, cs, z, 21, 22, 23)
                    gwant = gwant or g
                    assert g == gwant, (kind, gate, fuse, waves)
            # the placement pass of the sort, one level (round 5) against two (round 6), and the lone-proof layouts: the same bytes
            for two_level, lone in ((0, 0), (1, 3), (0, 1)):
                c2.tune("sort_two_level", two_level); c2.tune("lone_sched", lone); c2.tune("fold_lines", two_level + lone % 2)
                c2.tune("fold_hop", two_level + lone % 2)      # (the fold chains on second streams of their lanes: never / every lane / the G2 lane)
                assert native.prove_g16(c2, pk, cs, z, *rs[0]) == want[0], (kind, two_level, lone)
                proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z] * len(rs)), rs)
                assert proofs == want, (kind, two_level, lone)
                assert native.prove_gm17(c2, gpk, cs, z, 21, 22, 23) == gwant
        finally:
            c2.tune("z_gate", 1); c2.tune("fuse_z", 1); c2.tune("msm_fused_waves", 0); c2.tune("sort_two_level", 1); c2.tune("lone_sched", 0); c2.tune("fold_lines", 0); c2.tune("fold_hop", 0)
        assert gwant == cpu.gm17_trapdoor(oc, cpu.gm17_toxic_bytes(gtox), z, 21, 23)


def stream_plan_invariance(make_ctx, logn=5):
    """A resident prover's stream plan (ZKHIP_TUNE_PIPE_PLAN, core.cuh make_pipe_streams: the context's streams made in one go at its
    first proof, the fold chains on streams of their own, a lone proof's witness map on another) is placement only: lone proofs, a
    pipelined batch, a bound key and GM17 give the bytes the oracle gives.  The plan is chosen before the first proof and refused after."""
    from oracle import gm17
    import pytest
    c2 = make_ctx()
    try:
        c2.tune("pipe_plan", 1)
        oc = cpu.Circuit.synth(0, (1 << logn) - 2, 0x51A7 + logn, "sha")
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(BN254))
        cs = native.ConstraintSystem(c2, 0, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
        pk = native.ProvingKey(c2, 0, cpu.ProvingKey.setup(oc, tox).serialize())
        z = oc.assignment()
        rs = [(11, 13), (0, 5), (7, 0), (1 << 200, 3), (9, 9), (2, 1)]
        want = [cpu.trapdoor(oc, tox, z, a, b) for a, b in rs]
        assert native.prove_g16(c2, pk, cs, z, *rs[0]) == want[0]
        with pytest.raises(native.ZkhipError):
            c2.tune("pipe_plan", 0)           # the streams exist now
        proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z] * len(rs)), rs)
        assert proofs == want
        pk.bind(cs)
        assert native.prove_g16(c2, pk, cs, z, *rs[1]) == want[1]
        proofs, _ = native.prove_g16_batch(c2, pk, cs, np.concatenate([z] * len(rs)), rs)
        assert proofs == want
        gtox = gm17.Toxic.from_seed(BN254)
        gpk = native.ProvingKey(c2, 0, cpu.Gm17ProvingKey.setup(oc, cpu.gm17_toxic_bytes(gtox)).serialize(), scheme="gm17")
        assert native.prove_gm17(c2, gpk, cs, z, 21, 22, 23) == cpu.gm17_trapdoor(oc, cpu.gm17_toxic_bytes(gtox), z, 21, 23)
    finally:
        c2.close()
