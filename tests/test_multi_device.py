"""One proof across several GPUs of ONE process through the C ABI alone (zkhip_ctx_create_multi, zkhip_multi_*,
zkhip_prove_g16_multi / zkhip_prove_gm17_multi): no Python collective, no torch.  The members may share a device, which
is how the path runs on a one-GPU box; the result must equal the single-context proof and the oracle's closed form.
Reference behaviour to match: one `generate_proof` call, one proof (zokrates_cli/src/ops/generate_proof.rs:187)."""
import numpy as np
import pytest

from emu_util import emu_library
from oracle import cpu
from oracle import groth16 as g16
from oracle import gm17
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native, synth


def _checks(lib, devices, curve, log_domain, kind="dense"):
    cid = curve.curve_id
    circ = synth.circuit(cid, log_domain, kind=kind, seed=0xD1CE + log_domain)
    z = circ.assignment(0x5EED + log_domain)
    oc = cpu.Circuit.from_csr(cid, circ.n, circ.l, circ.w, circ.mats())
    multi = native.Multi(devices, lib)
    assert len(multi) == len(devices)
    try:
        multi.load_constraint_system(cid, circ.n, circ.l, circ.w, circ.mats())
        # Groth16
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        multi.load_proving_key(cid, raw)
        r, s = 0x1234567 % curve.r, 0x7654321 % curve.r
        got, tm = multi.prove_g16(z, r, s, want_timings=True)
        assert got == cpu.trapdoor(oc, tox, z, r, s)
        assert tm["total_ms"] > 0
        assert multi.prove_g16(z, 0, 0) == cpu.trapdoor(oc, tox, z, 0, 0)
        # the same proof from one context holding the whole key
        ctx = multi.member_context(0)
        cs = native.ConstraintSystem(ctx, cid, circ.n, circ.l, circ.w, circ.mats())
        assert native.prove_g16(ctx, native.ProvingKey(ctx, cid, raw), cs, z, r, s) == got
        with pytest.raises(native.ZkhipError):          # a Groth16 key does not prove GM17
            multi.prove_gm17(z, 1, 2, 3)
        # throughput mode on the same members: whole key everywhere, independent proofs dealt round-robin
        multi.load_proving_key_replicas(cid, raw)
        zs = [circ.assignment(0x5EED + 7 * i) for i in range(2 * len(devices) + 1)]
        rss = [(1000 + i, 2000 + 3 * i) for i in range(len(zs))]
        proofs, _ = multi.prove_g16_batch(zs, rss)
        assert proofs == [cpu.trapdoor(oc, tox, zs[i], *rss[i]) for i in range(len(zs))]
        assert multi.prove_g16_batch(zs[:1], rss[:1])[0] == proofs[:1]          # fewer proofs than members
        assert multi.prove_g16_batch([], [])[0] == []
        with pytest.raises(native.ZkhipError):          # whole keys do not take part in a sharded proof
            multi.prove_g16(z, r, s)
        multi.load_proving_key(cid, raw)
        # GM17 on the same members
        tox17 = gm17.Toxic.from_seed(curve)
        tb17 = cpu.gm17_toxic_bytes(tox17)
        raw17 = cpu.Gm17ProvingKey.setup(oc, tb17).serialize()
        multi.load_proving_key(cid, raw17, scheme="gm17")
        d1, d2, r_ = 0xabcdef % curve.r, 77, 0x13579 % curve.r
        assert multi.prove_gm17(z, d1, d2, r_) == cpu.gm17_trapdoor(oc, tb17, z, d1, r_)
        # errors surface with the member's message
        with pytest.raises(native.ZkhipError) as e:
            multi.load_proving_key(cid, raw17[:-3], scheme="gm17")
        assert e.value.code == -2 and "member" in str(e.value)
        with pytest.raises(native.ZkhipError):          # the failed load left no key behind
            multi.prove_gm17(z, d1, d2, r_)
    finally:
        multi.close()


@pytest.mark.parametrize("curve,ndev", [(BN254, 3), (BLS12_381, 2)], ids=lambda v: getattr(v, "name", str(v)))
def test_emu_multi_device(curve, ndev):
    _checks(emu_library(), [0] * ndev, curve, 5)


def test_emu_multi_more_members_than_points():
    _checks(emu_library(), [0] * 6, BN254, 2)


def test_multi_bad_arguments():
    lib = emu_library()
    with pytest.raises(native.ZkhipError):
        native.Multi([], lib)
    with pytest.raises(native.ZkhipError):
        native.Multi([0, 7], lib)            # the emulator has one device


@pytest.mark.gpu
def test_gpu_multi_device_two_members_one_gpu():
    lib = native.default_library()
    _checks(lib, [0, 0], BN254, 12)
    _checks(lib, [0, 0, 0], BLS12_381, 8, kind="sha")


@pytest.mark.gpu
def test_gpu_multi_device_all_gpus():
    """Every GPU of the box as one member each (one on a single-GPU box; the driver's 8-GPU node runs the real thing)."""
    lib = native.default_library()
    n = lib.device_count()
    _checks(lib, list(range(n)), BN254, 13)


@pytest.mark.gpu
def test_gpu_config3_2e22_eight_members():
    """BASELINE.json configs[2]: the synthetic 2^22-constraint BN254 circuit, ONE Groth16 proof across 8 members (1/8 of
    every base table each; they share the box's GPU(s)), against the C++ oracle's closed form (Fr arithmetic and three
    fixed-base multiplications at this size) and against the whole key on one context."""
    import os
    lg = int(os.environ.get("ZKHIP_TEST_CONFIG3_LOG", "22"))
    lib = native.default_library()
    ndev = lib.device_count()
    multi = native.Multi([k % ndev for k in range(8)], lib)
    try:
        circ = synth.circuit(0, lg, kind="dense")
        z = circ.assignment(0x5EED0022)
        ctx = multi.member_context(0)
        cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
        tox = synth.toxic_waste(0)
        raw = native.setup_g16(ctx, cs, tox)
        assert raw.size > (1 << lg) * 64 * 5
        multi.load_constraint_system(0, circ.n, circ.l, circ.w, circ.mats())
        multi.load_proving_key(0, raw)
        r, s = 0x123456789abcdef0123, 0xfedcba9876543210fed
        got, tm = multi.prove_g16(z, r, s, want_timings=True)
        oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
        tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
        assert got == cpu.trapdoor(oc, tb, z, r, s)
        multi.close()                                               # (frees the borrowed member context too)
        ctx = native.Context(0, lib)
        cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
        whole = native.ProvingKey(ctx, 0, raw)                      # 24 GiB of tables at 2^22
        assert native.prove_g16(ctx, whole, cs, z, r, s) == got
        ctx.close()
        print("config 3: 2^%d constraints, 8 members on %d GPU(s): %.1f ms per proof (slowest member phases: %s)" % (lg, ndev, tm["total_ms"], tm))
    finally:
        multi.close()


# ---------------------------------------------------------------- the exchange step over RCCL (zkhip_multi_use_rccl)
def _rccl_checks(lib, devices, curve, log_domain):
    """Same proofs with the members' shares exchanged on the device (all-gather of the raw bucket-set sums) as through host
    memory and as from one context; both schemes; switching back and forth."""
    cid = curve.curve_id
    circ = synth.circuit(cid, log_domain, kind="dense", seed=0xACC1 + log_domain)
    z = circ.assignment(0x5EED + log_domain)
    oc = cpu.Circuit.from_csr(cid, circ.n, circ.l, circ.w, circ.mats())
    multi = native.Multi(devices, lib)
    try:
        multi.load_constraint_system(cid, circ.n, circ.l, circ.w, circ.mats())
        tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
        raw = cpu.ProvingKey.setup(oc, tox).serialize()
        multi.load_proving_key(cid, raw)
        r, s = 0x1234567 % curve.r, 0x7654321 % curve.r
        want = cpu.trapdoor(oc, tox, z, r, s)
        assert "host memory" in multi.exchange()
        assert multi.prove_g16(z, r, s) == want
        multi.use_rccl(True)
        desc = multi.exchange()
        assert ("RCCL" in desc or "emulated all-gather" in desc) and "host memory" not in desc
        for rep in range(3):
            got, tm = multi.prove_g16(z, r + rep, s, want_timings=True)
            assert got == cpu.trapdoor(oc, tox, z, r + rep, s) and tm["total_ms"] > 0
        with pytest.raises(native.ZkhipError):          # a bad assignment fails on every member BEFORE anyone enters the collective
            bad = z.copy(); bad[0] = 2
            multi.prove_g16(bad, r, s)
        assert multi.prove_g16(z, r, s) == want          # ... and the group is usable afterwards
        tb17 = cpu.gm17_toxic_bytes(gm17.Toxic.from_seed(curve))
        raw17 = cpu.Gm17ProvingKey.setup(oc, tb17).serialize()
        multi.load_proving_key(cid, raw17, scheme="gm17")
        assert multi.prove_gm17(z, 5, 6, 7) == cpu.gm17_trapdoor(oc, tb17, z, 5, 7)
        multi.use_rccl(False)
        assert "host memory" in multi.exchange()
        assert multi.prove_gm17(z, 5, 6, 7) == cpu.gm17_trapdoor(oc, tb17, z, 5, 7)
        return desc
    finally:
        multi.close()


@pytest.mark.parametrize("curve,ndev", [(BN254, 3), (BLS12_381, 2)], ids=lambda v: getattr(v, "name", str(v)))
def test_emu_multi_device_exchange_on_the_device(curve, ndev):
    _rccl_checks(emu_library(), [0] * ndev, curve, 5)


def test_libzkhip_binds_rccl_at_run_time_only():
    """libzkhip.so must not depend on librccl at load time (the single-GPU prover and the CLI never need it); the symbols it
    resolves with dlopen must exist in the RCCL of this image."""
    import ctypes
    import os
    import subprocess
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zokrates_amd", "libzkhip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "librccl" not in needed and "libamdhip64" in needed
    rccl = "/opt/rocm/lib/librccl.so.1"
    if not os.path.exists(rccl):
        pytest.skip("no RCCL in this image")
    syms = subprocess.run(["nm", "-D", "--defined-only", rccl], capture_output=True, text=True).stdout
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclAllGather", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString", "ncclGetVersion"):
        assert (" " + name + "\n") in syms or (" " + name + "@") in syms, name


@pytest.mark.gpu
def test_gpu_rccl_exchange_every_gpu_one_rank_each():
    """One member per GPU of the box with the real RCCL all-gather (a single-GPU box runs it with one rank: communicator,
    collective call, gather buffers and the combine are the same code; the driver's 8-GPU node runs eight)."""
    lib = native.default_library()
    n = lib.device_count()
    desc = _rccl_checks(lib, list(range(n)), BN254, 12)
    assert "RCCL" in desc and ("%d rank(s)" % n) in desc, desc


@pytest.mark.gpu
def test_gpu_rccl_refuses_two_ranks_on_one_gpu():
    lib = native.default_library()
    multi = native.Multi([0, 0], lib)
    try:
        with pytest.raises(native.ZkhipError) as e:
            multi.use_rccl(True)
        assert e.value.code == -1 and "one device per member" in str(e.value)
        assert "host memory" in multi.exchange()
    finally:
        multi.close()


@pytest.mark.gpu
def test_gpu_context_churn_with_host_transfers():
    """Contexts created and destroyed in turn, each moving caller-owned host memory both ways (the library's pinned staging
    ring outlives a context: it must not hold events of streams that are gone), alone and with two contexts alive at once."""
    lib = native.default_library()
    circ = synth.circuit(0, 11, kind="dense", seed=0xC0DE)
    z = circ.assignment(5)
    oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
    tox = cpu.toxic_bytes(g16.Toxic.from_seed(BN254))
    raw = cpu.ProvingKey.setup(oc, tox).serialize()
    want = cpu.trapdoor(oc, tox, z, 3, 4)
    keep = None
    for rep in range(8):
        ctx = native.Context(0, lib)
        cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
        pk = native.ProvingKey(ctx, 0, raw)
        assert native.prove_g16(ctx, pk, cs, z, 3, 4) == want, rep
        assert cs.witness_map(z).size == 32 * 2048
        if rep % 3 == 0 and keep is None:
            keep = (ctx, cs, pk)                       # stays alive across the next iterations
            continue
        pk.close(); cs.close(); ctx.close()
        if keep is not None and rep % 3 == 2:
            kctx, kcs, kpk = keep
            assert native.prove_g16(kctx, kpk, kcs, z, 3, 4) == want
            kpk.close(); kcs.close(); kctx.close()
            keep = None
