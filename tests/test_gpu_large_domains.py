"""`-m gpu`: domains above 2^22.  The reference has no size limit below the scalar field's two-adicity — `Groth16::prove` /
`GM17::prove` (/root/reference/zokrates_ark/src/groth16.rs:44, gm17.rs:63) take whatever radix-2 domain the circuit needs (2^28
on bn128, 2^32 on bls12_381) — and until round 4 this library stopped at 2^22 (two LDS-resident sub-transforms of <= 2^11
points).  Larger domains take THREE passes (N = N1 * N2 * N3, kernels_ntt.cuh) and leave the quotient in a three-digit sigma
order the key's h bases are permuted into; keys whose window-multiple tables would not fit the device keep every 2nd / 4th ...
multiple (MsmShape::sets).  Checked here at the sizes VERDICT r3 item 2 names: the transform itself against the oracle's radix-2
NTT at 2^23, a Groth16 proof of the literal n = 2^22 circuit (domain 2^23) and a GM17 proof of n = 2^22 - 2 (SAP domain 2^23)
against the C++ oracle's closed-form trapdoor proofs."""
import os
import time

import numpy as np
import pytest

from oracle import cpu
from zokrates_amd import native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    d = c.describe()
    assert "EMULATOR" not in d and "gfx950" in d, d
    yield c
    c.close()


def _rand_fr(n, seed):
    rnd = np.random.default_rng(seed)
    a = rnd.integers(0, 256, size=n * 32, dtype=np.uint8)
    a.reshape(-1, 32)[:, 31] &= 0x0f        # < r on both curves
    return a


@pytest.mark.parametrize("curve_id,logn", [(0, 23), (1, 23), (0, 24)])
def test_ntt_three_passes(ctx, curve_id, logn):
    a = _rand_fr(1 << logn, logn + curve_id)
    for d in ("fft", "coset_ifft") if logn > 23 else ("fft", "ifft", "coset_fft", "coset_ifft"):
        assert ctx.ntt(curve_id, a, d).tobytes() == cpu.ntt(curve_id, a, d).tobytes(), d
    assert ctx.ntt(curve_id, ctx.ntt(curve_id, a, "coset_fft"), "coset_ifft").tobytes() == a.tobytes()


def test_groth16_literal_2e22_constraints(ctx):
    """n = 2^22 constraints exactly (BASELINE.json configs[2] read literally): n + l = 2^22 + 2 -> domain 2^23, three NTT passes."""
    circ = synth.circuit(0, n=1 << 22)
    assert circ.N == 1 << 23
    cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(0)
    raw = native.setup_g16(ctx, cs, tox)
    pk = native.ProvingKey(ctx, 0, raw)
    assert pk.hlen == (1 << 23) - 1
    del raw
    z = circ.assignment(0x5EED0001)
    oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
    tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
    r_, s_ = 0xDEADBEEF12345678, 0xCAFEBABE87654321
    want = cpu.trapdoor(oc, tb, z, r_, s_)
    assert native.prove_g16(ctx, pk, cs, z, r_, s_) == want
    # the pipelined batch (three proofs in flight) over the same domain, and thinned tables (every 4th window multiple)
    za = native.Assignment(ctx, cs, z)
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za] * 3, [(r_, s_), (5, 6), (r_, s_)])
    assert proofs[0] == want and proofs[2] == want and proofs[1] == cpu.trapdoor(oc, tb, z, 5, 6)
    pk.close()
    ctx.tune("msm_sets", 4)
    try:
        pk4 = native.ProvingKey.from_image(ctx, 0, _image_with_sets(ctx, cs, tox))
        assert native.prove_g16_resident(ctx, pk4, cs, za, r_, s_) == want
        pk4.close()
    finally:
        ctx.tune("msm_sets", 0)


def _image_with_sets(ctx, cs, tox):
    """A key built under the context's current MSM_SETS, as its image (the import must honour the shape the image names)."""
    pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, tox))
    img = pk.export_image()
    pk.close()
    return img


def test_gm17_2e22_constraints(ctx):
    """GM17 over n = 2^22 - 2 constraints: the SAP has 2n + 2(l - 1) + 1 rows -> domain 2^23, ~2^23 variables."""
    circ = synth.circuit(0, 22)
    cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(0)
    t4 = (tox[0], tox[1], tox[2], tox[4])
    raw = native.setup_gm17(ctx, cs, t4)
    pk = native.ProvingKey(ctx, 0, raw, scheme="gm17")
    assert pk.hlen - 1 == 1 << 23 and pk.m == 1 + 2 + circ.w + circ.n
    del raw
    z = circ.assignment(0x5EED0001)
    oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
    tb17 = b"".join(int(v).to_bytes(32, "little") for v in t4)
    d1, d2, r_ = 0x123456789abcdef0123456789, 0xfedcba9876543210, 0x1111222233334444555566667777
    assert native.prove_gm17(ctx, pk, cs, z, d1, d2, r_) == cpu.gm17_trapdoor(oc, tb17, z, d1, r_)
    pk.close()
