"""The two binary key parsers behind the C ABI take bytes from files (`proving.key`, a cached key image): whatever those
bytes are, the call returns a handle or an error code — it never aborts, never reads out of bounds, never allocates by a
length field it has not checked.  Emulator build (same host code as the product; reference for the formats:
/root/reference/zokrates_ark/src/groth16.rs:40-42 `deserialize_unchecked`)."""
import random
import struct

import numpy as np
import pytest

from oracle import cpu
from oracle import gm17
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

from emu_util import emu_library


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0, emu_library())
    yield c
    c.close()


def _keys(curve):
    oc = cpu.Circuit.synth(curve.curve_id, 14, 0xF0221)
    raw = cpu.ProvingKey.setup(oc, cpu.toxic_bytes(g16.Toxic.from_seed(curve))).serialize().tobytes()
    graw = cpu.Gm17ProvingKey.setup(oc, cpu.gm17_toxic_bytes(gm17.Toxic.from_seed(curve))).serialize().tobytes()
    return oc, raw, graw


def _mutations(good, rnd, count, hot=()):
    """byte flips, truncations, insertions, and 64-bit length fields overwritten with large / overflowing values"""
    for _ in range(count):
        b = bytearray(good)
        k = rnd.randrange(5)
        if k == 0:
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif k == 1:
            b = b[:rnd.randrange(len(b))]
        elif k == 2:
            i = rnd.randrange(len(b))
            b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
        elif k == 3 and hot:
            off = rnd.choice(hot)
            b[off:off + 8] = struct.pack("<Q", rnd.choice([0, 1, (1 << 32) - 1, 1 << 32, (1 << 61) + 5, (1 << 64) - 1, rnd.randrange(1 << 40)]))
        else:
            off = rnd.randrange(0, max(1, len(b) - 8))
            b[off:off + 8] = struct.pack("<Q", rnd.choice([(1 << 64) - 1, 1 << 63, rnd.randrange(1 << 64)]))
        yield bytes(b)


@pytest.mark.parametrize("curve", [BN254, BLS12_381], ids=lambda c: c.name)
def test_proving_key_loader_survives_mutations(ctx, curve):
    oc, raw, graw = _keys(curve)
    rnd = random.Random(7)
    nb = curve.fq_bytes
    # offsets of the Vec length fields of the Groth16 key (vk: alpha, beta2, gamma2, delta2, then gamma_abc's length)
    vk_len = 2 * nb + 3 * 4 * nb
    hot = [vk_len]
    for scheme, good in (("g16", raw), ("gm17", graw)):
        loaded = rejected = 0
        for b in _mutations(good, rnd, 60 if curve is BN254 else 40, hot if scheme == "g16" else ()):
            try:
                pk = native.ProvingKey(ctx, curve.curve_id, np.frombuffer(b, dtype=np.uint8), scheme=scheme)
                loaded += 1          # flipped coordinate bytes still parse ("unchecked", like the reference)
                pk.close()
            except native.ZkhipError as e:
                assert e.code in (-1, -2, -3), e
                rejected += 1
        assert rejected > 15, (scheme, loaded, rejected)
    # the good bytes still load after all that (the context survived)
    native.ProvingKey(ctx, curve.curve_id, np.frombuffer(raw, dtype=np.uint8)).close()
    with pytest.raises(native.ZkhipError):
        native.ProvingKey(ctx, curve.curve_id, np.zeros(0, dtype=np.uint8))
    with pytest.raises(native.ZkhipError):          # a GM17 key is not a Groth16 key
        native.ProvingKey(ctx, curve.curve_id, np.frombuffer(graw, dtype=np.uint8))


def test_key_image_importer_survives_mutations(ctx):
    curve = BN254
    oc, raw, _ = _keys(curve)
    pk = native.ProvingKey(ctx, curve.curve_id, np.frombuffer(raw, dtype=np.uint8))
    img = pk.export_image().tobytes()
    pk.close()
    cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)])
    z = oc.assignment()
    want = cpu.trapdoor(oc, cpu.toxic_bytes(g16.Toxic.from_seed(curve)), z, 5, 6)
    rnd = random.Random(11)
    imported = rejected = 0
    hot = list(range(0, min(len(img), 512), 8))      # the header: magic, sizes, counts
    for b in _mutations(img, rnd, 80, hot):
        try:
            p2 = native.ProvingKey.from_image(ctx, curve.curve_id, np.frombuffer(b, dtype=np.uint8))
            imported += 1
            try:
                native.prove_g16(ctx, p2, cs, z, 5, 6)       # a damaged table gives a wrong proof or an error, never a crash
            except native.ZkhipError:
                pass
            p2.close()
        except native.ZkhipError as e:
            assert e.code in (-1, -2, -3), e
            rejected += 1
    assert rejected > 25, (imported, rejected)
    p2 = native.ProvingKey.from_image(ctx, curve.curve_id, np.frombuffer(img, dtype=np.uint8))
    assert native.prove_g16(ctx, p2, cs, z, 5, 6) == want
    p2.close()
