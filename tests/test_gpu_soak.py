"""`-m gpu` soak: a long-lived prover (`zokrates_js`' call pattern: one program, one key, many witnesses — SURVEY.md §8a, the resident
entry points of include/zkhip.h) is asked for hundreds of proofs of BASELINE.json configs[1] (2^20 constraints, BN254) in calls of every
shape — lone proofs, short and long pipelined batches, resident and host assignments, the key bound and as loaded, in a seeded random
order — and EVERY proof must be the bytes the oracle's closed form gives for its (witness, r, s).  What the parity tests cannot see: a
slot, a stream or a workspace that carries something over from the proof before it.

ZKHIP_SOAK_PROOFS (default 400: ~6 s of device time; the expected proofs cost the CPU oracle ~1 s each) and ZKHIP_SOAK_LOG (20) scale it:
`ZKHIP_SOAK_PROOFS=5000 python -m pytest tests/test_gpu_soak.py -m gpu` is the long run (profiles/r7l_soak.txt)."""
import os
import random

import numpy as np
import pytest

from oracle import cpu
from zokrates_amd import native


@pytest.mark.gpu
def test_soak_every_proof_matches_the_oracle():
    ctx = native.Context(0)            # raises if libzkhip.so or the GPU is missing: no fallback
    d = ctx.describe()
    assert "EMULATOR" not in d and "gfx950" in d, d
    run_soak(ctx, int(os.environ.get("ZKHIP_SOAK_LOG", "20")), int(os.environ.get("ZKHIP_SOAK_PROOFS", "400")))


def test_soak_on_the_emulator():
    """the same call mix over the kernel sources on the fibre emulator (tests/_emu: test-only), a 2^6 circuit"""
    import emu_util
    run_soak(native.Context(0, library=emu_util.emu_library()), 6, 48)


def run_soak(ctx, lg, total):
    from zokrates_amd import synth
    try:
        circ = synth.circuit(0, lg)
        cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
        tox = synth.toxic_waste(0, 0x50A6)
        pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, tox))
        oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
        tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
        rnd = random.Random(0x50A6)
        r_mod = (1 << 253) - 1
        # twelve (witness, r, s) triples over four witnesses — edge values of r and s among them — and what the oracle says they prove to
        zs = [circ.assignment(0x5EED1000 + k) for k in range(4)]
        za = [native.Assignment(ctx, cs, z) for z in zs]
        triples = [(k % 4, rnd.randrange(r_mod), rnd.randrange(r_mod)) for k in range(9)] + [(0, 0, 0), (1, 1, 0), (2, 0, 1)]
        want = [cpu.trapdoor(oc, tb, zs[w], r, s) for w, r, s in triples]
        assert len(set(want)) == len(want)
        done = 0
        calls = {"lone_resident": 0, "lone_host": 0, "batch_resident": 0, "batch_host": 0, "bind": 0, "unbind": 0}
        bound = False
        while done < total:
            kind = rnd.random()
            if kind < 0.06:                                   # the key changes state between calls, never inside one
                if bound:
                    pk.unbind()
                    calls["unbind"] += 1
                else:
                    pk.bind(cs)
                    calls["bind"] += 1
                bound = not bound
                continue
            if kind < 0.30:
                i = rnd.randrange(len(triples))
                w, r, s = triples[i]
                if rnd.random() < 0.5:
                    got = native.prove_g16_resident(ctx, pk, cs, za[w], r, s)
                    calls["lone_resident"] += 1
                else:
                    got = native.prove_g16(ctx, pk, cs, zs[w], r, s)
                    calls["lone_host"] += 1
                assert got == want[i], ("lone", done, i, bound)
                done += 1
                continue
            count = rnd.choice([2, 3, 4, 7, 16, 33])
            idx = [rnd.randrange(len(triples)) for _ in range(count)]
            if kind < 0.85:
                proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za[triples[i][0]] for i in idx], [triples[i][1:] for i in idx])
                calls["batch_resident"] += 1
            else:
                idx = idx[:4]                                 # (host batches copy count x 32 MiB)
                proofs, _ = native.prove_g16_batch(ctx, pk, cs, np.concatenate([zs[triples[i][0]] for i in idx]), [triples[i][1:] for i in idx])
                calls["batch_host"] += 1
            for j, i in enumerate(idx):
                assert proofs[j] == want[i], ("batch", done, j, i, bound, len(idx))
            done += len(idx)
        print("soak: %d proofs identical to the oracle's closed form, calls %r" % (done, calls))
    finally:
        ctx.close()
