#!/usr/bin/env python3
"""Development probe: BASELINE.json configs[2] on ONE GPU — 2^22 constraints, BN254: whole-key proof, and the same proof
sharded over 8 virtual ranks (one after the other on this device), compared bit for bit and against the trapdoor oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zokrates_amd import native, synth
from oracle import cpu

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = native.Context(0)
t = time.time(); circ = synth.circuit(0, lg); print(f"circuit 2^{lg}: {time.time()-t:.1f}s", flush=True)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
tox = synth.toxic_waste(0)
t = time.time(); raw = native.setup_g16(ctx, cs, tox); print(f"gpu setup: {time.time()-t:.2f}s, key {raw.size/2**20:.0f} MiB", flush=True)
t = time.time(); pk = native.ProvingKey(ctx, 0, raw); print(f"pk load: {time.time()-t:.2f}s", flush=True)
t = time.time(); z = circ.assignment(0x5EED2222); print(f"assignment: {time.time()-t:.1f}s", flush=True)
za = native.Assignment(ctx, cs, z)
r_, s_ = 0x1122334455667788, 0x99aabbccddeeff00
for i in range(3):
    got, tm = native.prove_g16_resident(ctx, pk, cs, za, r_, s_, want_timings=True)
    print(f"whole-key proof: {tm['total_ms']:.1f} ms  accum g1 {tm['kernel_msm_accum_g1_ms']:.1f} g2 {tm['kernel_msm_accum_g2_ms']:.1f} ntt {tm['ntt_ms']:.1f}", flush=True)
proofs, tmb = native.prove_g16_resident_batch(ctx, pk, cs, [za] * 6, [(r_, s_)] * 6)
print(f"pipelined: {tmb['total_ms']/6:.1f} ms per proof", flush=True)
oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
t = time.time(); want = cpu.trapdoor(oc, tb, z, r_, s_); print(f"trapdoor oracle: {time.time()-t:.1f}s; parity {'OK' if want == got else 'FAIL'}", flush=True)
assert want == got and all(p == got for p in proofs)
pk.close()
parts = []
tt = []
for k in range(world):
    sh = native.ProvingKey(ctx, 0, raw, rank=k, world=world)
    native.prove_g16_partial(ctx, sh, cs, za, r_, s_)
    part, tm = native.prove_g16_partial(ctx, sh, cs, za, r_, s_, want_timings=True)
    parts.append(part); tt.append(tm["total_ms"])
    last = sh
print(f"sharded over {world} virtual ranks: per-rank partial {min(tt):.1f}..{max(tt):.1f} ms; combine parity",
      "OK" if native.combine_g16(ctx, last, parts, r_, s_) == got else "FAIL", flush=True)
