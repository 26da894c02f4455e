#!/usr/bin/env python3
"""Lone proofs of the 2^20 BN254 circuit over a bound key, one after the other with the host idle in between — the input of
tools/gantt.py under `rocprofv3 --kernel-trace` (where does a single proof's wall-clock go once nothing else is in flight?).
usage: lone_proof_probe.py [log_domain=20] [proofs=8] [bound=1]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zokrates_amd import native, synth  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bound = int(sys.argv[3]) if len(sys.argv) > 3 else 1
native.default_library().init(int(os.environ.get("HWQ", "16")))      # (as bench.py: a process with one resident prover)
ctx = native.Context(0)
circ = synth.circuit(0, lg)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, synth.toxic_waste(0)))
za = native.Assignment(ctx, cs, circ.assignment(7))
if bound:
    pk.bind(cs)
rows = []
for i in range(count):
    t0 = time.perf_counter()
    _, tm = native.prove_g16_resident(ctx, pk, cs, za, 11 + i, 13, want_timings=True)
    tm["wall_ms"] = 1000 * (time.perf_counter() - t0)
    rows.append(tm)
    time.sleep(0.01)
best = min(rows[1:], key=lambda t: t["total_ms"])
print(json.dumps({"bound": bool(bound), "total_ms": [round(t["total_ms"], 3) for t in rows], "best": {k: round(v, 3) for k, v in best.items()}}))
