#!/usr/bin/env python3
"""Lone proofs of one workload in ONE process: min / quartiles / median of `count` isolated proofs over a bound key (a lone proof's latency moves by
+-0.4 ms from proof to proof: bench.py's min-of-three cannot carry an A/B).  Settings come from the environment (ZKHIP_*), so an A/B is one process
per setting, alternating.  usage: lone_stats.py <dense|poseidon|sha256> [count]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zokrates_amd import native, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
native.default_library().init(16)
ctx = native.Context(0)
if os.environ.get("ZKHIP_PIPES", "1") not in ("0", "-"):
    ctx.tune("pipe_plan", 1)
if kind == "poseidon":
    from zokrates_amd import poseidon
    curve_id, circ = 1, poseidon.chain(1, 1024)
elif kind == "sha256":
    from zokrates_amd import sha256_circuit as sha
    curve_id = 0
    circ = sha.circuit(0, max(1, (1 << 20) // (len(sha.template()[0]) + 7)))
else:
    curve_id, circ = 0, synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, curve_id, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, curve_id, native.setup_g16(ctx, cs, synth.toxic_waste(curve_id)))
zhost = [circ.assignment(7 + i) for i in range(4)]
zas = [native.Assignment(ctx, cs, z) for z in zhost]
from_host = bool(os.environ.get("FROM_HOST"))      # FROM_HOST=1: the assignment comes from host memory in every proof (wall clock of the call)
pk.bind(cs)
ref = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13)
# PRE_BATCH=n: a pipelined batch of n proofs first (the state bench.py's lone proofs find the chip in: straight after its timed regions); PAUSE_S: idle after it
import time
if int(os.environ.get("PRE_BATCH", "0")):
    nb = int(os.environ["PRE_BATCH"])
    for _ in range(3):
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 4] for i in range(nb)], [(100 + i, 7) for i in range(nb)])
    time.sleep(float(os.environ.get("PAUSE_S", "0")))
lone = []
for i in range(count + 2):
    if from_host:
        t0 = time.perf_counter()
        p = native.prove_g16(ctx, pk, cs, zhost[0], 11, 13)
        ms = 1000.0 * (time.perf_counter() - t0)
    else:
        p, tm = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13, want_timings=True)
        ms = tm["total_ms"]
    assert p == ref
    if i >= 2:
        lone.append(ms)
lone.sort()
q = lambda f: round(lone[min(len(lone) - 1, int(f * len(lone)))], 3)
print(json.dumps({"kind": kind, "env": {k: v for k, v in os.environ.items() if k.startswith("ZKHIP_") and k != "ZKHIP_BENCH_CHILD"}, "from_host": from_host, "pre_batch": os.environ.get("PRE_BATCH"), "pause_s": os.environ.get("PAUSE_S"), "count": len(lone),
                  "min": q(0), "p25": q(0.25), "median": q(0.5), "p75": q(0.75), "max": q(1.0)}), flush=True)
