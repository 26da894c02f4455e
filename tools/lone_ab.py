#!/usr/bin/env python3
"""Lone proofs and a pipelined batch of the 2^20 BN254 circuit over a bound key, one tunable alternating in ONE process.
usage: lone_ab.py <hw_queues> <tunable> <value> [<value> ...]   (env: ZKHIP_* as the library reads them; ROUNDS=3)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zokrates_amd import native, synth  # noqa: E402

hwq, name, values = int(sys.argv[1]), sys.argv[2], [int(v) for v in sys.argv[3:]]
native.default_library().init(hwq)
ctx = native.Context(0)
circ = synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, synth.toxic_waste(0)))
zas = [native.Assignment(ctx, cs, circ.assignment(7 + i)) for i in range(8)]
pk.bind(cs)
ref = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13)
tag = {"hw_queues": hwq, "env": {k: v for k, v in os.environ.items() if k.startswith("ZKHIP_") and k != "ZKHIP_BENCH_CHILD"}}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for v in values:
        if name != "none":
            ctx.tune(name, v)
        lone = []
        for i in range(8):
            p, tm = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13, want_timings=True)
            assert p == ref
            lone.append(tm["total_ms"])
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 8] for i in range(6)], [(100 + i, 7) for i in range(6)])
        t0 = time.perf_counter()
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 8] for i in range(32)], [(100 + i, 7) for i in range(32)])
        dt = time.perf_counter() - t0
        row = dict(tag)
        row.update({"round": rnd, name: v, "lone_ms": sorted(round(t, 3) for t in lone[1:]), "proofs_per_s": round(32 / dt, 2)})
        print(json.dumps(row), flush=True)
