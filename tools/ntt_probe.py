#!/usr/bin/env python3
"""Time of the transform passes alone (one stream) under a few settings of one knob, alternating in ONE process:
usage: ntt_probe.py <tunable> v1 v2 ...   e.g.  ntt_probe.py ntt_skew_us 0 4 8 12 16
prints kernel_ntt_ms of lone serial proofs over a bound 2^20 key (8 pass-vectors + the pointwise kernel) per setting."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zokrates_amd import native, synth  # noqa: E402

knob, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
ctx = native.Context(0)
circ = synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, synth.toxic_waste(0)))
za = native.Assignment(ctx, cs, circ.assignment(7))
pk.bind(cs)
ref = native.prove_g16_resident(ctx, pk, cs, za, 11, 13)
ctx.tune("serial", 1)
for rnd in range(3):
    for v in values:
        ctx.tune(knob, v)
        tms = []
        for i in range(6):
            p, tm = native.prove_g16_resident(ctx, pk, cs, za, 11, 13, want_timings=True)
            assert p == ref
            tms.append(tm["kernel_ntt_ms"])
        best = min(tms[1:])
        print(json.dumps({"round": rnd, knob: v, "kernel_ntt_ms": round(best, 4), "us_per_pass_vector": round(1000 * (best - 0.025) / 8, 2), "all": [round(t, 3) for t in tms[1:]]}), flush=True)
