#!/usr/bin/env python3
"""Single-proof latency with the key bound / as loaded, alternating in ONE process (session r5k: the Poseidon chain on BLS12-381
measured 8.1 ms bound against 7.0 unbound in its bench line, the 2^20 headline 10.4 against 10.3 — is that the binding or the
order of the legs?).  usage: gpu_bound_latency_ab.py [poseidon|dense]"""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from zokrates_amd import native, synth, poseidon  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "poseidon"
if kind == "head_start":
    # session r5n: ZKHIP_G2_HEAD_START = 0 / 1 / 2 (a lone proof's G1 lanes held for the G2 accumulation: never / over a bound key /
    # always), a context per setting, the Poseidon chain on BLS12-381: isolated proofs bound and as loaded, and a pipelined batch
    import os
    circ = poseidon.chain(1, 1024)
    tox = synth.toxic_waste(1)
    raw = None
    for rnd in range(2):
        for setting in ("0", "1", "2"):
            os.environ["ZKHIP_G2_HEAD_START"] = setting
            ctx = native.Context(0)
            cs = native.ConstraintSystem(ctx, 1, circ.n, circ.l, circ.w, circ.mats())
            if raw is None:
                raw = native.setup_g16(ctx, cs, tox)
            pk = native.ProvingKey(ctx, 1, raw)
            za = native.Assignment(ctx, cs, circ.assignment(7))
            row = {"round": rnd, "setting": int(setting)}
            for bound in (False, True):
                if bound:
                    pk.bind(cs)
                tms = [native.prove_g16_resident(ctx, pk, cs, za, 11 + i, 13, want_timings=True)[1] for i in range(9)][1:]
                best = min(tms, key=lambda t: t["total_ms"])
                row["bound" if bound else "as_loaded"] = {"total_ms": sorted(round(t["total_ms"], 3) for t in tms), "g2_acc_ms": round(best["kernel_msm_accum_g2_ms"], 3),
                                                          "msm_z_ms": round(best["msm_z_ms"], 3), "ntt_ms": round(best["ntt_ms"], 3)}
            t0 = time.time()
            native.prove_g16_resident_batch(ctx, pk, cs, [za] * 32, [(100 + i, 7) for i in range(32)])
            row["bound_batch_ms_per_proof"] = round(1000 * (time.time() - t0) / 32, 3)
            print(json.dumps(row), flush=True)
            pk.close(); cs.close() if hasattr(cs, "close") else None; ctx.close()
    sys.exit(0)
ctx = native.Context(0)
if kind == "poseidon":
    curve_id, circ = 1, poseidon.chain(1, 1024)
else:
    curve_id, circ = 0, synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, curve_id, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, curve_id, native.setup_g16(ctx, cs, synth.toxic_waste(curve_id)))
za = native.Assignment(ctx, cs, circ.assignment(7))
rows = []
for rnd in range(4):
    for bound in (True, False):
        if bound:
            t0 = time.time(); pk.bind(cs); bind_ms = 1000 * (time.time() - t0)
        else:
            pk.unbind(); bind_ms = None
        tms = [native.prove_g16_resident(ctx, pk, cs, za, 11 + i, 13, want_timings=True)[1] for i in range(6)][1:]
        best = min(tms, key=lambda t: t["total_ms"])
        rows.append({"round": rnd, "bound": bound, "bind_ms": bind_ms, "total_ms": [round(t["total_ms"], 3) for t in tms],
                     "best": {k: round(v, 3) for k, v in best.items()}})
        print(json.dumps(rows[-1]), flush=True)
