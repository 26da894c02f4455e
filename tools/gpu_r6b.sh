#!/bin/bash
# round 6, session b: the new binding code (GM17, shards, images, 4-bit windows) and the lone-proof layouts on the device.
set -u
tag=${1:-r6b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "GPU tests of the binding and the lone layouts"
timeout 600 python -m pytest tests/test_gpu_bound.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_bound.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_bound.log"; tail -4 "$out/pytest_bound.log"
step "lone proof latency per layout (2^20, bound), bind time"
python - > "$out/lone_layouts.txt" 2>&1 <<'PY'
import json, sys, time
sys.path.insert(0, '.')
from zokrates_amd import native, synth
native.default_library().init(16)
ctx = native.Context(0)
circ = synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, synth.toxic_waste(0)))
za = native.Assignment(ctx, cs, circ.assignment(7))
t0 = time.time(); pk.bind(cs); print("bind_ms", round(1000 * (time.time() - t0), 1), flush=True)
ref = native.prove_g16_resident(ctx, pk, cs, za, 11, 13)
for rnd in range(3):
    for sched in (0, 1, 2, 3):
        ctx.tune("lone_sched", sched)
        tms = []
        for i in range(7):
            p, tm = native.prove_g16_resident(ctx, pk, cs, za, 11, 13, want_timings=True)
            assert p == ref
            tms.append(tm)
        best = min(tms[1:], key=lambda t: t["total_ms"])
        print(json.dumps({"round": rnd, "lone_sched": sched, "total_ms": sorted(round(t["total_ms"], 3) for t in tms[1:]),
                          "best": {k: round(v, 3) for k, v in best.items() if k in ("ntt_ms", "msm_h_ms", "msm_z_ms", "finish_ms", "kernel_msm_accum_g1_ms", "kernel_msm_accum_g2_ms", "kernel_ntt_ms")}}), flush=True)
ctx.tune("lone_sched", 0)
t0 = time.time()
native.prove_g16_resident_batch(ctx, pk, cs, [za] * 40, [(100 + i, 7) for i in range(40)])
print("batch_ms_per_proof", round(1000 * (time.time() - t0) / 40, 3))
PY
cat "$out/lone_layouts.txt"
step "gantt of a lone proof, layout 3"
( cd /tmp && ZKHIP_LONE_SCHED=3 timeout 180 rocprofv3 --kernel-trace -d "$out/prof_lone3" -o lone -- python "$root/tools/lone_proof_probe.py" 20 8 1 > "$out/lone3.log" 2>&1 )
tail -1 "$out/lone3.log"
db=$(find "$out/prof_lone3" -name "*.db" | head -1)
[ -n "$db" ] && python tools/gantt.py "$db" -2 > "$out/${tag}_lone_bound_proof_gantt_layout3.txt" 2>&1
step "GM17 2^20: as loaded and bound"
python - > "$out/gm17_bound.txt" 2>&1 <<'PY'
import json, sys, time
sys.path.insert(0, '.')
from zokrates_amd import native, synth
ctx = native.Context(0)
circ = synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
tox = synth.toxic_waste(0)
pk = native.ProvingKey(ctx, 0, native.setup_gm17(ctx, cs, (tox[0], tox[1], tox[2], tox[4])), scheme="gm17")
zas = [native.Assignment(ctx, cs, circ.assignment(7 + i)) for i in range(4)]
ref = native.prove_gm17(ctx, pk, cs, circ.assignment(7), 1, 2, 3)
for bound in (False, True):
    if bound:
        t0 = time.time(); pk.bind(cs); print("gm17 bind_ms", round(1000 * (time.time() - t0), 1))
        assert native.prove_gm17(ctx, pk, cs, circ.assignment(7), 1, 2, 3) == ref
    native.prove_gm17_resident_batch(ctx, pk, cs, [zas[i % 4] for i in range(6)], [(i, 2, 3) for i in range(6)])
    for rep in range(3):
        t0 = time.time()
        native.prove_gm17_resident_batch(ctx, pk, cs, [zas[i % 4] for i in range(24)], [(i, 2, 3) for i in range(24)])
        dt = time.time() - t0
        print(json.dumps({"bound": bound, "proofs_per_s": round(24 / dt, 2), "ms_per_proof": round(1000 * dt / 24, 3)}), flush=True)
PY
cat "$out/gm17_bound.txt"
find "$out" -name "*.db" -size +8M -delete
step "done"
