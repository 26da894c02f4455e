#!/bin/bash
# round 6, session m: the fold's layouts (k_msm_fold_lines never / always / for one-table launches) and the column pass's shares, lone and pipelined, one process.
set -u
tag=${1:-r6m}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "A/B in one process: lone and pipelined, dense 2^20 bound"
python - > "$out/fold_ab.txt" 2>&1 <<'PY'
import json, sys, time
sys.path.insert(0, '.')
from zokrates_amd import native, synth
native.default_library().init(16)
ctx = native.Context(0)
circ = synth.circuit(0, 20)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, synth.toxic_waste(0)))
zas = [native.Assignment(ctx, cs, circ.assignment(7 + i)) for i in range(8)]
t0 = time.perf_counter(); pk.bind(cs); print(json.dumps({"bind_ms": round(1000 * (time.perf_counter() - t0), 1)}), flush=True)
ref = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13)
for rnd in range(3):
    for lines, hg in ((0, 32), (0, 128), (0, 256), (2, 32), (2, 256), (1, 32)):
        ctx.tune("fold_lines", lines); ctx.tune("fold_hg", hg)
        lone = []
        for i in range(7):
            p, tm = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13, want_timings=True)
            assert p == ref
            lone.append(tm["total_ms"])
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 8] for i in range(6)], [(100 + i, 7) for i in range(6)])
        t0 = time.perf_counter()
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 8] for i in range(32)], [(100 + i, 7) for i in range(32)])
        dt = time.perf_counter() - t0
        print(json.dumps({"round": rnd, "fold_lines": lines, "fold_hg": hg, "lone_ms": sorted(round(t, 3) for t in lone[1:]), "batch_ms_per_proof": round(1000 * dt / 32, 3), "proofs_per_s": round(32 / dt, 2)}), flush=True)
PY
cat "$out/fold_ab.txt"
for mode in 0 2; do
step "kernel times, one stream, fold_lines $mode"
( cd /tmp && ZKHIP_FOLD_LINES=$mode ZKHIP_SERIAL=1 timeout 120 rocprofv3 --kernel-trace --stats -d "$out/prof_serial$mode" -o serial -- python "$root/tools/lone_proof_probe.py" 20 6 1 > "$out/serial$mode.log" 2>&1 )
db=$(find "$out/prof_serial$mode" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" "$out/${tag}_bound_serial_kernel_stats_fold_lines$mode.md" > /dev/null
grep -E "fold|heavy" "$out/${tag}_bound_serial_kernel_stats_fold_lines$mode.md" | cut -c1-200
done
find "$out" -name "*.db" -size +8M -delete
step "done"
