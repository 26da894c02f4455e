#!/bin/bash
# round 6, session l: the fold in one launch per MSM (k_msm_fold_lines), the streaming group addition (xyzz_add_from), the register-resident bind: parity, kernel times, lone proof and pipelined A/B in one process.
set -u
tag=${1:-r6l}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "MSM / proof parity"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gm17.py tests/test_random_circuits.py tests/test_poseidon.py tests/test_sha256_circuit.py tests/test_gpu_bound.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_sort.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_sort.log"; tail -3 "$out/pytest_sort.log"
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
    for two in (0, 1):
        ctx.tune("fold_lines", two)
        lone = []
        for i in range(7):
            p, tm = native.prove_g16_resident(ctx, pk, cs, zas[0], 11, 13, want_timings=True)
            assert p == ref
            lone.append(tm["total_ms"])
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 8] for i in range(6)], [(100 + i, 7) for i in range(6)])
        t0 = time.perf_counter()
        native.prove_g16_resident_batch(ctx, pk, cs, [zas[i % 8] for i in range(32)], [(100 + i, 7) for i in range(32)])
        dt = time.perf_counter() - t0
        print(json.dumps({"round": rnd, "fold_lines": two, "lone_ms": sorted(round(t, 3) for t in lone[1:]), "batch_ms_per_proof": round(1000 * dt / 32, 3), "proofs_per_s": round(32 / dt, 2)}), flush=True)
PY
cat "$out/fold_ab.txt"
step "kernel times, one stream"
( cd /tmp && ZKHIP_SERIAL=1 timeout 120 rocprofv3 --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/tools/lone_proof_probe.py" 20 6 1 > "$out/serial.log" 2>&1 )
db=$(find "$out/prof_serial" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" "$out/${tag}_bound_serial_kernel_stats.md" > /dev/null
grep -E "k_msm_place|k_msm_part|k_msm_count|k_msm_digits|k_msm_tile|k_ntt|fold|k_bind_fft|k_bind_l|k_bind_scale" "$out/${tag}_bound_serial_kernel_stats.md" | cut -c1-200
step "gantt of a lone proof"
( cd /tmp && timeout 120 rocprofv3 --kernel-trace -d "$out/prof_lone" -o lone -- python "$root/tools/lone_proof_probe.py" 20 8 1 > "$out/lone.log" 2>&1 )
tail -1 "$out/lone.log" | cut -c1-300
db=$(find "$out/prof_lone" -name "*.db" | head -1)
[ -n "$db" ] && python tools/gantt.py "$db" -2 > "$out/${tag}_lone_bound_proof_gantt.txt" 2>&1
find "$out" -name "*.db" -size +8M -delete
step "done"
