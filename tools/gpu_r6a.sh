#!/bin/bash
# round 6, session a: where round 5's build stands on today's box — the driver's line, the gantt of a LONE proof over a bound key
# (VERDICT r5 item 2), and the FETCH_SIZE calibration of the accumulation's gathers (item 1b).
set -u
tag=${1:-r6a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "driver flags"
timeout 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --e2e 0 > "$out/bench_driver_flags.json" 2> "$out/bench.err"
python - "$out/bench_driver_flags.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); b=d['bound_key']
        print('  headline', round(d['value'],2), 'proofs/s', [round(x,3) for x in d['repeats']['ms_per_step']], '| unbound', round(b.get('unbound_ms_per_step',0),3), '| single', round(d['single_proof_ms'],2), round(b.get('unbound_single_proof_ms',0),2))
        print('  serial', {k: round(v,3) for k,v in d['phases_ms_serial'].items()})
PY
step "lone proofs under the kernel trace"
( cd /tmp && timeout 180 rocprofv3 --kernel-trace -d "$out/prof_lone" -o lone -- python "$root/tools/lone_proof_probe.py" 20 8 1 > "$out/lone.log" 2>&1 )
tail -1 "$out/lone.log"
db=$(find "$out/prof_lone" -name "*.db" | head -1)
[ -n "$db" ] && python tools/gantt.py "$db" -2 > "$out/${tag}_lone_bound_proof_gantt.txt" 2>&1
head -5 "$out/${tag}_lone_bound_proof_gantt.txt"
step "FETCH_SIZE calibration"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/fetch_calib tools/fetch_calib.hip 2> "$out/fetch_calib_build.log"
( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/prof_calib" -o calib -- "$root/tools/fetch_calib" > "$out/fetch_calib.log" 2>&1 )
db=$(find "$out/prof_calib" -name "*.db" | head -1)
[ -n "$db" ] && python tools/pmc_stats.py "$db" "$out/${tag}_fetch_calibration.md"
grep -E "stream|gather" "$out/fetch_calib.log"
find "$out" -name "*.db" -size +8M -delete
step "done"
