#!/bin/bash
# kernel trace of a short bench run and the per-stream listing of one isolated single proof (tools/gantt.py).
# Usage: bash tools/gpu_gantt.sh <tag> [extra bench args]
set -u
tag=${1:-gantt}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; tail -5 "$out/smoke.log"; exit 0; }
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$out/prof" -o t -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --warmup 2 --serial-proofs 0 "$@" > "$out/prof.log" 2>&1
db=$(find "$out/prof" -name "*.db" | head -1)
grep -h '^{"metric"' "$out/prof.log" > "$out/bench_under_rocprof.json"
for w in -6 -5 -4; do python "$root/tools/gantt.py" "$db" $w > "$out/single_proof_gantt_$w.txt" 2>&1; done
for w in -3 -1; do python "$root/tools/gantt.py" "$db" $w > "$out/single_proof_from_host_gantt_$w.txt" 2>&1; done
find "$out" -name "*.db" -size +8M -delete
head -3 "$out/single_proof_gantt_-4.txt"; tail -2 "$out/single_proof_gantt_-4.txt"
python - "$out/bench_under_rocprof.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(round(d['value'],2),'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d['single_proof_from_host_ms'],2),'from host')
PY
