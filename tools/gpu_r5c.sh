#!/bin/bash
# round 5, session C: the whole -m gpu suite on the new default build, the driver's bench command (with the sampler's view of the
# chip under load), kernel traces (pipelined + one stream) of the same.   bash tools/gpu_r5c.sh [tag]
set -u
tag=${1:-r5c}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
ls /sys/bus/pci/devices/*/hwmon/hwmon*/ 2>/dev/null | head -60 > "$out/hwmon_ls.txt"
timeout 1500 python -m pytest tests/ -x -q -m gpu > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -6 "$out/pytest_gpu.log"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"
python - "$out/bench_driver_command.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('driver command:', round(d['value'],2), 'proofs/s', round(d['ms_per_step'],3), 'ms/step; repeats', [round(x,3) for x in d['repeats']['ms_per_step']], 'single', round(d['single_proof_ms'],2))
print('under_load', json.dumps(d.get('under_load'))[:900])
print('serial', json.dumps(d.get('phases_ms_serial'))[:400])
print('cpu', json.dumps(d.get('cpu_baseline'))[:300])
e=d.get('cli_end_to_end_ms') or {}
print('cli', {k:(round(v.get('process_wall_ms',0)), round(v.get('total_in_process_ms',0)) if 'total_in_process_ms' in v else None) for k,v in e.items() if isinstance(v,dict)})
PY
SKIP_PMC=1 PROF_TIMEOUT=300 bash tools/profile_round.sh $tag > "$out/profile_round.log" 2>&1; tail -25 "$out/profile_round.log"
