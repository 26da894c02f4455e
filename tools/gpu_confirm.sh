#!/bin/bash
# The driver's two commands on a cold box (suite first, then the bench with its flags), then — time permitting — the two HBM
# counter passes of the same build.  Usage: bash tools/gpu_confirm.sh <tag>
set -u
tag=${1:-confirm}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 400 python -m pytest tests/ -x -q -m gpu > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -3 "$out/pytest_gpu.log"
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"; echo "bench rc=$?"
python3 -c "
import json,sys
d=json.loads(open('$out/bench_driver_command.json').read().strip().splitlines()[-1])
print('value', d.get('value'), 'single', d.get('single_proof_ms'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('gpu_proof_identical'), 'attempts', d.get('attempts'))
e=d.get('cli_end_to_end_ms') or {}
print({k: round(v['process_wall_ms']) for k,v in e.items() if isinstance(v, dict) and 'process_wall_ms' in v})
" 2>&1 | tail -3
( cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ZKHIP_SERIAL=1 timeout ${PMC_TIMEOUT:-110} rocprofv3 --pmc $ctr --kernel-trace -d "$out/prof_pmc_$ctr" -o pmc -- python "$root/bench.py" --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_pmc_$ctr.log" 2>&1
    echo "pmc $ctr rc=$?"
    db=$(find "$out/prof_pmc_$ctr" -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && python "$root/tools/pmc_stats.py" "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null
  done
  f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then
    python "$root/tools/pmc_traffic.py" "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0; profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null && echo "pmc_traffic.json written"
  else echo "PMC passes incomplete"; fi
  find "$out" -name "*.db" -delete )
