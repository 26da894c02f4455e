#!/bin/bash
# round 6, session p: the evidence of the round's last build — the driver's command, smoke + the GPU suite, then the profiler passes
# (tools/profile_round.sh: pipelined / one-stream kernel stats, FETCH_SIZE / WRITE_SIZE counter passes, the GM17 leg).
set -u
tag=${1:-r6p}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "the driver's command"
( time timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench.err" ) 2>&1 | grep real
python - "$out/bench_driver_command.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('  value', round(d['value'],2), d['value_is'], 'bound', d['value_bound'] and round(d['value_bound'],2), 'unbound', d['value_unbound'] and round(d['value_unbound'],2), '| single', round(d['single_proof_ms'],2), d.get('single_proof_unbound_ms'))
        print('  identical_to_oracle', d.get('identical_to_oracle'), '| clock', d['shader_clock'])
        print('  roofline', json.dumps({k:v for k,v in d['roofline'].items() if k!='compute_bound'})[:700])
        print('  compute_bound', json.dumps(d['roofline']['compute_bound'])[:900])
        print('  roofline_ntt', json.dumps(d['roofline_ntt'])[:700])
        for k,v in d.get('configs',{}).items(): print('  cfg', k, json.dumps({a:b for a,b in v.items() if a not in ('config','oracle')})[:500])
        print('  cli', {k:(round(v['process_wall_ms']) if isinstance(v,dict) and 'process_wall_ms' in v else None) for k,v in d['cli_end_to_end_ms'].items() if k.startswith('native')})
        print('  timeline', d['timeline_s'])
PY
tail -5 "$out/bench.err"
step "smoke + the GPU suite"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=6 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -12 "$out/pytest_gpu.log"
step "profiler passes"
PROF_TIMEOUT=300 bash tools/profile_round.sh "$tag" > "$out/profile_round.log" 2>&1; tail -30 "$out/profile_round.log"
step "done"
