#!/bin/bash
# round 6, session z: the evidence of the round's LAST build — the driver's command, smoke + the GPU suite, the profiler passes
# (pipelined / one-stream kernel stats, the GM17 leg), the counter passes of the bound pipeline (tools/gpu_pmc_r6.sh).
set -u
tag=${1:-r6z}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "counter passes (first: the bench line reads them)"
bash tools/gpu_pmc_r6.sh "$tag" > "$out/pmc.log" 2>&1; tail -4 "$out/pmc.log" | cut -c1-500
cp "$out/pmc_valu.json" "$out/pmc_traffic.json" profiles/ 2>/dev/null
step "the driver's command"
( time timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench.err" ) 2>&1 | grep real
python - "$out/bench_driver_command.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('  value', round(d['value'],2), d['value_is'], 'unbound', d['value_unbound'] and round(d['value_unbound'],2), '| single', round(d['single_proof_ms'],2), d.get('single_proof_unbound_ms'), '| plan', d.get('stream_plan',{}).get('on'), '| repeats', d['repeats']['ms_per_step'])
        r=d['roofline']; print('  roofline', json.dumps({k:v for k,v in r.items() if k not in ('compute_bound','note','offline_evidence')})[:800])
        print('  offline', json.dumps(r.get('offline_evidence'))[:300])
        print('  compute_bound', json.dumps(r['compute_bound'])[:1200])
        print('  roofline_ntt', json.dumps(d['roofline_ntt'])[:600])
        for k,v in d.get('configs',{}).items(): print('  cfg', k, json.dumps({a:b for a,b in v.items() if a in ('proofs_per_s','proofs_per_s_unbound','single_proof_ms','identical_to_oracle','wall_s')}))
        print('  cli', {k:(round(v['process_wall_ms']) if isinstance(v,dict) and 'process_wall_ms' in v else None) for k,v in d['cli_end_to_end_ms'].items() if k.startswith('native')})
PY
tail -3 "$out/bench.err"
step "smoke + the GPU suite"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=6 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -12 "$out/pytest_gpu.log"
step "profiler passes"
SKIP_PMC=1 PROF_TIMEOUT=300 bash tools/profile_round.sh "$tag" > "$out/profile_round.log" 2>&1; tail -25 "$out/profile_round.log"
step "done"
