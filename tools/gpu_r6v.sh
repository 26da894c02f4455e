#!/bin/bash
# round 6, session v: the stream plan as bench.py's default — the driver's command, the same command with --pipe-plan 0 (short form),
# smoke + the GPU suite, the counter passes of this build.
set -u
tag=${1:-r6v}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
line() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('  value', round(d['value'],2), d['value_is'], 'bound', d['value_bound'] and round(d['value_bound'],2), 'unbound', d['value_unbound'] and round(d['value_unbound'],2), '| single', round(d['single_proof_ms'],2), d.get('single_proof_unbound_ms'), '| plan', d.get('stream_plan',{}).get('on'))
        print('  repeats', d['repeats']['ms_per_step'], '| identical_to_oracle', d.get('identical_to_oracle'))
        print('  roofline', json.dumps({k:v for k,v in d['roofline'].items() if k in ('frac','frac_serial','traffic','traffic_ratio','ms_per_launch_serial')}), '| pipeline', json.dumps(d['roofline']['compute_bound'].get('pipeline_issue_bound'))[:300])
        print('  roofline_ntt', json.dumps({k:v for k,v in d['roofline_ntt'].items() if k in ('frac_serial','us_per_pass_serial','traffic_ratio')}))
        for k,v in d.get('configs',{}).items(): print('  cfg', k, json.dumps({a:b for a,b in v.items() if a in ('proofs_per_s','proofs_per_s_unbound','single_proof_ms','identical_to_oracle','wall_s')}), json.dumps(v.get('members'))[:200] if v.get('members') else '')
        if 'cli_end_to_end_ms' in d: print('  cli', {k:(round(v['process_wall_ms']) if isinstance(v,dict) and 'process_wall_ms' in v else None) for k,v in d['cli_end_to_end_ms'].items() if k.startswith('native')})
PY
}
step "the driver's command"
( time timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench.err" ) 2>&1 | grep real
line "$out/bench_driver_command.json"; tail -3 "$out/bench.err"
step "the same region without the plan"
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --pipe-plan 0 --configs 0 --e2e 0 --cpu-seconds 0 > "$out/bench_no_plan.json" 2> "$out/bench_no_plan.err"; line "$out/bench_no_plan.json"
step "and with it again"
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --configs 0 --e2e 0 --cpu-seconds 0 > "$out/bench_plan_2.json" 2> "$out/bench_plan_2.err"; line "$out/bench_plan_2.json"
step "smoke + the GPU suite"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=6 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -12 "$out/pytest_gpu.log"
step "counter passes"
bash tools/gpu_pmc_r6.sh "$tag" > "$out/pmc.log" 2>&1; tail -6 "$out/pmc.log" | cut -c1-600
step "done"
