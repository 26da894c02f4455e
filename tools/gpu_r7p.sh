#!/bin/bash
# round 6, session r7p: the driver's command on the round's last build, twice, on whatever box the pool hands out (r7n's was of the slower kind)
set -u
tag=${1:-r7p}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
for k in 1 2; do
  ( time timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command_$k.json" 2> "$out/bench_$k.err" ) 2>&1 | grep real
  python3 - "$out/bench_driver_command_$k.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('  value', round(d['value'],2), d['value_is'], '| unbound', round(d['value_unbound'],2), '| lone', round(d['single_proof_ms'],2), 'from host', round(d['single_proof_from_host_ms'],2), '| regions', [round(1000/x,1) for x in d['repeats']['ms_per_step']], '| identical', d['identical_to_oracle'], '| issue bound', round(d['roofline']['compute_bound']['pipeline_issue_bound']['frac_of_ms_per_step'],3), '| stale counters', d['roofline']['offline_evidence']['stale'])
for k,v in d['configs'].items(): print('   ', k, round(v['proofs_per_s'],1), round(v['single_proof_ms'],2), v['identical_to_oracle'])
print('    cli', {k:(round(v['process_wall_ms']), v.get('process_wall_ms_runs')) for k,v in d['cli_end_to_end_ms'].items() if isinstance(v,dict) and 'process_wall_ms' in v})
print('    box', d['box_probe'], d['under_load']['sclk_mhz'], d['under_load']['power_w'])
PY
done
