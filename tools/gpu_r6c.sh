#!/bin/bash
# round 6, session c: the driver's command with the configs block (how long does the whole line take?), the calibration with the
# transforms' access patterns, the suite.
set -u
tag=${1:-r6c}
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
        print('  compute_bound', json.dumps(d['roofline']['compute_bound'])[:900])
        for k,v in d.get('configs',{}).items(): print('  cfg', k, json.dumps({a:b for a,b in v.items() if a not in ('config','oracle')}))
        print('  cli', {k:(round(v['process_wall_ms']) if isinstance(v,dict) and 'process_wall_ms' in v else None) for k,v in d['cli_end_to_end_ms'].items() if k.startswith('native')})
        print('  timeline', d['timeline_s'])
PY
tail -5 "$out/bench.err"
step "FETCH_SIZE calibration with the transforms' patterns"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/fetch_calib tools/fetch_calib.hip 2> "$out/fetch_calib_build.log"
( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/prof_calib" -o calib -- "$root/tools/fetch_calib" > "$out/fetch_calib.log" 2>&1 )
db=$(find "$out/prof_calib" -name "*.db" | head -1)
[ -n "$db" ] && python tools/pmc_stats.py "$db" "$out/${tag}_fetch_calibration.md"
grep -E "stream|gather|pair|seg" "$out/fetch_calib.log" | tail -9
step "the GPU suite"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -4 "$out/pytest_gpu.log"
find "$out" -name "*.db" -size +8M -delete
step "done"
