#!/bin/bash
# round 5, session l: the N > 1 path of bench.py with bound keys on hardware (two ranks sharing this box's one GPU, gloo: every rank binds
# its own key) and a kernel timeline of the bound pipeline (how much of the steady-state window has an accumulation kernel in flight).
set -u
tag=${1:-r5l}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
ZKHIP_DIST_BACKEND=gloo ZKHIP_BENCH_DEVICE=0 timeout 120 python bench.py --gpus 2 --steps 16 --warmup 4 --e2e 0 > "$out/bench_gpus2_self_spawned_one_gpu.json" 2> "$out/bench_gpus2.err"
python - "$out/bench_gpus2_self_spawned_one_gpu.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('two ranks on one GPU:', round(d['value'],2), 'proofs/s aggregate, n_gpus', d['n_gpus'], '| bound_key', {k:v for k,v in d.get('bound_key',{}).items() if k!='note'}, '| per_rank', [(r['rank'], round(r['ms_per_step'],2)) for r in d.get('per_rank',[])],
              '| sharded', d.get('sharded_single_proof'), '| multi', {k:v for k,v in (d.get('multi_single_proof') or {}).items() if k in ('ms','members','identical_to_unsharded','error')})
PY
tail -2 "$out/bench_gpus2.err"
export TMPDIR=/tmp
( cd /tmp && ZKHIP_BENCH_CHILD=1 timeout 90 rocprofv3 --kernel-trace -d "$out/prof_pipe" -o st -- python "$root/bench.py" --bind 2 --cpu-seconds 0 --steps 32 --warmup 5 --serial-proofs 0 --e2e 0 --repeats 1 > "$out/prof_pipe.log" 2>&1 )
db=$(find "$out/prof_pipe" -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/timeline.py "$db" --proofs 14 34 > "$out/${tag}_bound_pipelined_timeline.txt" 2>&1 && tail -12 "$out/${tag}_bound_pipelined_timeline.txt"
find "$out" -name "*.db" -size +8M -delete
