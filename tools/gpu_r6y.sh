#!/bin/bash
# round 6, session y: the lone proof under the built-in plan again (the G2 lane's fold hops in lone proofs too), in bench.py's own process
# and in the probe's; plan / no plan alternating, one box.
set -u
tag=${1:-r6y}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
for i in 1 2; do
for plan in 1 0; do
  ZKHIP_BENCH_CHILD=1 timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --pipe-plan $plan --configs 0 --e2e 0 --cpu-seconds 0 --serial-proofs 0 > "$out/bench_plan${plan}_$i.json" 2> "$out/bench_plan${plan}_$i.err"
  python - "$out/bench_plan${plan}_$i.json" $plan <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('  bench plan', sys.argv[2], 'value', round(d['value'],2), [round(x,3) for x in d['repeats']['ms_per_step']], 'unbound', round(d['value_unbound'],2), '| single', round(d['single_proof_ms'],2), round(d.get('single_proof_unbound_ms') or 0,2), '| from host', round(d['single_proof_from_host_ms'],2))
PY
done
done
ZKHIP_PIPES=1 ROUNDS=2 timeout 120 python tools/lone_ab.py 16 none 0 | grep '^{' | cut -c1-200
ZKHIP_PIPES=- ROUNDS=2 timeout 120 python tools/lone_ab.py 16 none 0 | grep '^{' | cut -c1-200
