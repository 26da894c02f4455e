#!/bin/bash
# same-box comparison of several environment settings of ONE build: bash tools/gpu_env_ab.sh <tag> "<VAR=a>" "<VAR=b>" ... [-- bench args]
# ("-" stands for the unmodified environment)
set -u
tag=$1; shift
settings=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do settings+=("$1"); shift; done
[ $# -gt 0 ] && shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; tail -5 "$out/smoke.log"; exit 0; }
for rep in 1 2; do
  for s in "${settings[@]}"; do
    name=$(echo "$s" | tr -c 'A-Za-z0-9_=\n' '_')
    if [ "$s" = "-" ]; then timeout 300 python bench.py --cpu-seconds 0 --steps 32 "$@" >> "$out/bench_$name.json" 2>> "$out/bench.err"
    else env $s timeout 300 python bench.py --cpu-seconds 0 --steps 32 "$@" >> "$out/bench_$name.json" 2>> "$out/bench.err"; fi
  done
done
for f in "$out"/bench_*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print('%-28s' % sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d['single_proof_from_host_ms'],2), 'from host; serial total', round(s.get('total_ms',0),2))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -2 "$out/bench.err"
