#!/bin/bash
# one-box A/B of the environment knobs on the default bench line: bash tools/gpu_knob_ab.sh <tag> "VAR=val" "VAR=val" ...   ("" = defaults)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
i=0
for kv in "$@"; do
  i=$((i+1))
  env $kv timeout ${RUN_TIMEOUT:-40} python3 bench.py --steps ${STEPS:-24} --warmup 4 --cpu-seconds 0 --serial-proofs 0 --e2e 0 > "$out/run$i.json" 2> "$out/run$i.err"
  python3 - "$out/run$i.json" "$kv" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-28s' % (sys.argv[2] or 'defaults'), round(d['value'],2), 'proofs/s |', round(d['single_proof_ms'],2), 'ms single |', round(d['single_proof_from_host_ms'],2), 'from host')
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
done
