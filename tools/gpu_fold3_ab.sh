#!/bin/bash
# three-digit fold at c = 16 (H = 128 rows -> 64 x 2) against the two-digit one, same box: bash tools/gpu_fold3_ab.sh <tag> [min_h values]
set -u
tag=$1; shift
vals=${*:-512 128}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
for v in $vals; do
  ZKHIP_FOLD3_MIN_H=$v timeout ${RUN_TIMEOUT:-40} python3 bench.py --steps 16 --warmup 3 --cpu-seconds 0 --serial-proofs 3 --e2e 0 > "$out/minh$v.json" 2> "$out/minh$v.err"
  python3 - "$out/minh$v.json" $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['phases_ms_serial']
    print('fold3_min_h', sys.argv[2], round(d['value'],2), 'proofs/s |', round(d['single_proof_ms'],2), 'ms single | serial total', round(s['total_ms'],2), 'msm_z', round(s['msm_z_ms'],2), 'msm_h', round(s['msm_h_ms'],2), 'finish', round(s['finish_ms'],3))
except Exception as e: print('fold3_min_h', sys.argv[2], 'ERR', e)
PY
done
