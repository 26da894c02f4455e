#!/bin/bash
# the same sweep at 2^22 constraints (BASELINE config 3 on one GPU): bash tools/gpu_msm_c_sweep_2e22.sh <tag> [c values...]
set -u
tag=$1; shift
cs=${*:-16 20}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
for c in $cs; do
  ZKHIP_MSM_C=$c timeout ${RUN_TIMEOUT:-70} python3 bench.py --log-domain 22 --steps 6 --warmup 2 --cpu-seconds 0 --serial-proofs 2 --e2e 0 > "$out/c$c.json" 2> "$out/c$c.err"
  python3 - "$out/c$c.json" $c <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['phases_ms_serial']
    print('2^22 c', sys.argv[2], round(d['value'],2), 'proofs/s |', round(d['single_proof_ms'],2), 'ms single | serial total', round(s['total_ms'],2), 'accum g1/g2', round(s['kernel_msm_accum_g1_ms'],2), round(s['kernel_msm_accum_g2_ms'],2), 'msm_z', round(s['msm_z_ms'],2), 'msm_h', round(s['msm_h_ms'],2), '| pk_load', round(d['host_ms']['pk_load']), '| identical', (d.get('cpu_baseline') or {}).get('gpu_proof_identical'))
except Exception as e: print('c', sys.argv[2], 'ERR', e)
PY
done
