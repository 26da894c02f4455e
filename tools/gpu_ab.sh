#!/bin/bash
# same-box A/B of two builds of the library: bash tools/gpu_ab.sh <tag> <libA> <libB> [bench args...]
set -u
tag=$1; la=$2; lb=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; tail -5 "$out/smoke.log"; exit 0; }
B="timeout 300 python bench.py --cpu-seconds 0 --steps 32 $*"
for rep in 1 2; do
  ZKHIP_LIBRARY=$root/$la $B >> "$out/bench_A.json" 2>> "$out/bench.err"
  ZKHIP_LIBRARY=$root/$lb $B >> "$out/bench_B.json" 2>> "$out/bench.err"
done
for f in A B; do python - "$out/bench_$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d['single_proof_from_host_ms'],2), 'from host; serial total', round(s.get('total_ms',0),2), 'g1/g2', round(s.get('kernel_msm_accum_g1_ms',0),2), round(s.get('kernel_msm_accum_g2_ms',0),2))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -2 "$out/bench.err"
