#!/bin/bash
# same-box comparison of N builds of the library: bash tools/gpu_abn.sh <tag> <reps> <lib1> <lib2> ... [-- bench args]
# (each lib: path relative to the repository root; the last one also runs the hot-path parity file)
set -u
tag=$1; reps=$2; shift 2
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ $# -gt 0 ] && shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
last=${libs[${#libs[@]}-1]}
ZKHIP_LIBRARY=$root/$last timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$out/parity_last.log" 2>&1; echo "parity($last) rc=$?"; tail -3 "$out/parity_last.log"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 32 $*"
for rep in $(seq 1 "$reps"); do
  for l in "${libs[@]}"; do
    n=$(basename "$l" .so)
    ZKHIP_LIBRARY=$root/$l $B >> "$out/bench_$n.json" 2>> "$out/bench.err"
  done
done
for l in "${libs[@]}"; do n=$(basename "$l" .so); python - "$out/bench_$n.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; serial total', round(s.get('total_ms',0),2), 'g1/g2/ntt', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), round(s.get('kernel_ntt_ms',0),3))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -2 "$out/bench.err"
