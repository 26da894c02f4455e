#!/bin/bash
# round-2 GPU session G: full suite; window width 16 / 17 / 18 (split histograms); defaults after the arithmetic A/B
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2g
mkdir -p "$out"
cd "$root"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 24"
for c in 16 17 18 16 17; do
  ZKHIP_MSM_C=$c $B >> "$out/bench_c$c.json" 2>> "$out/bench.err"
done
tail -4 "$out/pytest_gpu.log"
for f in bench_c16 bench_c17 bench_c18; do python - "$out/$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; serial g1/g2', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), 'ntt', round(s.get('kernel_ntt_ms',0),3), 'pk_load', round(d['host_ms']['pk_load']))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -5 "$out/bench.err"
