#!/bin/bash
# round-2 GPU session E: full suite with the multi-device path and config 3, G2 register-allocation A/B, N = 2^21 variant, 2^22
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2e
mkdir -p "$out"
cd "$root"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 -s > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 24"
for v in A B C D; do
  lib=$root/zokrates_amd/libzkhip.so; [ $v != A ] && lib=$root/zokrates_amd_v2$v/libzkhip.so
  ZKHIP_LIBRARY=$lib $B > "$out/bench_g2_$v.json" 2>> "$out/bench.err"
done
$B --members 8 > "$out/bench_2e20_members8.json" 2>> "$out/bench.err"
timeout 600 python bench.py --cpu-seconds 0 --steps 16 --constraints 1048576 > "$out/bench_n2e20_domain2e21.json" 2>> "$out/bench.err"
timeout 900 python bench.py --cpu-seconds 0 --steps 8 --log-domain 22 --members 8 > "$out/bench_2e22.json" 2>> "$out/bench.err"
tail -6 "$out/pytest_gpu.log"; grep "config 3" "$out/pytest_gpu.log"
for f in bench_g2_A bench_g2_B bench_g2_C bench_g2_D bench_2e20_members8 bench_n2e20_domain2e21 bench_2e22; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get('roofline_ntt') or {}; s=d.get('phases_ms_serial') or {}
    print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; serial g1/g2', round(s.get('kernel_msm_accum_g1_ms',0),2), round(s.get('kernel_msm_accum_g2_ms',0),2), 'pipelined g2', round(d['phases_ms']['kernel_msm_accum_g2_ms'],2), 'multi', d.get('multi_single_proof'), d['host_ms'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -5 "$out/bench.err"
