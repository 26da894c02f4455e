#!/bin/bash
# round-2 GPU session F: fused arithmetic of the hot mixed addition (Karatsuba Fq2 product, one-reduction Y3) — parity + same-box A/B
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2f
mkdir -p "$out"
cd "$root"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 24"
for v in main K0 L0 K0L0 main; do
  lib=$root/zokrates_amd/libzkhip.so; [ $v != main ] && lib=$root/zokrates_amd_v2$v/libzkhip.so
  ZKHIP_LIBRARY=$lib $B >> "$out/bench_$v.json" 2>> "$out/bench.err"
done
$B --curve bls12_381 --log-domain 18 --kind poseidon > "$out/bench_poseidon_bls.json" 2>> "$out/bench.err"
$B --scheme gm17 > "$out/bench_gm17.json" 2>> "$out/bench.err"
tail -4 "$out/pytest_gpu.log"; grep "config 3" "$out/pytest_gpu.log"
for f in bench_main bench_K0 bench_L0 bench_K0L0 bench_poseidon_bls bench_gm17; do python - "$out/$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; serial g1/g2', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), 'ntt', round(s.get('kernel_ntt_ms',0),3))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -5 "$out/bench.err"
