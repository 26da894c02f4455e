#!/bin/bash
# quick same-box A/B on the Poseidon chain (BLS12-381, 2^18) and the sha-like 2^20 circuit: bash tools/gpu_abn_pos.sh <tag> <lib>...
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"; cd "$root"
for rep in 1 2; do for l in "$@"; do n=$(basename "$l" .so)
  ZKHIP_LIBRARY=$root/$l timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --curve bls12_381 --log-domain 18 --kind poseidon >> "$out/poseidon_$n.json" 2>> "$out/bench.err"
  ZKHIP_LIBRARY=$root/$l timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --kind poseidon --log-domain 20 --steps 16 >> "$out/poseidon_bn_2e20_$n.json" 2>> "$out/bench.err"
done; done
for w in poseidon poseidon_bn_2e20; do for l in "$@"; do n=$(basename "$l" .so); python - "$out/${w}_$n.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; serial total', round(s.get('total_ms',0),2), 'g1/g2/ntt', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), round(s.get('kernel_ntt_ms',0),3))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done; done
tail -2 "$out/bench.err"
