#!/bin/bash
# round 5, session F: the two accumulation designs the product does not use (tools/accum_alt_bench), and the fold kernels with the
# doubling as a call / tighter register budgets.
set -u
tag=${1:-r5f}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 300 tools/accum_alt_bench > "$out/accum_alt_bench.txt" 2>&1; cat "$out/accum_alt_bench.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
tail -1 "$out/smoke.log"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 64 --warmup 8 --serial-proofs 2"
cfgs=(
 "dblcall|libzkhip.so|"
 "nodblcall|libzkhip_nodblcall.so|"
 "cold42|libzkhip_cold42.so|"
)
for rep in 1 2 3; do
  for c in "${cfgs[@]}"; do
    IFS='|' read -r name lib envs <<< "$c"
    env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib $B >> "$out/bench_$name.json" 2>> "$out/bench.err"
  done
done
for c in "${cfgs[@]}"; do
  IFS='|' read -r name lib envs <<< "$c"
  env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --curve bls12_381 --log-domain 18 --kind poseidon >> "$out/bench_poseidon_$name.json" 2>> "$out/bench.err"
  env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --kind sha256 >> "$out/bench_sha256_$name.json" 2>> "$out/bench.err"
  env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --scheme gm17 >> "$out/bench_gm17_$name.json" 2>> "$out/bench.err"
done
for f in "$out"/bench_*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); r=d['repeats']; s=d.get('phases_ms_serial') or {}
        print('%-22s' % sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s; median', round(r['median_ms_per_step'],3), 'single', round(d['single_proof_ms'],2), '| serial total', round(s.get('total_ms',0),2), 'g1/g2/ntt', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), round(s.get('kernel_ntt_ms',0),3))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 "$out/bench.err"
