#!/bin/bash
# round 5, session E: does the sort's 128 KiB histogram block the kernels beside it?  Same build, window width x histogram size.
set -u
tag=${1:-r5e}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 64 --warmup 8 --serial-proofs 2"
cfgs=(
 "c17_kh15|"
 "c17_kh14|ZKHIP_SORT_KH_LOG=14"
 "c17_kh13|ZKHIP_SORT_KH_LOG=13"
 "c16_kh15|ZKHIP_MSM_C=16"
 "c16_kh14|ZKHIP_MSM_C=16 ZKHIP_SORT_KH_LOG=14"
 "c16_kh13|ZKHIP_MSM_C=16 ZKHIP_SORT_KH_LOG=13"
 "c16_kh12|ZKHIP_MSM_C=16 ZKHIP_SORT_KH_LOG=12"
 "c15_kh14|ZKHIP_MSM_C=15"
 "hwq8|GPU_MAX_HW_QUEUES=8"
 "hwq24|GPU_MAX_HW_QUEUES=24"
)
for rep in 1 2; do
  for c in "${cfgs[@]}"; do
    IFS='|' read -r name envs <<< "$c"
    env $envs $B >> "$out/bench_$name.json" 2>> "$out/bench.err"
  done
done
for c in "${cfgs[@]}"; do IFS='|' read -r name envs <<< "$c"; python - "$out/bench_$name.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); r=d['repeats']; s=d.get('phases_ms_serial') or {}
        print('%-10s' % sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s; median', round(r['median_ms_per_step'],3), 'single', round(d['single_proof_ms'],2), '| serial total', round(s.get('total_ms',0),2), 'g1/g2/ntt', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), round(s.get('kernel_ntt_ms',0),3), '| W', round((d['under_load'].get('power_w') or {}).get('mean',0)))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 "$out/bench.err"
