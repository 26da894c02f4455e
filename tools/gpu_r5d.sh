#!/bin/bash
# round 5, session D: scheduling knobs of the new default build (same box, 64-step regions), the warm-up through the pipelined path,
# and whether a CLI process pays for its predecessor's teardown.   bash tools/gpu_r5d.sh [tag]
set -u
tag=${1:-r5d}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 64 --warmup 8 --serial-proofs 0"
cfgs=(
 "default|"
 "gate0|ZKHIP_Z_GATE=0"
 "gate2|ZKHIP_Z_GATE=2"
 "slots2|ZKHIP_SLOTS=2"
 "slots4|ZKHIP_SLOTS=4"
 "hwq4|GPU_MAX_HW_QUEUES=4"
 "hwq12|GPU_MAX_HW_QUEUES=12"
 "hwq16|GPU_MAX_HW_QUEUES=16"
 "g2prio0|ZKHIP_G2_PRIORITY=0"
 "sort128|ZKHIP_SORT_WGS=128"
 "sort512|ZKHIP_SORT_WGS=512"
 "fused5|ZKHIP_MSM_FUSED_WAVES=5"
 "fused8|ZKHIP_MSM_FUSED_WAVES=8"
 "g1s6|ZKHIP_MSM_G1_WAVES=6"
 "g2s2|ZKHIP_MSM_G2_WAVES=2"
 "g2s6|ZKHIP_MSM_G2_WAVES=6"
 "c16|ZKHIP_MSM_C=16"
 "nofuse|ZKHIP_FUSE_Z=0"
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
        d=json.loads(line); r=d['repeats']
        print('%-10s' % sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s; regions', [round(x,3) for x in r['ms_per_step']], 'median', round(r['median_ms_per_step'],3), 'single', round(d['single_proof_ms'],2), 'sclk', round((d['under_load'].get('sclk_mhz') or {}).get('mean',0)), 'W', round((d['under_load'].get('power_w') or {}).get('mean',0)))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 "$out/bench.err"
# the CLI legs: back to back, and with a pause before each process
timeout 600 python bench.py --cpu-seconds 0 --steps 8 --warmup 4 --serial-proofs 0 --repeats 1 > "$out/bench_cli_back_to_back.json" 2>> "$out/bench.err"
ZKHIP_BENCH_CLI_GAP_S=1.5 timeout 600 python bench.py --cpu-seconds 0 --steps 8 --warmup 4 --serial-proofs 0 --repeats 1 > "$out/bench_cli_spaced.json" 2>> "$out/bench.err"
for f in back_to_back spaced; do python - "$out/bench_cli_$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); e=d.get('cli_end_to_end_ms') or {}
print(sys.argv[1].split('/')[-1], {k:(round(v.get('process_wall_ms',0)), round(v.get('total_in_process_ms',0)), round(v.get('hip_init_ms',0))) for k,v in e.items() if isinstance(v,dict) and 'process_wall_ms' in v})
PY
done
