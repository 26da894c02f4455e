#!/bin/bash
# round 5, session B: the accumulation loop with every fetch issued ahead of its use (bucket ends a bucket ahead, sorted entries two
# steps ahead, the boundary code before the step's fetches) against session A's builds, slicings, stall counters.  bash tools/gpu_r5b.sh [tag]
set -u
tag=${1:-r5b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
tail -1 "$out/smoke.log"
ZKHIP_LIBRARY=$root/zokrates_amd/libzkhip_w42p.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$out/parity_w42p.log" 2>&1; echo "parity(w42p) rc=$?"; tail -3 "$out/parity_w42p.log"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 32"
cfgs=(
 "r4|libzkhip_r4.so|"
 "w42_17|libzkhip_w42.so|"
 "p31_16|libzkhip.so|ZKHIP_MSM_C=16"
 "p31_17|libzkhip.so|"
 "p32_16|libzkhip_w32p.so|ZKHIP_MSM_C=16"
 "p32_17|libzkhip_w32p.so|"
 "p42_16|libzkhip_w42p.so|ZKHIP_MSM_C=16"
 "p42_17|libzkhip_w42p.so|"
 "p42_17_f4|libzkhip_w42p.so|ZKHIP_MSM_FUSED_WAVES=4 ZKHIP_MSM_G1_WAVES=4"
 "p42_17_f6|libzkhip_w42p.so|ZKHIP_MSM_FUSED_WAVES=6 ZKHIP_MSM_G1_WAVES=4"
 "p42_17_f8|libzkhip_w42p.so|ZKHIP_MSM_FUSED_WAVES=8 ZKHIP_MSM_G1_WAVES=4"
 "p42_17_f8g8|libzkhip_w42p.so|ZKHIP_MSM_FUSED_WAVES=8 ZKHIP_MSM_G1_WAVES=8"
 "p42_17_g23|libzkhip_w42p.so|ZKHIP_MSM_G2_WAVES=3"
 "p42_17_g24|libzkhip_w42p.so|ZKHIP_MSM_G2_WAVES=4"
 "p32_17_f6|libzkhip_w32p.so|ZKHIP_MSM_FUSED_WAVES=6"
)
for rep in 1 2; do
  for c in "${cfgs[@]}"; do
    IFS='|' read -r name lib envs <<< "$c"
    env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib $B >> "$out/bench_$name.json" 2>> "$out/bench.err"
  done
done
for c in "${cfgs[@]}"; do IFS='|' read -r name lib envs <<< "$c"; python - "$out/bench_$name.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print('%-14s' % sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; serial total', round(s.get('total_ms',0),2), 'g1/g2/ntt', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), round(s.get('kernel_ntt_ms',0),3))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 "$out/bench.err"
( cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1 ZKHIP_SERIAL=1
  A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
  for v in w42p w32p; do
    ZKHIP_LIBRARY=$root/zokrates_amd/libzkhip_$v.so timeout 240 rocprofv3 --pmc $A --kernel-trace -d "$out/prof_stall_${v}_A" -o pmc -- \
        python "$root/bench.py" --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_stall_${v}_A.log" 2>&1
    a=$(find "$out/prof_stall_${v}_A" -name "*.db" 2>/dev/null | head -1)
    [ -n "$a" ] && python "$root/tools/pmc_stall.py" "$out/${tag}_stall_$v.md" $a | head -4
    [ -n "$a" ] && python "$root/tools/pmc_valu.py" "$a" "$out/${tag}_valu_$v.md" | grep accum
  done
  find "$out" -name "*.db" -size +8M -delete )
