#!/bin/bash
# round 5, session A (one gpurun call): the accumulation kernel with its general path out of line — register budgets 4 waves per
# SIMD (G1) / 2 (G2) — and 17-bit windows, against round 4's build on the SAME box; stall counters of old and new.
#   bash tools/gpu_r5a.sh [tag]
set -u
tag=${1:-r5a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
{ rocminfo | grep -E "Marketing Name|Compute Unit:|Max Clock|gfx9" | head -12; rocm-smi --showclocks --showpower --showtemp --showmaxpower 2>/dev/null | head -40; } > "$out/box.txt" 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
tail -1 "$out/smoke.log"
for v in r4 cold w42 w52xy; do timeout 180 tools/accum_bench_$v > "$out/accum_bench_$v.txt" 2>&1; echo "== accum_bench_$v"; cat "$out/accum_bench_$v.txt"; done
ZKHIP_LIBRARY=$root/zokrates_amd/libzkhip_w42.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$out/parity_w42.log" 2>&1; echo "parity(w42) rc=$?"; tail -3 "$out/parity_w42.log"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 32"
cfgs=(
 "r4|libzkhip_r4.so|"
 "cold16|libzkhip.so|ZKHIP_MSM_C=16"
 "cold17|libzkhip.so|"
 "g1w4_16|libzkhip_g1w4.so|ZKHIP_MSM_C=16"
 "g2w2_16|libzkhip_g2w2.so|ZKHIP_MSM_C=16"
 "w42_16|libzkhip_w42.so|ZKHIP_MSM_C=16"
 "w42_17|libzkhip_w42.so|"
 "w42_17_f4|libzkhip_w42.so|ZKHIP_MSM_FUSED_WAVES=4 ZKHIP_MSM_G1_WAVES=4"
 "w42_17_f8|libzkhip_w42.so|ZKHIP_MSM_FUSED_WAVES=8 ZKHIP_MSM_G1_WAVES=4"
 "w42_17_f8g8|libzkhip_w42.so|ZKHIP_MSM_FUSED_WAVES=8 ZKHIP_MSM_G1_WAVES=8 ZKHIP_MSM_G2_WAVES=4"
 "w42xy_17|libzkhip_w42xy.so|"
 "w42_17_s4|libzkhip_w42.so|ZKHIP_SLOTS=4"
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
# stall counters: two PMC passes each (their own runs, kernel trace only), one stream, for round 4's build and the new one
( cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1 ZKHIP_SERIAL=1
  A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
  Bc="SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQC_ICACHE_REQ SQC_ICACHE_MISSES TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
  for v in r4 w42; do
    for pass in A B; do
      ctrs=$A; [ $pass = B ] && ctrs=$Bc
      ZKHIP_LIBRARY=$root/zokrates_amd/libzkhip_$v.so timeout 240 rocprofv3 --pmc $ctrs --kernel-trace -d "$out/prof_stall_${v}_$pass" -o pmc -- \
        python "$root/bench.py" --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_stall_${v}_$pass.log" 2>&1
    done
    a=$(find "$out/prof_stall_${v}_A" -name "*.db" 2>/dev/null | head -1); b=$(find "$out/prof_stall_${v}_B" -name "*.db" 2>/dev/null | head -1)
    [ -n "$a" ] && python "$root/tools/pmc_stall.py" "$out/${tag}_stall_$v.md" $a $b
    [ -n "$a" ] && python "$root/tools/pmc_valu.py" "$a" "$out/${tag}_valu_$v.md" > /dev/null
  done
  find "$out" -name "*.db" -size +8M -delete )
