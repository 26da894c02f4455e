#!/bin/bash
# round 5, session G: the sort with precomputed digits against the library before it (libzkhip_s1.so); fold doubling inline vs call at
# the tight register budgets; sort workgroup counts with the cheaper passes.
set -u
tag=${1:-r5g}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
tail -1 "$out/smoke.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$out/parity.log" 2>&1; echo "parity rc=$?"; tail -2 "$out/parity.log"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 64 --warmup 8 --serial-proofs 2"
cfgs=(
 "s1|libzkhip_s1.so|"
 "s2|libzkhip.so|"
 "s2_c16|libzkhip.so|ZKHIP_MSM_C=16"
 "s1_c16|libzkhip_s1.so|ZKHIP_MSM_C=16"
 "inl42|libzkhip_inl42.so|"
 "s2_wg512|libzkhip.so|ZKHIP_SORT_WGS=512"
 "s2_wg128|libzkhip.so|ZKHIP_SORT_WGS=128"
 "s2_kh14|libzkhip.so|ZKHIP_SORT_KH_LOG=14"
)
for rep in 1 2 3; do
  for c in "${cfgs[@]}"; do
    IFS='|' read -r name lib envs <<< "$c"
    env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib $B >> "$out/bench_$name.json" 2>> "$out/bench.err"
  done
done
python tools/ab_summary.py "$out"
tail -3 "$out/bench.err"
( cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1 ZKHIP_SERIAL=1
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --serial-proofs 0 --e2e 0 > "$out/prof_serial.log" 2>&1
  db=$(find "$out/prof_serial" -name "*.db" | head -1)
  [ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_g16_serial_kernel_stats.md" | head -30
  find "$out" -name "*.db" -size +8M -delete )
