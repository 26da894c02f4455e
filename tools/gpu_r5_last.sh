#!/bin/bash
# round 5, last session: same-box A/B of the final build against the one before the loose quotient digits, then the whole final
# verification (tools/gpu_final.sh) and the kernel traces of the final build (tools/profile_round.sh, counters taken by gpu_final).
set -u
tag=${1:-r5_final3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}_ab
mkdir -p "$out"
cd "$root"
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 64 --warmup 8 --serial-proofs 2"
for rep in 1 2 3; do
  ZKHIP_LIBRARY=$root/zokrates_amd/libzkhip_preloose.so $B >> "$out/bench_before_loose_digits.json" 2>> "$out/bench.err"
  $B >> "$out/bench_final_build.json" 2>> "$out/bench.err"
done
python tools/ab_summary.py "$out"
bash tools/gpu_final.sh $tag
SKIP_PMC=1 PROF_TIMEOUT=300 bash tools/profile_round.sh ${tag}_prof > "$root/gpurun_out/${tag}_prof.log" 2>&1; tail -8 "$root/gpurun_out/${tag}_prof.log"
