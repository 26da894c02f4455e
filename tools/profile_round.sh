#!/bin/bash
# rocprofv3 kernel traces of the two bench legs (run on the GPU box through gpurun); summaries land in gpurun_out/
# and are copied into profiles/ by hand.  Usage: bash tools/profile_round.sh <tag>
set -u
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for leg in g16 gm17; do
  rm -rf "$out/prof_${tag}_$leg"
  rocprofv3 --kernel-trace --stats -d "$out/prof_${tag}_$leg" -o "${tag}_$leg" -- python "$root/bench.py" --cpu-seconds 0 --scheme $leg \
      > "$out/prof_${tag}_$leg.log" 2>&1
  db=$(find "$out/prof_${tag}_$leg" -name "*.db" | head -1)
  [ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_${leg}_kernel_stats.md" > /dev/null
  grep -h '^{"metric"' "$out/prof_${tag}_$leg.log" > "$out/${tag}_${leg}_bench_under_rocprof.json"
  find "$out/prof_${tag}_$leg" -name "*.db" -size +20M -delete     # keep the merge-back small
done
head -8 "$out/${tag}_g16_kernel_stats.md"; head -8 "$out/${tag}_gm17_kernel_stats.md"
