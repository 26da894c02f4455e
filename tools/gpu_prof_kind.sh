#!/bin/bash
# kernel tables of one workload kind, one stream and pipelined: bash tools/gpu_prof_kind.sh <tag> -- <bench args>
set -u
tag=$1; shift; [ "${1:-}" = "--" ] && shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export ZKHIP_BENCH_CHILD=1
for mode in serial pipelined; do
  [ $mode = serial ] && export ZKHIP_SERIAL=1 || unset ZKHIP_SERIAL
  rm -rf "$out/prof_$mode"
  timeout 420 rocprofv3 --kernel-trace --stats -d "$out/prof_$mode" -o $mode -- python "$root/bench.py" --cpu-seconds 0 --steps 10 --warmup 2 --serial-proofs 0 --e2e 0 "$@" > "$out/prof_$mode.log" 2>&1
  db=$(find "$out/prof_$mode" -name "*.db" | head -1)
  [ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_${mode}_kernel_stats.md" > /dev/null
  grep -h '^{"metric"' "$out/prof_$mode.log" > "$out/${tag}_${mode}_bench_under_rocprof.json"
done
find "$out" -name "*.db" -size +8M -delete
head -30 "$out/${tag}_serial_kernel_stats.md" | cut -c1-150
