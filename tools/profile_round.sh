#!/bin/bash
# rocprofv3 evidence of one round (run on the GPU box through gpurun); everything lands in gpurun_out/<tag>/ and the
# summaries are copied into profiles/ by hand.  Usage: bash tools/profile_round.sh <tag>
#   1. kernel trace of the default bench line, pipelined (the line itself is printed under the profiler)
#   2. kernel trace with every kernel on one stream (un-overlapped durations)
#   3. / 4.  PMC passes FETCH_SIZE and WRITE_SIZE (separate runs, kernel trace only) -> pmc_traffic.json
#   5. kernel trace of the GM17 leg
set -u
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export ZKHIP_BENCH_CHILD=1     # bench.py measures in this very process (no supervising parent between the profiler and the kernels)
run() {     # run <name> <rocprof args> -- <bench args>; env taken from the caller
  name=$1; shift
  rm -rf "$out/prof_$name"
  timeout ${PROF_TIMEOUT:-420} rocprofv3 "$@" > "$out/prof_$name.log" 2>&1
}
# 1. pipelined
run pipelined --kernel-trace --stats -d "$out/prof_pipelined" -o pipelined -- python "$root/bench.py" --cpu-seconds 0 --serial-proofs 0 --e2e 0
db=$(find "$out/prof_pipelined" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_g16_pipelined_kernel_stats.md" > /dev/null
[ -n "$db" ] && python "$root/tools/timeline.py" "$db" 0.6 > "$out/${tag}_g16_pipelined_timeline.txt" 2>&1
grep -h '^{"metric"' "$out/prof_pipelined.log" > "$out/${tag}_g16_bench_under_rocprof.json"
# 2. one stream
export ZKHIP_SERIAL=1
run serial --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --serial-proofs 0 --e2e 0
db=$(find "$out/prof_serial" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_g16_serial_kernel_stats.md" > /dev/null
# 3. / 4. counters (their own runs: --pmc with the kernel trace only); SKIP_PMC=1: tools/gpu_final.sh took them already
for ctr in ${SKIP_PMC:+} $([ -z "${SKIP_PMC:-}" ] && echo FETCH_SIZE WRITE_SIZE); do
  run pmc_$ctr --pmc $ctr --kernel-trace -d "$out/prof_pmc_$ctr" -o pmc -- python "$root/bench.py" --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0
  db=$(find "$out/prof_pmc_$ctr" -name "*.db" | head -1)
  [ -n "$db" ] && python "$root/tools/pmc_stats.py" "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null
done
f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
[ -z "${SKIP_PMC:-}" ] && [ -n "$f" ] && [ -n "$w" ] && python "$root/tools/pmc_traffic.py" "$f" "$w" "$out/pmc_traffic.json" \
  "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0; profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null
unset ZKHIP_SERIAL
# 5. GM17
run gm17 --kernel-trace --stats -d "$out/prof_gm17" -o gm17 -- python "$root/bench.py" --cpu-seconds 0 --scheme gm17 --steps 16 --serial-proofs 0 --e2e 0
db=$(find "$out/prof_gm17" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_gm17_pipelined_kernel_stats.md" > /dev/null
grep -h '^{"metric"' "$out/prof_gm17.log" > "$out/${tag}_gm17_bench_under_rocprof.json"
find "$out" -name "*.db" -size +8M -delete     # keep the merge-back small
head -14 "$out/${tag}_g16_serial_kernel_stats.md"; cat "$out/${tag}_g16_pipelined_timeline.txt"; [ -f "$out/pmc_traffic.json" ] && cat "$out/pmc_traffic.json"
