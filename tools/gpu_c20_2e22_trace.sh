#!/bin/bash
# one-stream kernel table of a 2^22 proof at c = 20 (where does the time the accumulation saves go?): bash tools/gpu_c20_2e22_trace.sh <tag> [c]
set -u
tag=${1:-c20trace}; c=${2:-20}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1 ZKHIP_SERIAL=1 ZKHIP_MSM_C=$c
timeout ${RUN_TIMEOUT:-90} rocprofv3 --kernel-trace --stats -d "$out/prof" -o serial -- python "$root/bench.py" --log-domain 22 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0 --e2e 0 > "$out/prof.log" 2>&1
echo "rocprofv3 rc=$?"
db=$(find "$out/prof" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_serial_kernel_stats.md" > /dev/null
find "$out" -name "*.db" -delete
head -30 "$out/${tag}_serial_kernel_stats.md" | cut -c1-110
