#!/bin/bash
# per-kernel table of the one-stream bench run under rocprofv3 (kernel trace): bash tools/gpu_kernel_table.sh <tag> [ENV=VAL ...]
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
ZKHIP_SERIAL=1 ZKHIP_BENCH_CHILD=1 timeout 400 rocprofv3 --kernel-trace --stats -d "$out/prof" -o serial -- python "$root/bench.py" --cpu-seconds 0 --steps 6 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof.log" 2>&1
db=$(find "$out/prof" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/kernel_stats.md" | cut -c1-110 | head -40
find "$out" -name "*.db" -size +8M -delete
