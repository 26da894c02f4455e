#!/bin/bash
# round 6, session r7f: local search over stream plans on four workloads at once (tools/plan_search.py), 22 minutes
set -u
tag=${1:-r7f}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
timeout 1700 python tools/plan_search.py ${MINUTES:-22} ${SEED:-1} > "$out/plan_search.jsonl" 2> "$out/plan_search.err"
tail -3 "$out/plan_search.jsonl" | cut -c1-1200
