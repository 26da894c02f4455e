#!/bin/bash
# round 6, session r7h: the stream plan search once more, the lone proof's latency in the score (weight 1), from the plan r7f ended on
set -u
tag=${1:-r7h}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp PLAN_LONE_WEIGHT=1 PLAN_START="M=1,N=3,O=1,n=3,G0=0,Z0=3,H0=0,G1=2,Z1=2,H1=0,G2=3,Z2=2,H2=1"
timeout 1800 python tools/plan_search.py ${MINUTES:-24} ${SEED:-7} > "$out/plan_search.jsonl" 2> "$out/plan_search.err"
tail -3 "$out/plan_search.jsonl" | cut -c1-1500
