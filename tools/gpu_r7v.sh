#!/bin/bash
# round 6, session r7v: what separates a fast lone proof from a slow one — a kernel trace of 40 lone proofs under the stream plan, milestones per proof
set -u
tag=${1:-r7v}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_PIPES=1
( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d "$out/prof_lone" -o lone -- python "$root/tools/lone_proof_probe.py" 20 40 1 > "$out/lone.log" 2>&1 )
db=$(find "$out/prof_lone" -name "*.db" | head -1)
[ -n "$db" ] && python tools/lone_milestones.py "$db" > "$out/${tag}_lone_milestones.txt" 2>&1
[ -n "$db" ] && python tools/gantt.py "$db" -3 > "$out/${tag}_gantt_a.txt" 2>&1
cat "$out/${tag}_lone_milestones.txt"
tail -3 "$out/lone.log"
find "$out" -name "*.db" -size +8M -delete
