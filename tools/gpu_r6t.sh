#!/bin/bash
# round 6, session t: the streams of a resident prover placed on the four dispatchers by plan (ZKHIP_PIPES, core.cuh make_pipe_streams):
# lone proofs and a pipelined batch per plan, one process each (dense 2^20 BN254, key bound, 16 hardware queues).
set -u
tag=${1:-r6t}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
i=0
while read -r plan; do
  [ -z "$plan" ] && continue
  [ "$plan" = "-" ] && plan=""
  i=$((i+1))
  step "plan $i: '$plan'"
  ZKHIP_PIPES="$plan" ROUNDS=3 timeout 120 python tools/lone_ab.py 16 none 0 > "$out/plan$i.txt" 2>&1
  grep '^{' "$out/plan$i.txt" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   lone', d['lone_ms'][:5], ' batch', d['proofs_per_s'])"
  grep -v '^{' "$out/plan$i.txt" | tail -2
done <<'PLANS'
-
M=1,N=2,n=3,O=0,G=0,Z=1,H=2,g=3,z=3,h=3
M=1,N=2,O=0,G=0,Z=1,H=2,g=3,z=3,h=3
M=1,N=3,O=3,G=0,Z=1,H=2,g=3
M=1,N=2,n=3,O=0,G=0,Z=1,H=2,g=3,z=0,h=3
M=1,N=2,n=3,O=3,G=0,Z=1,H=2,g=3,z=3,h=0
M=1,N=2,n=0,O=3,G=0,Z=1,H=2,g=3,z=3,h=3
M=1,N=2,n=3,O=0,G=0,Z=1,H=2,g=3,z=3,h=2
1
PLANS
step "done"
