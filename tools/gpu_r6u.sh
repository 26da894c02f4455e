#!/bin/bash
# round 6, session u: with the streams on their pipes by plan (ZKHIP_PIPES=1), the scheduling knobs again — z_gate, proofs in flight,
# slices per lane — and a kernel trace of lone proofs under the plan.
set -u
tag=${1:-r6u}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
export ZKHIP_PIPES=${PLAN:-1}
show() { grep '^{' "$1" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', {k: v for k, v in d.items() if k not in ('hw_queues', 'env', 'lone_ms', 'proofs_per_s', 'round')}, 'lone', d['lone_ms'][:4], ' batch', d['proofs_per_s'])"; grep -v '^{' "$1" | tail -2; }
step "z_gate 1 / 0 / 2"
ROUNDS=2 timeout 120 python tools/lone_ab.py 16 z_gate 1 0 2 > "$out/z_gate.txt" 2>&1; show "$out/z_gate.txt"
step "slots 3 / 4 / 2"
ROUNDS=2 timeout 120 python tools/lone_ab.py 16 slots 3 4 2 > "$out/slots.txt" 2>&1; show "$out/slots.txt"
step "msm_fused_waves 6 / 4 / 8"
ROUNDS=2 timeout 120 python tools/lone_ab.py 16 msm_fused_waves 6 4 8 > "$out/fused_waves.txt" 2>&1; show "$out/fused_waves.txt"
step "lone_sched 0 / 2"
ROUNDS=2 timeout 120 python tools/lone_ab.py 16 lone_sched 0 2 > "$out/lone_sched.txt" 2>&1; show "$out/lone_sched.txt"
step "kernel trace of lone proofs under the plan"
( cd /tmp && timeout 120 rocprofv3 --kernel-trace -d "$out/prof_lone" -o lone -- python "$root/tools/lone_proof_probe.py" 20 8 1 > "$out/lone.log" 2>&1 )
db=$(find "$out/prof_lone" -name "*.db" | head -1)
[ -n "$db" ] && python tools/gantt.py "$db" > "$out/${tag}_lone_bound_proof_gantt_pipe_plan.txt" 2>&1
cat "$out/${tag}_lone_bound_proof_gantt_pipe_plan.txt" | head -70
find "$out" -name "*.db" -size +8M -delete
step "done"
