#!/bin/bash
# round 6, session r: which hardware queue a lane's fold chain sits on — the fold on a second stream of its lane (fold_hop 0 / 1 / 2),
# the number of hardware queues, idle streams made first (ZKHIP_STREAM_SKEW) — lone proofs and a batch; then a kernel trace of lone proofs with the hop.
set -u
tag=${1:-r6r}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "parity with the hop on"
ZKHIP_FOLD_HOP=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bound.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_hop.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_hop.log"; tail -3 "$out/pytest_hop.log"
for q in 16 8 24; do
step "fold_hop 0 / 2 / 1 at $q hardware queues"
timeout 200 python tools/lone_ab.py $q fold_hop 0 2 1 > "$out/hop_q$q.txt" 2>&1; grep '^{' "$out/hop_q$q.txt" | cut -c1-400
done
for k in 1 2 3 5; do
step "ZKHIP_STREAM_SKEW=$k, 16 queues"
ZKHIP_STREAM_SKEW=$k ROUNDS=2 timeout 200 python tools/lone_ab.py 16 fold_hop 0 2 > "$out/skew$k.txt" 2>&1; grep '^{' "$out/skew$k.txt" | cut -c1-400
done
step "kernel trace of lone proofs, fold_hop 2"
( cd /tmp && ZKHIP_FOLD_HOP=2 timeout 120 rocprofv3 --kernel-trace -d "$out/prof_lone" -o lone -- python "$root/tools/lone_proof_probe.py" 20 8 1 > "$out/lone.log" 2>&1 )
db=$(find "$out/prof_lone" -name "*.db" | head -1)
[ -n "$db" ] && python tools/gantt.py "$db" > "$out/${tag}_lone_bound_proof_gantt_fold_hop2.txt" 2>&1
grep -E "fold|heavy|accum|clusters|proof:" "$out/${tag}_lone_bound_proof_gantt_fold_hop2.txt" | head -40
find "$out" -name "*.db" -size +8M -delete
step "done"
