#!/bin/bash
# round 6, session r7x: lone_sched bit 4 (a lone proof's witness map waits for the sort of its assignment) per workload, 40 lone proofs per process
# (tools/lone_stats.py), three processes per setting, alternating — r7w: dense medians 9.72-9.89 against 10.05-10.23 inside one process
set -u
tag=${1:-r7x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
for kind in dense sha256 poseidon; do
  for ls in 0 4 0 4 0 4; do
    ZKHIP_LONE_SCHED=$ls timeout 300 python3 tools/lone_stats.py $kind 40 2> /dev/null | tee -a "$out/lone_stats.jsonl" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  %-8s lone_sched %d: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f' % (d['kind'], $ls, d['min'], d['p25'], d['median'], d['p75'], d['max']))"
  done
done
