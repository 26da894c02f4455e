#!/bin/bash
# round 6, session r8b: why bench.py's lone proofs (9.6-10.2 ms) read slower than tools/lone_stats.py's (9.4): the chip after three pipelined batches against the
# chip a lone request finds.  16 lone proofs (bench.py's count) straight after 3 x 20 proofs, after a pause, and with no batch before them; alternating
set -u
tag=${1:-r8b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
for cfg in "0 0" "20 0" "20 0.5" "0 0" "20 0" "20 0.5" "20 2"; do
  set -- $cfg
  PRE_BATCH=$1 PAUSE_S=$2 timeout 300 python3 tools/lone_stats.py dense 16 2> /dev/null | tee -a "$out/lone_stats.jsonl" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  batches before: 3 x %-2s pause %-3s s: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f' % ('$1', '$2', d['min'], d['p25'], d['median'], d['p75'], d['max']))"
done
