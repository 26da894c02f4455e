#!/bin/bash
# round 6, session r7w: a lone proof with the witness map held for the assignment's sort (lone_sched bit 4: the sort has the machine to itself, 0.25 ms
# sooner) AND the G1 lanes over z not waiting for the witness map (z_gate 0) — r7v says the witness map has 3 ms of slack before H needs it.
# tools/lone_ab.py alternates lone_sched inside one process per z_gate; the batch column of the z_gate 0 rows is NOT the product's (batches keep gate 1)
set -u
tag=${1:-r7w}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_PIPES=1
for gate in 1 0 1 0; do
  echo "z_gate $gate"
  ZKHIP_Z_GATE=$gate ROUNDS=3 timeout 300 python tools/lone_ab.py 16 lone_sched 0 4 2> /dev/null | tee -a "$out/lone_ab_gate$gate.txt" | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); x = d['lone_ms']
    print('   lone_sched %d: min %.2f median %.2f max %.2f   (batch %.1f)' % (d['lone_sched'], x[0], x[len(x)//2], x[-1], d['proofs_per_s']))"
done
