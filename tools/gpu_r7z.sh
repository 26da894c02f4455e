#!/bin/bash
# round 6, session r7z: does bench.py's clock probe (a second context whose one-wave kernel sleeps 2 ms at a time, back to back, beside the proofs) cost the
# proofs anything?  tools/lone_stats.py — no probe — reads lone proofs 0.2-0.4 ms faster than bench.py does.  With / without, alternating
set -u
tag=${1:-r7z}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
for np in "" 1 "" 1 "" 1; do
  ZKHIP_BENCH_NO_PROBE=$np timeout 200 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 3 --oracle none --configs 0 2> /dev/null |
    python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['single_proof_ms_stats']; print('  probe %-3s: regions %s proofs/s   lone min %.2f median %.2f p90 %.2f' % ('off' if '$np' else 'on', [round(1000/x,1) for x in d['repeats']['ms_per_step']], s['min'], s['median'], s['p90']))"
done
