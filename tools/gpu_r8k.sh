#!/bin/bash
# round 6, session r8k: do bench.py's own observers (the hwmon sampler at 4 ms, the clock probe's second context) cost the regions they observe?  both on /
# both off / the probe alone, alternating, the driver's steps, three regions each
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
run() {
  timeout 200 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 3 --oracle none --configs 0 2> /dev/null |
    LABEL="$1" python3 -c "import os,sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  %-16s regions %s proofs/s' % (os.environ['LABEL'], [round(1000/x,1) for x in d['repeats']['ms_per_step']]))"
}
for rep in 1 2 3; do
  run "observers on"
  ZKHIP_BENCH_NO_PROBE=1 ZKHIP_BENCH_SAMPLER_PERIOD_S=0 run "observers off"
  ZKHIP_BENCH_SAMPLER_PERIOD_S=0 run "probe only"
done
