#!/bin/bash
# round 6, session r8h: why the configuration legs of the bench line read 5-8 % below the same workloads run alone — the leg's own arguments, run with no
# parent around them, against the arguments of the A/B sessions (r7q)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
run() { timeout 200 python3 bench.py "$@" 2> /dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   %.1f proofs/s (regions %s)  lone %.2f' % (d['value'], [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['single_proof_ms']))"; }
for w in "--scheme gm17" "--curve bls12_381 --log-domain 18 --kind poseidon"; do
  echo "$w"
  for rep in 1 2; do
    echo "  the leg's arguments (--witnesses 2, bind 1, warm-up 4):"; run $w --steps 32 --warmup 4 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 2 --oracle trapdoor --configs 0
    echo "  the same with a witness per step:";                      run $w --steps 32 --warmup 4 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 2 --oracle trapdoor --configs 0
    echo "  r7q's arguments (bind 2, warm-up 5, repeats 3):";        run $w --steps 32 --warmup 5 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 3 --oracle trapdoor --configs 0 --bind 2
  done
done
