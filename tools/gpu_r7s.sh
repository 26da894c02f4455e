#!/bin/bash
# round 6, session r7s: the lone proof under k_msm_heavy_reduce's workgroup size, with enough samples (tools/lone_stats.py: 40 lone proofs per process;
# bench.py's min-of-three moved by +-0.3 ms between processes of ONE setting in r7q / r7r) — dense 2^20 BN254 and the Poseidon chain on BLS12-381
set -u
tag=${1:-r7s}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
for kind in dense poseidon; do
  for ht in 256 64 128 256 64 128 256 64; do
    ZKHIP_HEAVY_THREADS=$ht timeout 200 python3 tools/lone_stats.py $kind 40 2> /dev/null | tee -a "$out/lone_stats.jsonl" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  %-8s heavy threads %3d: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f' % (d['kind'], $ht, d['min'], d['p25'], d['median'], d['p75'], d['max']))"
  done
done
