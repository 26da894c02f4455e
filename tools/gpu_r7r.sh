#!/bin/bash
# round 6, session r7r: lighter workgroups in the fold chain, continued — k_msm_heavy_reduce at 64 work-items (r7q) and k_msm_fold_cols at 64 / 128 instead of
# 256 (fewer columns per workgroup, the same chain per column).  (heavy, cols) alternating, one process per run
set -u
tag=${1:-r7r}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "parity at (64, 64)"
ZKHIP_HEAVY_THREADS=64 ZKHIP_FOLD_COLS_THREADS=64 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sha256_circuit.py tests/test_gm17.py -m gpu -q -x -p no:cacheprovider > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -3 "$out/pytest.log"
for w in "--kind dense --log-domain 20" "--kind sha256 --log-domain 20" "--kind poseidon --curve bls12_381 --log-domain 18"; do
  step "$w"
  for cfg in "256 256" "64 256" "64 64" "64 128" "256 256" "64 256" "64 64" "64 128"; do
    set -- $cfg
    ZKHIP_HEAVY_THREADS=$1 ZKHIP_FOLD_COLS_THREADS=$2 timeout 200 python3 bench.py $w --steps 32 --warmup 5 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 3 --oracle trapdoor --configs 0 --bind 2 2> /dev/null |
      python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  heavy %3d cols %3d: %.1f proofs/s (regions %s)  lone %.2f ms  oracle %s' % ($1, $2, d['value'], [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['single_proof_ms'], d.get('identical_to_oracle')))"
  done
done
step "done"
