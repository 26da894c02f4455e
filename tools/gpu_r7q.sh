#!/bin/bash
# round 6, session r7q: k_msm_heavy_reduce in workgroups of 64 work-items instead of 256 (ZKHIP_HEAVY_THREADS): the launch exists in every fold chain, with or
# without heavy buckets, and has to find a place beside the accumulations before it can return.  Four workloads, alternating, one process per run.
set -u
tag=${1:-r7q}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "parity at 64 (the heavy-bucket tests: skewed scalars, the SHA-256 witness of bits)"
ZKHIP_HEAVY_THREADS=64 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sha256_circuit.py -m gpu -q -x -p no:cacheprovider > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -3 "$out/pytest.log"
for w in "--kind dense --log-domain 20" "--kind sha256 --log-domain 20" "--kind poseidon --curve bls12_381 --log-domain 18" "--scheme gm17 --log-domain 20"; do
  step "$w"
  for ht in 256 64 256 64; do
    ZKHIP_HEAVY_THREADS=$ht timeout 200 python3 bench.py $w --steps 32 --warmup 5 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 3 --oracle trapdoor --configs 0 --bind 2 2> /dev/null |
      python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  heavy threads %3d: %.1f proofs/s (regions %s)  lone %.2f ms  oracle %s' % ($ht, d['value'], [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['single_proof_ms'], d.get('identical_to_oracle')))"
  done
done
step "done"
