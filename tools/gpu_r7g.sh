#!/bin/bash
# round 6, session r7g: the stream plans the local search (r7f) liked, against streams in order of first use, three alternating rounds on four workloads,
# one bench.py process per (plan, workload); and the proof from host memory with the staging ring filled by several threads (ZKHIP_COPY_THREADS 1 / 6).
set -u
tag=${1:-r7g}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "from host memory: copy threads 1 / 6 / 1 / 6 (dense 2^20)"
for t in 1 6 1 6; do
  ZKHIP_COPY_THREADS=$t timeout 120 python3 bench.py --steps 8 --warmup 3 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 1 --oracle none --configs 0 --bind 2 2> /dev/null |
    python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  copy threads $t: single resident %.2f ms, from host %.2f ms, pk_load %.0f ms' % (d['single_proof_ms'], d['single_proof_from_host_ms'], d['host_ms']['pk_load']))"
done
P1="M=1,N=3,O=1,n=3,G0=1,Z0=0,H0=0,G1=2,Z1=2,H1=0,G2=3,Z2=2,H2=1"
P2="M=1,N=3,O=1,n=3,G0=0,Z0=0,H0=0,G1=2,Z1=2,H1=0,G2=3,Z2=2,H2=1"
P3="M=1,N=3,O=1,n=3,G0=0,Z0=3,H0=0,G1=2,Z1=2,H1=0,G2=3,Z2=2,H2=1"
run() { plan=$1; name=$2; shift 2
  ZKHIP_PIPES="$plan" timeout 150 python3 bench.py --steps 24 --warmup 6 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 2 --oracle none --configs 0 --bind 2 --pipe-plan 1 "$@" 2> /dev/null |
    python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ms=sorted(d['repeats']['ms_per_step']); print('   %-9s %-70s %7.2f proofs/s (regions %s)  lone %6.2f ms' % ('$name', '$plan', 1000/ms[0], [round(1000/m,1) for m in d['repeats']['ms_per_step']], d['single_proof_ms']))"
}
for rnd in 1 2 3; do
  for plan in - "$P1" "$P2" "$P3"; do
    step "round $rnd plan $plan"
    run "$plan" dense
    run "$plan" sha --kind sha256 --log-domain 20
    run "$plan" poseidon --curve bls12_381 --log-domain 18 --kind poseidon
    run "$plan" gm17 --scheme gm17
  done
done
step "done"
