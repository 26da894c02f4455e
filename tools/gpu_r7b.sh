#!/bin/bash
# round 6, session 7b: the thin workloads (accumulations in flight 73 % of the SHA-256 pipeline's window): proofs in flight and the z gate per workload.
set -u
tag=${1:-r7b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
leg() {  # leg <name> <label> <bench args...>
  local name=$1 label=$2; shift 2
  timeout 120 python3 bench.py --steps 32 --warmup 6 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 3 --oracle none --configs 0 --bind 2 "$@" > "$out/$name.json" 2> "$out/$name.err"
  python - "$out/$name.json" "$name" "$label" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('   %-10s %-28s bound %7.2f (regions %s)  lone %6.2f' % (sys.argv[2], sys.argv[3], d['value'], [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['single_proof_ms']))
PY
}
for cfg in "3 1" "4 1" "3 0" "4 0" "2 1" "3 1"; do
  set -- $cfg
  export ZKHIP_SLOTS=$1 ZKHIP_Z_GATE=$2
  echo "slots $1, z_gate $2"
  leg sha "slots $1 gate $2" --kind sha256 --log-domain 20
  leg poseidon "slots $1 gate $2" --curve bls12_381 --log-domain 18 --kind poseidon
  leg gm17 "slots $1 gate $2" --scheme gm17
done
