#!/bin/bash
# round 6, session w: the stream plans on the OTHER configurations (GM17 2^20, Poseidon chain on BLS12-381, stdlib SHA-256 2^20):
# bench.py's config legs, one process per (leg, plan), bound and unbound proofs/s and the lone proof.
set -u
tag=${1:-r6w}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
leg() {  # leg <name> <plan> <bench args...>
  local name=$1 plan=$2; shift 2
  ZKHIP_PIPES="$plan" timeout 120 python3 bench.py --steps ${STEPS:-16} --warmup 6 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 2 --oracle none --configs 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  python - "$out/$name.json" "$name" "$plan" <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); ok=True
        print('   %-10s %-52s bound %7.2f (regions %s)  unbound %7.2f  lone %6.2f / %6.2f' % (sys.argv[2], sys.argv[3], d['value_bound'] or 0, [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['value_unbound'] or 0, d['single_proof_ms'], d.get('single_proof_unbound_ms') or 0))
if not ok: print('   ', sys.argv[2], sys.argv[3], 'NO LINE')
PY
}
R="M=0,N=2,O=1,G0=1,Z0=3,H0=0,G1=2,Z1=3,H1=0,G2=1,Z2=2,H2=3"
for plan in "-" "$R" "$R,n=3" "$R,gl=3" "$R,n=3,gl=3" "$R,n=3,gl=2" "$R,n=1,gl=3" "-"; do
  step "plan '$plan'"
  leg dense "$plan"
  leg sha "$plan" --kind sha256 --log-domain 20
  leg poseidon "$plan" --curve bls12_381 --log-domain 18 --kind poseidon
  leg gm17 "$plan" --scheme gm17
done
export ZKHIP_LONE_SCHED=4
for plan in "-" "$R,n=3,gl=3"; do
  step "ZKHIP_LONE_SCHED=4, plan '$plan'"
  leg dense "$plan"
  leg sha "$plan" --kind sha256 --log-domain 20
  leg poseidon "$plan" --curve bls12_381 --log-domain 18 --kind poseidon
done
step "done"
