#!/bin/bash
# round 5, the last GPU seconds: the build with the lone-proof G2 head start (ZKHIP_G2_HEAD_START, session r5n) — its counter passes
# (the sources' fingerprint moved), the GPU tests that touch BLS12-381 and the binding, the Poseidon and the headline bench lines.
set -u
tag=${1:-r5o}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
T0=$(date +%s)
BUDGET_S=${BUDGET_S:-85}
left() { echo $(( BUDGET_S - ( $(date +%s) - T0 ) )); }
fits() { [ "$(left)" -ge "$1" ] || { echo "skipped (needs $1 s, $(left) s left): $2"; return 1; }; }
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
B="python $root/bench.py --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0"
pmc() { local name=$1; shift; ( cd /tmp && ZKHIP_SERIAL=1 timeout 40 rocprofv3 --pmc "$@" --kernel-trace -d "$out/prof_pmc_$name" -o pmc -- $B > "$out/prof_pmc_$name.log" 2>&1 ); }
step "counter passes"
pmc VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
db=$(find "$out/prof_pmc_VALU" -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python tools/pmc_valu.py "$db" "$out/${tag}_pmc_VALU.md" > /dev/null && cp "$out/${tag}_pmc_VALU.json" "$out/pmc_valu.json" && echo "pmc_valu.json written"
for ctr in FETCH_SIZE WRITE_SIZE; do pmc $ctr $ctr
  db=$(find "$out/prof_pmc_$ctr" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/pmc_stats.py "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null; done
f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_traffic.py "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0; profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null && echo "pmc_traffic.json written"
find "$out" -name "*.db" -size +8M -delete
unset ZKHIP_BENCH_CHILD
step "GPU tests: BLS12-381, the binding, the head start"
fits 25 "tests" && { timeout $(( $(left) < 60 ? $(left) : 60 )) python -m pytest tests/test_gpu_parity.py tests/test_random_circuits.py tests/test_poseidon.py -m gpu -q -x -p no:cacheprovider \
    -k "lone_proofs or (prove_matches_oracle and bls12_381) or (prove_matches_oracle and 10) or random_systems or poseidon" > "$out/pytest_head_start.log" 2>&1
  echo "pytest rc=$?" >> "$out/pytest_head_start.log"; tail -3 "$out/pytest_head_start.log"; }
step "Poseidon chain, BLS12-381"
fits 12 "poseidon bench" && { timeout 40 python bench.py --cpu-seconds 0 --e2e 0 --curve bls12_381 --log-domain 18 --kind poseidon > "$out/bench_poseidon_bls12_381_2e18.json" 2>> "$out/bench.err"
  python - "$out/bench_poseidon_bls12_381_2e18.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); b=d['bound_key']
        print('  poseidon', round(d['value'],2), 'proofs/s | single bound', round(d['single_proof_ms'],2), 'unbound', round(b['unbound_single_proof_ms'],2), '| unbound ms/step', round(b['unbound_ms_per_step'],3), 'bound', round(d['ms_per_step'],3))
PY
}
step "the driver's flags"
fits 10 "headline" && { timeout 40 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --e2e 0 > "$out/bench_driver_flags.json" 2>> "$out/bench.err"
  python - "$out/bench_driver_flags.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); b=d['bound_key']
        print('  headline', round(d['value'],2), 'proofs/s', [round(x,3) for x in d['repeats']['ms_per_step']], '| unbound', round(b['unbound_ms_per_step'],3), '| single', round(d['single_proof_ms'],2), round(b['unbound_single_proof_ms'],2))
PY
}
step "done"
