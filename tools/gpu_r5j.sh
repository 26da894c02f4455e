#!/bin/bash
# round 5, last GPU call (4.7 GPU-minutes left): the bench lines of the build with the bound key — the driver's command in full (CPU
# baseline, CLI legs, this build's counter files), a one-stream kernel table, then BASELINE.json's other configurations, each line
# carrying its own same-process figure with the key unbound (`bound_key.unbound_ms_per_step`).  Most important first, every step
# only if the budget (BUDGET_S of command time) still holds it.
set -u
tag=${1:-r5j}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
T0=$(date +%s)
BUDGET_S=${BUDGET_S:-235}
left() { echo $(( BUDGET_S - ( $(date +%s) - T0 ) )); }
fits() { [ "$(left)" -ge "$1" ] || { echo "skipped (needs $1 s, $(left) s left): $2"; return 1; }; }
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
line() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); c=d.get('cpu_baseline') or {}; b=d.get('bound_key') or {}; e=(d.get('roofline') or {}).get('offline_evidence') or {}
        print(' ', sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['ms_per_step'],3), 'ms/step | single', round(d['single_proof_ms'],2), '| bound', b.get('bound'), 'bind_ms', b.get('bind_ms') and round(b['bind_ms']),
              'unbound ms/step', b.get('unbound_ms_per_step') and round(b['unbound_ms_per_step'],3), 'unbound single', b.get('unbound_single_proof_ms') and round(b['unbound_single_proof_ms'],2),
              '| cpu', c.get('value'), c.get('gpu_proof_identical'), c.get('gpu_bound_key_proof_identical'), '| stale', e.get('stale'), '| attempts', len(d.get('attempts') or []))
PY
}
step "the driver's command"
timeout 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"; line "$out/bench_driver_command.json"
export TMPDIR=/tmp
step "one-stream kernel table, key bound"
fits 40 "serial table" && { ( cd /tmp && ZKHIP_BENCH_CHILD=1 ZKHIP_SERIAL=1 timeout 60 rocprofv3 --kernel-trace -d "$out/prof_serial" -o st -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --warmup 2 --serial-proofs 0 --e2e 0 --repeats 1 > "$out/prof_serial.log" 2>&1 )
  db=$(find "$out/prof_serial" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" "$out/${tag}_bound_serial_kernel_stats.md" > /dev/null 2>&1 && echo "  serial kernel table written"; find "$out" -name "*.db" -size +8M -delete; }
B="timeout 100 python bench.py --cpu-seconds 0 --e2e 0"
step "stdlib sha256"
fits 30 "sha256" && { $B --kind sha256 > "$out/bench_sha256_stdlib_2e20.json" 2>> "$out/bench.err"; line "$out/bench_sha256_stdlib_2e20.json"; }
step "Poseidon chain, BLS12-381"
fits 30 "poseidon" && { $B --curve bls12_381 --log-domain 18 --kind poseidon > "$out/bench_poseidon_bls12_381_2e18.json" 2>> "$out/bench.err"; line "$out/bench_poseidon_bls12_381_2e18.json"; }
step "2^22 constraints"
fits 60 "2^22" && { $B --log-domain 22 --steps 8 > "$out/bench_2e22.json" 2>> "$out/bench.err"; line "$out/bench_2e22.json"; }
step "the literal n = 2^20 (domain 2^21)"
fits 40 "2^21" && { $B --constraints 1048576 --steps 16 > "$out/bench_n2e20_literal_domain2e21.json" 2>> "$out/bench.err"; line "$out/bench_n2e20_literal_domain2e21.json"; }
step "sha-like witness"
fits 25 "sha-like" && { $B --kind sha > "$out/bench_sha_like.json" 2>> "$out/bench.err"; line "$out/bench_sha_like.json"; }
step "smoke"
fits 15 "smoke" && { timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"; }
tail -2 "$out/bench.err" 2>/dev/null
step "done"
