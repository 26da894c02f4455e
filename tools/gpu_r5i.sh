#!/bin/bash
# round 5, the session of the bound key (zkhip_pk_bind_r1cs), written for the 8.8 GPU-minutes the round had left: most important
# first, every step under its own timeout and only if the budget (BUDGET_S, default 330 s of command time) still holds it,
# everything written as it goes.  1. the bench line with the driver's flags (self-checking: bound proof == unbound proof == CPU
# baseline's; one region with the key unbound again), 2. the GPU tests that exercise the binding, 3. the counter passes of this build
# (key as loaded: --bind 0) and a kernel table of the bound pipeline, 4. the rest of the GPU suite if there is time.
set -u
tag=${1:-r5i}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
T0=$(date +%s)
BUDGET_S=${BUDGET_S:-330}
left() { echo $(( BUDGET_S - ( $(date +%s) - T0 ) )); }
fits() { [ "$(left)" -ge "$1" ] || { echo "skipped (needs $1 s, $(left) s left): $2"; return 1; }; }
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }

step "bench, the driver's flags"
timeout 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 4 --e2e 0 > "$out/bench_driver_flags_bound.json" 2> "$out/bench.err"
python - "$out/bench_driver_flags_bound.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); c=d.get('cpu_baseline') or {}
        print('bench:', round(d['value'],2), 'proofs/s', round(d['ms_per_step'],3), 'ms/step | regions', [round(x,3) for x in d['repeats']['ms_per_step']], '| single', round(d['single_proof_ms'],2),
              '| bound_key', d.get('bound_key'), '| cpu', c.get('value'), c.get('gpu_proof_identical'), c.get('gpu_bound_key_proof_identical'),
              '| ntt', {k: (d.get('roofline_ntt') or {}).get(k) for k in ('frac_serial','us_per_pass_serial')}, '| serial', d.get('phases_ms_serial'))
PY
tail -2 "$out/bench.err"
step "GPU tests of the binding"
fits 60 "bound tests" && { timeout $(( $(left) < 150 ? $(left) : 150 )) python -m pytest tests/test_gpu_parity.py tests/test_random_circuits.py -m gpu -q -x -k "prove_matches_oracle or full_size or random_systems or long_row" -p no:cacheprovider > "$out/pytest_bound.log" 2>&1
  echo "pytest(bound) rc=$?" >> "$out/pytest_bound.log"; tail -3 "$out/pytest_bound.log"; }
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
B="python $root/bench.py --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0"
pmc() {   # name, counters...
  local name=$1; shift
  ( cd /tmp && ZKHIP_SERIAL=1 timeout 100 rocprofv3 --pmc "$@" --kernel-trace -d "$out/prof_pmc_$name" -o pmc -- $B > "$out/prof_pmc_$name.log" 2>&1 )
}
step "counter pass: VALU"
fits 60 "VALU pass" && { pmc VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  db=$(find "$out/prof_pmc_VALU" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/pmc_valu.py "$db" "$out/${tag}_pmc_VALU.md" > /dev/null && cp "$out/${tag}_pmc_VALU.json" "$out/pmc_valu.json" && echo "pmc_valu.json written"; }
step "kernel table of the bound pipeline"
fits 60 "kernel table" && { ( cd /tmp && timeout 100 rocprofv3 --kernel-trace -d "$out/prof_stats" -o st -- python "$root/bench.py" --cpu-seconds 0 --steps 20 --warmup 5 --serial-proofs 0 --e2e 0 --repeats 1 > "$out/prof_stats.log" 2>&1 )
  db=$(find "$out/prof_stats" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" "$out/${tag}_bound_pipelined_kernel_stats.md" > /dev/null 2>&1 && echo "kernel table written"; }
step "counter passes: FETCH_SIZE, WRITE_SIZE"
fits 110 "traffic passes" && { for ctr in FETCH_SIZE WRITE_SIZE; do pmc $ctr $ctr
    db=$(find "$out/prof_pmc_$ctr" -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && python tools/pmc_stats.py "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null; done
  f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then
    python tools/pmc_traffic.py "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0; profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null && echo "pmc_traffic.json written"
  else echo "PMC traffic passes incomplete"; fi; }
find "$out" -name "*.db" -size +8M -delete
step "the rest of the GPU suite"
fits 120 "rest of the suite" && { timeout $(left) python -m pytest tests/ -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_full_size_properties > "$out/pytest_gpu_rest.log" 2>&1
  echo "pytest(all) rc=$?" >> "$out/pytest_gpu_rest.log"; tail -3 "$out/pytest_gpu_rest.log"; }
step "done"
