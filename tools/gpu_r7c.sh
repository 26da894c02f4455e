#!/bin/bash
# round 6, session 7c: rows of more than 512 terms on a workgroup each (k_matvec_huge) against a wavefront each (ZKHIP_MATVEC_HUGE=0) —
# stdlib SHA-256 2^20 (rows of 7 041 terms), the sha-like witness, Poseidon; parity of the mat-vec tests first.
set -u
tag=${1:-r7c}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
timeout 600 python -m pytest tests/test_random_circuits.py tests/test_sha256_circuit.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > "$out/pytest.log" 2>&1; tail -2 "$out/pytest.log"
leg() {  # leg <name> <label> <bench args...>
  local name=$1 label=$2; shift 2
  timeout 120 python3 bench.py --steps 32 --warmup 6 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 3 --repeats 3 --oracle trapdoor --configs 0 "$@" > "$out/$name.json" 2> "$out/$name.err"
  python - "$out/$name.json" "$name" "$label" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('   %-10s %-10s bound %7.2f (regions %s) unbound %7.2f lone %6.2f / %6.2f identical %s | one-stream ntt_ms %.3f total %.2f' % (sys.argv[2], sys.argv[3], d['value'], [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['value_unbound'] or 0, d['single_proof_ms'], d.get('single_proof_unbound_ms') or 0, d.get('identical_to_oracle'), d['phases_ms_serial']['ntt_ms'], d['phases_ms_serial']['total_ms']))
PY
}
for h in 1 0 1 0; do
  export ZKHIP_MATVEC_HUGE=$h
  leg sha "huge=$h" --kind sha256 --log-domain 20
  leg shalike "huge=$h" --kind sha --log-domain 20
done
unset ZKHIP_MATVEC_HUGE
( cd /tmp && ZKHIP_SERIAL=1 timeout 120 rocprofv3 --kernel-trace --stats -d "$out/prof_sha" -o sha -- python "$root/bench.py" --steps 8 --warmup 2 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 1 --oracle none --configs 0 --bind 2 --kind sha256 --log-domain 20 > "$out/prof_sha.log" 2>&1 )
db=$(find "$out/prof_sha" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" "$out/${tag}_sha256_serial_kernel_stats.md" > /dev/null; grep -E "matvec|ntt_|accum" "$out/${tag}_sha256_serial_kernel_stats.md" | cut -c1-120
find "$out" -name "*.db" -size +8M -delete
