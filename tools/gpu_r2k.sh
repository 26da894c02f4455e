#!/bin/bash
# round-2 GPU session K: fold kernels with one add site (code fits the instruction cache) vs the previous build
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2k
mkdir -p "$out"
cd "$root"
# a sick box (memory access faults on the first kernel) costs ten minutes of core dumps: probe first
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; tail -5 "$out/smoke.log"; exit 0; }
timeout 1500 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 32"
run() { name=$1; shift; env "$@" $B >> "$out/bench_$name.json" 2>> "$out/bench.err"; }
run new X=1
run prev ZKHIP_LIBRARY=$root/zokrates_amd_v2P/libzkhip.so
run new X=1
run prev ZKHIP_LIBRARY=$root/zokrates_amd_v2P/libzkhip.so
$B --scheme gm17 > "$out/bench_gm17.json" 2>> "$out/bench.err"
$B --curve bls12_381 --log-domain 18 --kind poseidon > "$out/bench_poseidon_bls.json" 2>> "$out/bench.err"
ZKHIP_LIBRARY=$root/zokrates_amd_v2P/libzkhip.so $B --curve bls12_381 --log-domain 18 --kind poseidon > "$out/bench_poseidon_bls_prev.json" 2>> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
ZKHIP_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --serial-proofs 0 > "$out/prof_serial.log" 2>&1
db=$(find "$out/prof_serial" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/serial_kernel_stats.md" > /dev/null
find "$out/prof_serial" -name "*.db" -size +8M -delete
cd "$root"
tail -3 "$out/pytest_gpu.log"
for f in new prev gm17 poseidon_bls poseidon_bls_prev; do python - "$out/bench_$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d['single_proof_from_host_ms'],2), 'from host; serial total', round(s.get('total_ms',0),2))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
grep -E "fold|heavy" "$out/serial_kernel_stats.md"; tail -3 "$out/bench.err"
