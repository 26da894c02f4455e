#!/bin/bash
# round 6, session r7m: k_scan_one with the next tile's counters fetched before this tile's scan and 16-byte stores: parity of everything that
# sorts, the kernel's duration in a one-stream trace (44 us before), lone proofs of the four workloads
set -u
tag=${1:-r7m}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gm17.py tests/test_sha256_circuit.py tests/test_poseidon.py tests/test_gpu_bound.py -m gpu -q -x -p no:cacheprovider > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -3 "$out/pytest.log"
step "one-stream trace"
( cd /tmp && ZKHIP_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --serial-proofs 0 --e2e 0 --configs 0 --oracle none > "$out/prof_serial.log" 2>&1 )
db=$(find "$out/prof_serial" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" "$out/${tag}_g16_serial_kernel_stats.md" > /dev/null
grep -h "k_scan_one\|k_msm_count\|k_msm_digits\|k_msm_part" "$out/${tag}_g16_serial_kernel_stats.md" | cut -c1-120
find "$out" -name "*.db" -size +8M -delete
step "lone proofs and batches, four workloads"
for w in "--kind dense --log-domain 20" "--kind sha256 --log-domain 20" "--kind poseidon --curve bls12_381 --log-domain 18" "--scheme gm17 --log-domain 20"; do
  for rep in 1 2; do
    timeout 200 python3 bench.py $w --steps 24 --warmup 5 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 2 --oracle trapdoor --configs 0 2> /dev/null |
      python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  %-50s %.1f proofs/s (regions %s)  lone %.2f ms  oracle %s' % ('$w', d['value'], [round(1000/x,1) for x in d['repeats']['ms_per_step']], d['single_proof_ms'], d.get('identical_to_oracle')))"
  done
done
step "done"
