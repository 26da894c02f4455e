#!/bin/bash
# round-2 GPU session D: NTT twiddle plans, proofs in flight 2/3/4
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2d
mkdir -p "$out"
cd "$root"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$out/pytest_gpu_parity.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu_parity.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 24"
ZKHIP_SLOTS=2 $B > "$out/bench_s2.json" 2> "$out/bench.err"
ZKHIP_SLOTS=3 $B > "$out/bench_s3.json" 2>> "$out/bench.err"
ZKHIP_SLOTS=4 $B > "$out/bench_s4.json" 2>> "$out/bench.err"
ZKHIP_SLOTS=3 ZKHIP_NTT_COLS=4 $B > "$out/bench_s3_cols4.json" 2>> "$out/bench.err"
ZKHIP_SLOTS=3 ZKHIP_NTT_COLS=1 $B > "$out/bench_s3_cols1.json" 2>> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
ZKHIP_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/bench.py" --cpu-seconds 0 --steps 8 --serial-proofs 0 > "$out/prof_serial.log" 2>&1
db=$(find "$out/prof_serial" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/serial_kernel_stats.md" > /dev/null
find "$out/prof_serial" -name "*.db" -size +20M -delete
ZKHIP_SLOTS=3 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_s3" -o s3 -- python "$root/bench.py" --cpu-seconds 0 --steps 24 --serial-proofs 0 > "$out/prof_s3.log" 2>&1
db=$(find "$out/prof_s3" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/timeline.py" "$db" 0.6 > "$out/s3_timeline.txt" 2>&1
find "$out/prof_s3" -name "*.db" -size +20M -delete
cd "$root"
tail -3 "$out/pytest_gpu_parity.log"
for f in bench_s2 bench_s3 bench_s4 bench_s3_cols4 bench_s3_cols1; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get('roofline_ntt') or {}
    print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; ntt us/pass serial', r.get('us_per_pass_serial'), 'frac_serial', r.get('frac_serial'), 'accum serial g1/g2', d['phases_ms_serial']['kernel_msm_accum_g1_ms'], d['phases_ms_serial']['kernel_msm_accum_g2_ms'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
grep -E "k_ntt|k_msm_accum|fold_final|k_quotient|k_matvec" "$out/serial_kernel_stats.md"; cat "$out/s3_timeline.txt"
tail -5 "$out/bench.err"
