#!/bin/bash
# round-2 GPU session B: MSM v2 (precomputed window multiples, one bucket set) — parity, same-box A/B against the round-1
# library, serial-mode kernel table
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2b
mkdir -p "$out"
cd "$root"
(ZKHIP_LIBRARY=$root/zokrates_amd_v1/libzkhip.so timeout 300 python tools/repro_partial_records.py; timeout 300 python tools/repro_partial_records.py) > "$out/repro.log" 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 300 python bench.py --cpu-seconds 0 --steps 16 > "$out/bench_v2.json" 2> "$out/bench_v2.err"
ZKHIP_MSM_C=15 timeout 300 python bench.py --cpu-seconds 0 --steps 16 > "$out/bench_v2_c15.json" 2>> "$out/bench_v2.err"
ZKHIP_MSM_C=14 timeout 300 python bench.py --cpu-seconds 0 --steps 16 > "$out/bench_v2_c14.json" 2>> "$out/bench_v2.err"
ZKHIP_MSM_WAVES=4 timeout 300 python bench.py --cpu-seconds 0 --steps 16 > "$out/bench_v2_w4.json" 2>> "$out/bench_v2.err"
ZKHIP_MSM_WAVES=2 timeout 300 python bench.py --cpu-seconds 0 --steps 16 > "$out/bench_v2_w2.json" 2>> "$out/bench_v2.err"
timeout 300 python bench.py --cpu-seconds 0 --steps 16 --scheme gm17 > "$out/bench_v2_gm17.json" 2>> "$out/bench_v2.err"
cd /tmp && export TMPDIR=/tmp
ZKHIP_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_serial" -o serial -- python "$root/bench.py" --cpu-seconds 0 > "$out/prof_serial.log" 2>&1
db=$(find "$out/prof_serial" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/serial_kernel_stats.md" > /dev/null
find "$out/prof_serial" -name "*.db" -size +20M -delete
cd "$root"
tail -3 "$out/repro.log"; tail -4 "$out/pytest_gpu.log"
for f in bench_v2 bench_v2_c15 bench_v2_c14 bench_v2_w4 bench_v2_w2 bench_v2_gm17; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', {k:round(v,2) for k,v in d['phases_ms'].items()}, d['host_ms'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -30 "$out/serial_kernel_stats.md"
