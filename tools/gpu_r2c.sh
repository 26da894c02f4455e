#!/bin/bash
# round-2 GPU session C: NTT v2 (9x29 limbs, radix-4 register butterflies, 38 KiB tiles) — parity, columns per tile A/B,
# hardware-queue experiment, pipelined timeline, round-1 library on the same box
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2c
mkdir -p "$out"
cd "$root"
(ZKHIP_PKG=zokrates_amd_v1 timeout 300 python tools/repro_partial_records.py; timeout 300 python tools/repro_partial_records.py) > "$out/repro.log" 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 16"
$B > "$out/bench_q8.json" 2> "$out/bench.err"
GPU_MAX_HW_QUEUES=4 $B > "$out/bench_q4.json" 2>> "$out/bench.err"
GPU_MAX_HW_QUEUES=16 $B > "$out/bench_q16.json" 2>> "$out/bench.err"
ZKHIP_NTT_COLS=1 $B > "$out/bench_cols1.json" 2>> "$out/bench.err"
ZKHIP_NTT_COLS=4 $B > "$out/bench_cols4.json" 2>> "$out/bench.err"
ZKHIP_PKG=zokrates_amd_v1 timeout 300 python <(git show 75d9233:bench.py 2>/dev/null || cat bench.py) --cpu-seconds 0 --steps 16 > "$out/bench_round1_lib.json" 2>> "$out/bench.err"
$B --scheme gm17 > "$out/bench_gm17.json" 2>> "$out/bench.err"
$B --curve bls12_381 --log-domain 18 --kind poseidon > "$out/bench_poseidon_bls.json" 2>> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
for tag in serial pipelined; do
  if [ $tag = serial ]; then export ZKHIP_SERIAL=1; else unset ZKHIP_SERIAL; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o $tag -- python "$root/bench.py" --cpu-seconds 0 --steps 16 --serial-proofs 0 > "$out/prof_$tag.log" 2>&1
  db=$(find "$out/prof_$tag" -name "*.db" | head -1)
  [ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_kernel_stats.md" > /dev/null
  [ -n "$db" ] && python "$root/tools/timeline.py" "$db" 0.6 > "$out/${tag}_timeline.txt" 2>&1
  find "$out/prof_$tag" -name "*.db" -size +20M -delete
done
unset ZKHIP_SERIAL
cd "$root"
tail -3 "$out/repro.log"; tail -4 "$out/pytest_gpu.log"
for f in bench_q8 bench_q4 bench_q16 bench_cols1 bench_cols4 bench_round1_lib bench_gm17 bench_poseidon_bls; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get('roofline_ntt') or {}
    print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d.get('single_proof_from_host_ms',0),2), 'from host; ntt us/pass', r.get('us_per_pass'), r.get('us_per_pass_serial'), 'frac_serial', r.get('frac_serial'), {k:round(v,2) for k,v in d['phases_ms'].items()})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
head -24 "$out/serial_kernel_stats.md"; cat "$out/pipelined_timeline.txt"
tail -5 "$out/bench.err"
