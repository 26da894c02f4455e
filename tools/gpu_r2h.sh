#!/bin/bash
# round-2 GPU session H: per-slot lane streams vs shared, hardware queues, accumulation waves, G2 WPE=1 (no scratch)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2h
mkdir -p "$out"
cd "$root"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 32"
run() { name=$1; shift; env "$@" $B >> "$out/bench_$name.json" 2>> "$out/bench.err"; }
run perslot_q8 X=1
run shared_q8 ZKHIP_SHARED_LANE_STREAMS=1
run perslot_q16 GPU_MAX_HW_QUEUES=16
run perslot_q24 GPU_MAX_HW_QUEUES=24
run perslot_q8_s4 ZKHIP_SLOTS=4
run perslot_q16_s4 ZKHIP_SLOTS=4 GPU_MAX_HW_QUEUES=16
run perslot_q8_w5 ZKHIP_MSM_WAVES=5
run perslot_q8_g2wpe1 ZKHIP_LIBRARY=$root/zokrates_amd_v2B/libzkhip.so
run perslot_q8 X=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_p" -o p -- python "$root/bench.py" --cpu-seconds 0 --steps 24 --serial-proofs 0 > "$out/prof_p.log" 2>&1
db=$(find "$out/prof_p" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/timeline.py" "$db" 0.6 > "$out/perslot_timeline.txt" 2>&1
find "$out/prof_p" -name "*.db" -size +20M -delete
cd "$root"
tail -3 "$out/pytest_gpu.log"
for f in perslot_q8 shared_q8 perslot_q16 perslot_q24 perslot_q8_s4 perslot_q16_s4 perslot_q8_w5 perslot_q8_g2wpe1; do python - "$out/bench_$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d['single_proof_from_host_ms'],2), 'from host; serial g1/g2', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat "$out/perslot_timeline.txt"; tail -3 "$out/bench.err"
