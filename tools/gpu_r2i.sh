#!/bin/bash
# round-2 GPU session I: NTT pipeline on its own stream; shared vs per-slot lane streams again; timeline
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2i
mkdir -p "$out"
cd "$root"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gm17.py -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
B="timeout 300 python bench.py --cpu-seconds 0 --steps 32"
run() { name=$1; shift; env "$@" $B >> "$out/bench_$name.json" 2>> "$out/bench.err"; }
run perslot X=1
run shared ZKHIP_SHARED_LANE_STREAMS=1
run perslot_s4 ZKHIP_SLOTS=4
run shared_s4 ZKHIP_SHARED_LANE_STREAMS=1 ZKHIP_SLOTS=4
run perslot_s2 ZKHIP_SLOTS=2
run perslot X=1
run shared ZKHIP_SHARED_LANE_STREAMS=1
$B --scheme gm17 > "$out/bench_gm17.json" 2>> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_p" -o p -- python "$root/bench.py" --cpu-seconds 0 --steps 24 --serial-proofs 0 > "$out/prof_p.log" 2>&1
db=$(find "$out/prof_p" -name "*.db" | head -1)
[ -n "$db" ] && python "$root/tools/timeline.py" "$db" 0.6 > "$out/timeline.txt" 2>&1
find "$out/prof_p" -name "*.db" -size +20M -delete
cd "$root"
tail -3 "$out/pytest_gpu.log"
for f in perslot shared perslot_s4 shared_s4 perslot_s2 gm17; do python - "$out/bench_$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single', round(d['single_proof_from_host_ms'],2), 'from host; serial g1/g2', round(s.get('kernel_msm_accum_g1_ms',0),3), round(s.get('kernel_msm_accum_g2_ms',0),3), 'pipelined ntt', round(d['phases_ms']['kernel_ntt_ms'],2))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat "$out/timeline.txt"; tail -3 "$out/bench.err"
