#!/bin/bash
# the three counter passes of the current build and nothing else (FETCH_SIZE, WRITE_SIZE, VALU: separate runs, kernel trace only,
# one-stream mode) -> gpurun_out/<tag>/pmc_traffic.json, pmc_valu.json (adopt with tools/adopt_evidence.py <tag>)
set -u
tag=${1:-pmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  ZKHIP_SERIAL=1 timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d "$out/prof_pmc_$ctr" -o pmc -- python "$root/bench.py" --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_pmc_$ctr.log" 2>&1
  echo "pmc $ctr rc=$?"
  db=$(find "$out/prof_pmc_$ctr" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python "$root/tools/pmc_stats.py" "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null
done
f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
[ -n "$f" ] && [ -n "$w" ] && python "$root/tools/pmc_traffic.py" "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0; profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null && echo "pmc_traffic.json written"
ZKHIP_SERIAL=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$out/prof_pmc_VALU" -o pmc -- \
  python "$root/bench.py" --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_pmc_VALU.log" 2>&1
echo "pmc VALU rc=$?"
db=$(find "$out/prof_pmc_VALU" -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python "$root/tools/pmc_valu.py" "$db" "$out/${tag}_pmc_VALU.md" | head -8 && cp "$out/${tag}_pmc_VALU.json" "$out/pmc_valu.json"
find "$out" -name "*.db" -delete
# and the default line of the same build on the same box (for the record next to the counters)
cd "$root" && unset ZKHIP_BENCH_CHILD && timeout 300 python bench.py --cpu-seconds 0 --e2e 0 > "$out/bench_default_no_cpu.json" 2> "$out/bench.err"
python - "$out/bench_default_no_cpu.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['phases_ms_serial']
print('default', round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2), 'ms single; serial g1/g2/ntt', round(s['kernel_msm_accum_g1_ms'],3), round(s['kernel_msm_accum_g2_ms'],3), round(s['kernel_ntt_ms'],3))
PY
