#!/bin/bash
# one PMC pass with the VALU counters over the one-stream bench run (kernel trace only): bash tools/gpu_pmc_valu.sh <tag>
set -u
tag=${1:-valu}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
[ -n "${WITH_SMOKE:-}" ] && { timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; exit 0; }; }
cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
ZKHIP_SERIAL=1 timeout ${PMC_TIMEOUT:-130} rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$out/prof" -o pmc -- \
  python "$root/bench.py" --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof.log" 2>&1
echo "rocprofv3 rc=$?"
db=$(find "$out/prof" -name "*.db" 2>/dev/null | head -1)
[ -n "$db" ] && python "$root/tools/pmc_valu.py" "$db" "$out/${tag}_pmc_VALU.md" | head -12 || { grep -v "^W2026" "$out/prof.log" | tail -5; }
find "$out" -name "*.db" -size +8M -delete
