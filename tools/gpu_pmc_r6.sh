#!/bin/bash
# The counter passes of a build (round 6): rocprofv3 --pmc in runs of their own (kernel trace only), one stream (ZKHIP_SERIAL=1), the
# key BOUND as in the bench line (`--bind 2`: 8 transform pass-vectors per proof, every transform launch over two vectors).
#   VALU  x 2 (the same box, back to back: do the two agree?)   -> <tag>_pmc_VALU{,_2}.md, pmc_valu.json
#   FETCH_SIZE, WRITE_SIZE                                         -> <tag>_pmc_{FETCH,WRITE}_SIZE.md, pmc_traffic.json (calibrated per access pattern)
# usage: bash tools/gpu_pmc_r6.sh <tag>   (writes under gpurun_out/<tag>/; copy the summaries into profiles/ by hand)
set -u
tag=${1:-r6pmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1 ZKHIP_PMC_PASSES=8 ZKHIP_NTT_COLS_PATTERN=${ZKHIP_NTT_COLS_PATTERN:-seg64}
B="python $root/bench.py --bind 2 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 --configs 0 --repeats 1"
pmc() { local name=$1; shift; ( cd /tmp && ZKHIP_SERIAL=1 timeout 90 rocprofv3 --pmc "$@" --kernel-trace -d "$out/prof_pmc_$name" -o pmc -- $B > "$out/prof_pmc_$name.log" 2>&1 ); }
for pass in VALU VALU_2; do
  pmc $pass SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  db=$(find "$out/prof_pmc_$pass" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/pmc_valu.py "$db" "$out/${tag}_pmc_$pass.md" > /dev/null
  grep -h '"shader_clock"' "$out/prof_pmc_$pass.log" > /dev/null 2>&1
done
[ -f "$out/${tag}_pmc_VALU.json" ] && cp "$out/${tag}_pmc_VALU.json" "$out/pmc_valu.json" && echo "pmc_valu.json written"
for ctr in FETCH_SIZE WRITE_SIZE; do pmc $ctr $ctr
  db=$(find "$out/prof_pmc_$ctr" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/pmc_stats.py "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null; done
f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_traffic.py "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --bind 2 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0 (key bound: 8 pass-vectors per proof); profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null && echo "pmc_traffic.json written"
find "$out" -name "*.db" -size +8M -delete
python - "$out" "$tag" <<'PY'
import json, sys, os
out, tag = sys.argv[1], sys.argv[2]
for name in ("VALU", "VALU_2"):
    p = os.path.join(out, "%s_pmc_%s.json" % (tag, name))
    if os.path.exists(p):
        d = json.load(open(p))
        print(name, {k: {"us": round(v["avg_us"], 1), "instr": round(v["valu_wave_instructions_per_launch"] / 1e6, 1), "derived_ghz": round(v["clock_ghz"], 3), "util_at_derived_clock": round(v["issue_utilisation"], 3)} for k, v in d.items() if isinstance(v, dict) and "avg_us" in v})
p = os.path.join(out, "pmc_traffic.json")
if os.path.exists(p):
    d = json.load(open(p))
    print({k: {a: b for a, b in v.items() if a in ("traffic_bytes_per_launch", "traffic_bytes_per_pass", "fetch_factor", "pattern", "patterns", "algorithmic_bytes_per_pass")} for k, v in d.items() if isinstance(v, dict) and k in ("G1", "G2", "NTT")})
PY
