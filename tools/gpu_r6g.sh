#!/bin/bash
# round 6, session g: where do the transform passes' cycles go?  (LDS conflicts? barriers? issue?)
set -u
tag=${1:-r6g}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TCP_[A-Z0-9_]+|SQC_[A-Z0-9_]+)\b" | sort -u > "$out/counters.txt" )
grep -E "LDS|BARRIER|WAIT|LEVEL_WAVES|OCCUP" "$out/counters.txt" | tr '\n' ' '; echo
pmc() { local name=$1; shift; ( cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace -d "$out/prof_$name" -o pmc -- python "$root/tools/ntt_probe.py" ntt_skew_us 0 > "$out/prof_$name.log" 2>&1 ); 
  db=$(find "$out/prof_$name" -name "*.db" | head -1); [ -n "$db" ] && python - "$db" "$name" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
agg = {}
for k, c, v, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    if "k_ntt_" not in k: continue
    kk = "cols" if "k_ntt_cols" in k else "rows"
    e = agg.setdefault(kk, {}).setdefault(c, [0, 0.0, 0.0]); e[0] += 1; e[1] += v; e[2] += d
for kk, cs in agg.items():
    print(sys.argv[2], kk, {c: round(e[1] / e[0]) for c, e in cs.items()}, "avg_us", round(next(iter(cs.values()))[2] / next(iter(cs.values()))[0] / 1e3, 1))
PY
}
pmc A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pmc B SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
pmc C SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE
find "$out" -name "*.db" -size +8M -delete
