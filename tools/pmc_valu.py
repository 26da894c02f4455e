#!/usr/bin/env python3
"""VALU issue occupation per kernel from one rocprofv3 PMC pass (--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE --kernel-trace; on gfx950 the two SQ VALU counters report the same number: wave-instructions).
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles = GRBM_GUI_ACTIVE / 8 (the clock that results, cycles / duration, is
the check: ~1.9 GHz under this load); a 16-lane SIMD issues one wave64 VALU instruction per 4 cycles at best, so
  cycles per instruction per SIMD = cycles * SIMDs / SQ_INSTS_VALU        and        issue utilisation = 4 / that.
usage: pmc_valu.py results.db out.md [simds=1024] [xcds=8]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"zk::Fe2<zk::(\w+)>", r"\1^2", name)
    name = re.sub(r"zk::Fe<zk::(\w+)>", r"\1", name)
    name = re.sub(r"zk::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return re.sub(r"^void ", "", name)


db = sqlite3.connect(sys.argv[1])
simds = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
xcds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
agg = {}
for k, c, v, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    e = agg.setdefault(short(k), {}).setdefault(c, [0, 0.0, 0.0])
    e[0] += 1; e[1] += v; e[2] += d
rows = []
for k, cs in agg.items():
    if "SQ_INSTS_VALU" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    n = cs["SQ_INSTS_VALU"][0]
    dur_us = cs["SQ_INSTS_VALU"][2] / n / 1e3
    insts = cs["SQ_INSTS_VALU"][1] / n
    cycles = cs["GRBM_GUI_ACTIVE"][1] / cs["GRBM_GUI_ACTIVE"][0] / xcds
    cpi = cycles * simds / insts if insts else float("inf")
    rows.append((dur_us * n, k, n, dur_us, insts, cycles, cycles / dur_us / 1e3, cpi, 100.0 * 4 / cpi))
lines = ["| kernel | launches | avg us | VALU wave-instructions per launch | GPU cycles per launch | clock GHz | cycles per VALU instruction per SIMD | VALU issue utilisation % |",
         "|---|---|---|---|---|---|---|---|"]
for _, k, n, dur, insts, cyc, ghz, cpi, util in sorted(rows, reverse=True)[:20]:
    lines.append("| `%s` | %d | %.1f | %.3g | %.3g | %.2f | %.2f | %.0f |" % (k, n, dur, insts, cyc, ghz, cpi, util))
import json
js = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0 (tools/gpu_pmc_valu.sh)",
      "definition": "cycles = GRBM_GUI_ACTIVE / %d XCDs; cycles_per_valu_instruction_per_simd = cycles * %d SIMDs / SQ_INSTS_VALU; issue_utilisation = 4 / that (one wave64 VALU instruction per 4 cycles per 16-lane SIMD)" % (xcds, simds)}
for _, k, n, dur, insts, cyc, ghz, cpi, util in rows:
    for tag, needle in (("G1", "k_msm_accum<Fu<"), ("G2", "k_msm_accum<Fu2<"), ("NTT_cols", "k_ntt_cols"), ("NTT_rows", "k_ntt_rows")):
        if k.startswith(needle) and "Bn254" in k:
            js[tag] = {"launches": n, "avg_us": dur, "valu_wave_instructions_per_launch": insts, "gpu_cycles_per_launch": cyc, "clock_ghz": ghz,
                       "cycles_per_valu_instruction_per_simd": cpi, "issue_utilisation": 4 / cpi}
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zokrates_amd.build import csrc_hash  # noqa: E402
js["csrc_hash"] = csrc_hash()      # the sources this pass ran on (bench.py refuses the figures next to another build)
if len(sys.argv) > 2:
    json.dump(js, open(re.sub(r"\.md$", "", sys.argv[2]) + ".json", "w"), indent=1)
text = "\n".join(lines)
print(text)
open(sys.argv[2], "w").write(text + "\n")
