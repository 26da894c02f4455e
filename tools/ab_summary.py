#!/usr/bin/env python3
"""One line per bench.py run of a same-box A/B session (gpurun_out/<tag>/bench_<name>.json, one JSON line per run):
    tools/ab_summary.py gpurun_out/r5a [title]   -> text for profiles/"""
import glob
import json
import os
import sys

d = sys.argv[1]
if len(sys.argv) > 2:
    print(sys.argv[2])
print("%-24s %10s %10s %9s | one stream: %8s %7s %7s %6s | %6s %6s" % ("run", "proofs/s", "ms/step", "single ms", "total ms", "G1 acc", "G2 acc", "NTT", "sclk", "W"))
for f in sorted(glob.glob(os.path.join(d, "bench_*.json")), key=os.path.getmtime):
    name = os.path.basename(f)[6:-5]
    for line in open(f):
        if not line.startswith("{"):
            continue
        try:
            j = json.loads(line)
        except Exception:
            continue
        if j.get("value") is None:
            continue
        s = j.get("phases_ms_serial") or {}
        u = j.get("under_load") or {}
        rp = j.get("repeats") or {}
        ms = rp.get("median_ms_per_step", j["ms_per_step"])
        print("%-24s %10.2f %10.3f %9.2f | %20.2f %7.3f %7.3f %6.3f | %6.0f %6.0f" % (
            name, j["value"], ms, j.get("single_proof_ms", 0), s.get("total_ms", 0), s.get("kernel_msm_accum_g1_ms", 0), s.get("kernel_msm_accum_g2_ms", 0),
            s.get("kernel_ntt_ms", 0), (u.get("sclk_mhz") or {}).get("mean", 0), (u.get("power_w") or {}).get("mean", 0)))
