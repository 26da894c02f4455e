#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from two rocprofv3 PMC passes of the SAME
command (`--pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`, separate runs as MI355X_MICROARCH.md
prescribes; FETCH_SIZE doubled: gfx950 tallies 128-byte requests as 64 bytes).
usage: pmc_traffic.py fetch.db write.db out.json <source note>"""
import json
import os
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    agg = {}
    for k, v in db.execute("select kernel_name, value from counters_collection"):
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += v
    return agg


def pick(agg, *needles):
    n = tot = 0
    for k, (cnt, val) in agg.items():
        if all(s in k for s in needles):
            n += cnt
            tot += val
    return n, tot


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
# FETCH_SIZE = 64 bytes per L2 -> fabric read request; a request is 64 OR 128 bytes depending on the access pattern, so the factor
# "true bytes per reported byte" is a property of the pattern — measured with tools/fetch_calib.hip (known byte counts, tables far
# beyond the Infinity Cache; profiles/r6*_fetch_calibration.md), passed in as argv[5] (a JSON object pattern -> factor) or defaulted
# to round 6's readings:
CAL = {"stream16": 2.0, "stream4": 2.0, "gather64": 1.0, "gather128": 2.0, "gather32": 0.5, "pair16": 2.0, "seg64": 1.0, "seg128": 2.0}
if len(sys.argv) > 5:
    CAL.update(json.load(open(sys.argv[5])))
out = {"source": sys.argv[4] if len(sys.argv) > 4 else "",
       "calibration": {"factors": CAL, "unit": "true bytes fetched per byte FETCH_SIZE reports, per access pattern",
                       "how": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE: requested bytes / reported bytes; MI355X_MICROARCH.md's x2 is the stream16 row — "
                              "a 64-byte gather by one lane (a BN254 G1 base) is reported at its true size"},
       "correction": "per kernel: fetch_kb_raw x fetch_factor(pattern) + write_kb_raw (WRITE_SIZE as reported); counters are in KB"}
PATTERN = {"G1": "gather64", "G2": "gather128"}      # one packed BN254 point per lane and step: 64 / 128 bytes
for tag in ("G1", "G2"):
    nf, f = pick({k: v for k, v in fetch.items() if ("Fu2<" in k) == (tag == "G2")}, "k_msm_accum")
    nw, w = pick({k: v for k, v in write.items() if ("Fu2<" in k) == (tag == "G2")}, "k_msm_accum")
    if nf and nw:
        fac = CAL[PATTERN[tag]]
        out[tag] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_kb_raw": f / nf, "write_kb_raw": w / nw, "pattern": PATTERN[tag], "fetch_factor": fac,
                    "traffic_bytes_per_launch": int(fac * 1024 * f / nf + 1024 * w / nw)}
nq_all, _ = pick(fetch, "k_quotient")
if "G1" in out and nq_all:
    out["G1"]["launches_per_proof"] = out["G1"]["launches_fetch_pass"] / nq_all   # 2: (a, b_g1, l) as one launch + h_query
nq, _ = pick(fetch, "k_quotient")
ncf, cf = pick(fetch, "k_ntt_cols"); nrf, rf = pick(fetch, "k_ntt_rows")
ncw, cw = pick(write, "k_ntt_cols"); nrw, rw = pick(write, "k_ntt_rows")
if ncf and nrf and ncw and nrw and nq:
    # pass-vectors per Groth16 proof: 12 with the key as loaded (6 transforms x 2 passes: c takes one), 8 over a bound key (`--bind 2`);
    # k_quotient runs once per proof
    passes = int(os.environ.get("ZKHIP_PMC_PASSES", "12"))
    cols_pat = os.environ.get("ZKHIP_NTT_COLS_PATTERN", "seg128")     # the cols pass reads C adjacent elements per row: seg64 at C = 2, seg128 at C = 4
    true_fetch = CAL[cols_pat] * 1024 * cf + CAL["pair16"] * 1024 * rf
    out["NTT"] = {"launches_fetch_pass": ncf + nrf, "proofs": nq, "fetch_kb_raw_per_proof": (cf + rf) / nq, "write_kb_raw_per_proof": (cw + rw) / nq,
                  "patterns": {"cols": cols_pat, "rows": "pair16"},
                  "traffic_bytes_per_pass": int((true_fetch + 1024 * (cw + rw)) / nq / passes), "algorithmic_bytes_per_pass": 2 * (1 << 20) * 32}
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zokrates_amd.build import csrc_hash  # noqa: E402
out["csrc_hash"] = csrc_hash()     # the sources these passes ran on (bench.py refuses the figures next to another build)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
