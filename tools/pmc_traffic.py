#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from two rocprofv3 PMC passes of the SAME
command (`--pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`, separate runs as MI355X_MICROARCH.md
prescribes; FETCH_SIZE doubled: gfx950 tallies 128-byte requests as 64 bytes).
usage: pmc_traffic.py fetch.db write.db out.json <source note>"""
import json
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
out = {"source": sys.argv[4] if len(sys.argv) > 4 else "",
       "correction": "FETCH_SIZE x 2 (gfx950 counts 128-byte requests as 64 bytes: MI355X_MICROARCH.md); WRITE_SIZE as reported; counters are in KB; "
                     "64-byte random gathers are outside the calibrated access pattern: treat the accumulation figures as approximate"}
for tag, needles in (("G1", ("k_msm_accum", "Fu<", "Bn254Fq")), ("G2", ("k_msm_accum", "Fu2<", "Bn254Fq"))):
    needles_ = needles if tag == "G2" else needles
    nf, f = pick({k: v for k, v in fetch.items() if ("Fu2<" in k) == (tag == "G2")}, "k_msm_accum")
    nw, w = pick({k: v for k, v in write.items() if ("Fu2<" in k) == (tag == "G2")}, "k_msm_accum")
    if nf and nw:
        out[tag] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_kb_raw": f / nf, "write_kb_raw": w / nw,
                    "traffic_bytes_per_launch": int(2 * 1024 * f / nf + 1024 * w / nw)}
nq_all, _ = pick(fetch, "k_quotient")
if "G1" in out and nq_all:
    out["G1"]["launches_per_proof"] = out["G1"]["launches_fetch_pass"] / nq_all   # 2: (a, b_g1, l) as one launch + h_query
nf, f = pick(fetch, "k_ntt_")
nw, w = pick(write, "k_ntt_")
nq, _ = pick(fetch, "k_quotient")
if nf and nw and nq:
    passes = 12   # pass-vectors per Groth16 proof (6 transforms x 2 passes: c takes one); k_quotient runs once per proof
    out["NTT"] = {"launches_fetch_pass": nf, "proofs": nq, "fetch_kb_raw_per_proof": f / nq, "write_kb_raw_per_proof": w / nq,
                  "traffic_bytes_per_pass": int((2 * 1024 * f / nq + 1024 * w / nq) / passes), "algorithmic_bytes_per_pass": 2 * (1 << 20) * 32}
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zokrates_amd.build import csrc_hash  # noqa: E402
out["csrc_hash"] = csrc_hash()     # the sources these passes ran on (bench.py refuses the figures next to another build)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
