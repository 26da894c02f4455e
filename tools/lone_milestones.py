#!/usr/bin/env python3
"""Milestones of EVERY isolated proof in a rocprofv3 rocpd database (--kernel-trace of tools/lone_proof_probe.py): one line per proof, sorted by
duration — when the assignment's sort ends, when the witness map ends, when the h sort ends, begin / end of the three accumulation launches, the end
of every lane's fold chain — so that the fast and the slow proofs of one process can be put side by side (a lone proof moves by +-0.4 ms).
Usage: lone_milestones.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = sorted(cur.execute("select name, start, end, %s from kernels" % scol), key=lambda r: r[1])
clusters, cur_c, hi = [], [], None
for r in rows:
    if hi is not None and r[1] > hi + 100_000:
        clusters.append(cur_c)
        cur_c = []
    cur_c.append(r)
    hi = r[2] if hi is None else max(hi, r[2])
clusters.append(cur_c)
singles = [c for c in clusters if sum("k_msm_accum" in r[0] and "Fu2" in r[0] for r in c) == 1]
out = []
for c in singles:
    t0 = c[0][1]
    us = lambda t: (t - t0) / 1e3
    acc = [r for r in c if "k_msm_accum" in r[0]]
    g2 = [r for r in acc if "Fu2" in r[0]]
    g1 = sorted([r for r in acc if "Fu2" not in r[0]], key=lambda r: r[1])
    if len(g1) != 2:
        continue
    fine = sorted([r for r in c if "k_msm_part_fine" in r[0]], key=lambda r: r[2])
    quot = [r for r in c if "k_quotient" in r[0]]
    scans = [r for r in c if "fold_final_scan" in r[0]]
    lane_end = {}
    for r in scans:
        lane_end[r[3]] = max(lane_end.get(r[3], 0), r[2])
    ends = sorted(us(v) for v in lane_end.values())
    out.append({"total": us(max(r[2] for r in c)), "z_sorted": us(fine[0][2]) if fine else -1, "quotient_end": us(quot[-1][2]) if quot else -1,
                "h_sorted": us(fine[-1][2]) if len(fine) > 1 else -1, "g2": (us(g2[0][1]), us(g2[0][2])), "abl": (us(g1[0][1]), us(g1[0][2])),
                "h": (us(g1[1][1]), us(g1[1][2])), "fold_ends": ends})
out.sort(key=lambda o: o["total"])
print("%d proofs; us from the proof's first kernel" % len(out))
print("%8s %9s %9s %9s  %-17s %-17s %-17s %s" % ("total", "z_sorted", "quot_end", "h_sorted", "G2 acc", "A/B1/L acc", "H acc", "fold chains end"))
for o in out:
    print("%8.0f %9.0f %9.0f %9.0f  %7.0f-%-9.0f %7.0f-%-9.0f %7.0f-%-9.0f %s" % (o["total"], o["z_sorted"], o["quotient_end"], o["h_sorted"], o["g2"][0], o["g2"][1],
                                                                                 o["abl"][0], o["abl"][1], o["h"][0], o["h"][1], " ".join("%.0f" % e for e in o["fold_ends"])))
