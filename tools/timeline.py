#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 rocpd database (--kernel-trace) of a pipelined bench run:
how much of the timed window has an accumulation kernel in flight, how much only small kernels, how much nothing.
Usage: timeline.py results.db [first_fraction_to_skip=0.5]
       timeline.py results.db --proofs A B     the window from the A-th to the B-th launch of k_quotient (one per Groth16 proof: a run
                                               whose trace also holds key set-up, a bind or other legs picks its steady state by proof)"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else None
    scol = "stream_id" if "stream_id" in cols else None
    sel = "name, start, end" + ("," + qcol if qcol else "") + ("," + scol if scol else "")
    rows = sorted(cur.execute("select %s from kernels" % sel), key=lambda r: r[1])
    if len(sys.argv) > 4 and sys.argv[2] == "--proofs":
        marks = [r[1] for r in rows if "k_quotient" in r[0]]
        a, b = int(sys.argv[3]), int(sys.argv[4])
        lo, hi = marks[a], marks[b]
        print("proofs %d..%d of %d (k_quotient launches)" % (a, b, len(marks)))
        rows = [r for r in rows if r[1] >= lo and r[1] < hi]
    else:
        skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
        t0, t1 = rows[0][1], max(r[2] for r in rows)
        lo = t0 + (t1 - t0) * skip
        rows = [r for r in rows if r[1] >= lo]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    ev = []
    for r in rows:
        big = "k_msm_accum" in r[0]
        ev.append((r[1], 1, big)); ev.append((r[2], -1, big))
    ev.sort()
    nbig = nall = 0
    last = t0
    acc = {"accum": 0, "other_only": 0, "idle": 0}
    conc = {}
    for t, d, big in ev:
        dt = t - last
        if dt > 0:
            key = "accum" if nbig else ("other_only" if nall else "idle")
            acc[key] += dt
            conc[nbig] = conc.get(nbig, 0) + dt
        last = t
        nall += d
        if big:
            nbig += d
    span = t1 - t0
    print("window %.2f ms, %d kernels" % (span / 1e6, len(rows)))
    for k, v in acc.items():
        print("  %-11s %.2f ms  %.1f %%" % (k, v / 1e6, 100.0 * v / span))
    print("  accumulation kernels in flight: " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(conc.items())))
    if qcol:
        qs = {}
        for r in rows:
            qs.setdefault(r[3], [0, 0])
            qs[r[3]][0] += 1; qs[r[3]][1] += r[2] - r[1]
        print("  hardware queues: " + ", ".join("q%s: %d kernels %.1f ms" % (q, n, t / 1e6) for q, (n, t) in sorted(qs.items())))
    if scol:
        ss = {}
        for r in rows:
            ss.setdefault(r[-1], [0, 0])
            ss[r[-1]][0] += 1; ss[r[-1]][1] += r[2] - r[1]
        print("  streams: " + ", ".join("s%s: %d kernels %.1f ms" % (q, n, t / 1e6) for q, (n, t) in sorted(ss.items())))


if __name__ == "__main__":
    main()
