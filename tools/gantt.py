#!/usr/bin/env python3
"""Per-stream listing of ONE isolated proof out of a rocprofv3 rocpd database (--kernel-trace) of a bench run.
The run's isolated single proofs (bench.py times them after the pipelined region) show up as clusters of kernels
separated by host-side gaps; a cluster with exactly one G2 accumulation is one proof.  Prints, for the chosen
cluster, every kernel as (start offset us, duration us, stream, name) and the critical facts: when each MSM lane's
accumulation starts/ends, when its fold ends, and how much of the span has an accumulation kernel in flight.
Usage: gantt.py results.db [which=-4]      (index into the list of single-proof clusters; -4 = a resident-input one)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("zk::", "").replace("Bn254", "").replace("Bls381", "")
    return re.sub(r"\s+", "", name)[:48]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = sorted(cur.execute("select name, start, end, %s from kernels" % scol), key=lambda r: r[1])
    clusters, cur_c, hi = [], [], None
    for r in rows:
        if hi is not None and r[1] > hi + 100_000:      # 100 us with nothing in flight: the host is between two calls
            clusters.append(cur_c)
            cur_c = []
        cur_c.append(r)
        hi = r[2] if hi is None else max(hi, r[2])
    clusters.append(cur_c)
    singles = [c for c in clusters if sum("k_msm_accum" in r[0] and "Fu2" in r[0] for r in c) == 1]   # one G2 accumulation = one proof
    print("%d clusters, %d of them single proofs" % (len(clusters), len(singles)))
    if not singles:
        return
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -4
    c = singles[which if -len(singles) <= which < len(singles) else -1]
    t0, t1 = c[0][1], max(r[2] for r in c)
    print("proof: %.3f ms of kernels, %d launches, streams %s" % ((t1 - t0) / 1e6, len(c), sorted({r[3] for r in c})))
    print("%9s %9s  %-6s %s" % ("start_us", "dur_us", "stream", "kernel"))
    for r in c:
        print("%9.1f %9.1f  s%-5s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], short(r[0])))
    # accumulation kernels in flight over the span
    ev = []
    for r in c:
        if "k_msm_accum" in r[0]:
            ev.append((r[1], 1)); ev.append((r[2], -1))
    ev.sort()
    n, last, busy = 0, t0, {}
    for t, d in ev:
        busy[n] = busy.get(n, 0) + (t - last)
        last = t
        n += d
    busy[0] = busy.get(0, 0) + (t1 - last)
    print("accumulation kernels in flight: " + ", ".join("%d: %.2f ms" % (k, v / 1e6) for k, v in sorted(busy.items())))
    print("sum of kernel durations %.3f ms; accumulations %.3f ms" % (sum(r[2] - r[1] for r in c) / 1e6,
                                                                       sum(r[2] - r[1] for r in c if "k_msm_accum" in r[0]) / 1e6))


if __name__ == "__main__":
    main()
