#!/usr/bin/env python3
"""Per-kernel averages of one PMC counter from a rocprofv3 rocpd database (--pmc X --kernel-trace).
usage: pmc_stats.py results.db [out.md]"""
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
rows = list(db.execute("select kernel_name, counter_name, value, duration from counters_collection"))
agg = {}
for k, c, v, d in rows:
    a = agg.setdefault((short(k), c), [0, 0.0, 0])
    a[0] += 1; a[1] += v; a[2] += d
lines = ["| kernel | counter | launches | avg per launch (KB) | avg duration us |", "|---|---|---|---|---|"]
for (k, c), (n, tot, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("| `%s` | %s | %d | %.1f | %.1f |" % (k, c, n, tot / n, dur / n / 1e3))
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
