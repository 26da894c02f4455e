#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel table: calls, total/avg/min/max
duration, share of GPU time, registers / scratch / LDS per dispatch.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"zk::Fe2<zk::(\w+)>", r"\1^2", name)
    name = re.sub(r"zk::Fe<zk::(\w+)>", r"\1", name)
    name = re.sub(r"zk::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    name = re.sub(r"^void ", "", name)
    return name


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    want = [c for c in ("name", "start", "end", "duration", "vgpr_count", "accum_vgpr_count", "sgpr_count", "scratch_size", "lds_size",
                        "workgroup_size", "grid_size", "private_segment_size", "group_segment_size", "arch_vgpr_count") if c in cols]
    rows = list(cur.execute("select %s from kernels" % ",".join(want)))
    idx = {c: i for i, c in enumerate(want)}
    agg = {}
    for r in rows:
        name = short(r[idx["name"]])
        dur = (r[idx["end"]] - r[idx["start"]]) if "end" in idx else r[idx["duration"]]
        a = agg.setdefault(name, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "row": r})
        a["n"] += 1; a["tot"] += dur; a["min"] = min(a["min"], dur); a["max"] = max(a["max"], dur)
    total = sum(a["tot"] for a in agg.values())
    extra = [c for c in want if c not in ("name", "start", "end", "duration")]
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | " + " | ".join(extra) + " |",
             "|---|---|---|---|---|---|---|" + "---|" * len(extra)]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s |" % (
            name, a["n"], a["tot"] / 1e6, a["tot"] / a["n"] / 1e3, a["min"] / 1e3, a["max"] / 1e3, 100.0 * a["tot"] / total,
            " | ".join(str(a["row"][idx[c]]) for c in extra)))
    lines.append("")
    lines.append("total kernel time %.3f ms over %d dispatches" % (total / 1e6, len(rows)))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
