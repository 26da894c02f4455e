#!/usr/bin/env python3
"""Where a wavefront's cycles go, per kernel, from rocprofv3 PMC passes (separate runs, --kernel-trace only):

  pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
  pass B: SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQC_ICACHE_REQ SQC_ICACHE_MISSES
          TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE

SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; WAIT_ANY (parked on s_waitcnt / barrier) +
WAIT_INST_ANY (ready but not issued: dependency, pipe busy, arbitration) + ACTIVE_INST_ANY ~ WAVE_CYCLES
(/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots").  Everything is printed as a share of SQ_WAVE_CYCLES; the
L2 hit rate is TCC_HIT / (TCC_HIT + TCC_MISS).

usage: pmc_stall.py out.md passA.db [passB.db]      (writes out.md and out.json; kernels: the accumulations and the transform passes)"""
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"zk::Fe2<zk::(\w+)>", r"\1^2", name)
    name = re.sub(r"zk::Fe<zk::(\w+)>", r"\1", name)
    name = re.sub(r"zk::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return re.sub(r"^void ", "", name)


def load(path):
    agg = {}
    db = sqlite3.connect(path)
    for k, c, v, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        e = agg.setdefault(short(k), {}).setdefault(c, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += v
        e[2] += d
    return agg


def main():
    out = sys.argv[1]
    agg = {}
    for p in sys.argv[2:]:
        for k, cs in load(p).items():
            for c, e in cs.items():
                agg.setdefault(k, {}).setdefault(c, e)      # (a counter present in both passes: the first pass's)
    want = ("k_msm_accum<", "k_ntt_cols", "k_ntt_rows", "k_msm_fold", "k_msm_place", "k_msm_count", "k_matvec")
    rows = []
    js = {"source": "rocprofv3 --pmc (two passes, --kernel-trace only) over ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0 --e2e 0",
          "definition": "shares of SQ_WAVE_CYCLES (quad-cycles summed over waves): wait_any = parked on s_waitcnt/barrier, wait_inst_any = ready but not issued, "
                        "active_inst_* = issuing; l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); waves_per_simd = SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES / 4 where available", "kernels": {}}
    for k, cs in agg.items():
        if not k.startswith(want) or "SQ_WAVE_CYCLES" not in cs:
            continue

        def per(c):
            return cs[c][1] / cs[c][0] if c in cs and cs[c][0] else None

        wc = per("SQ_WAVE_CYCLES")
        n = cs["SQ_WAVE_CYCLES"][0]
        dur_us = cs["SQ_WAVE_CYCLES"][2] / n / 1e3

        def share(c):
            v = per(c)
            return None if v is None or not wc else v / wc

        hit, miss = per("TCC_HIT_sum"), per("TCC_MISS_sum")
        e = {"launches": n, "avg_us": dur_us, "wave_quad_cycles": wc, "wait_any": share("SQ_WAIT_ANY"), "wait_inst_any": share("SQ_WAIT_INST_ANY"),
             "active_inst_any": share("SQ_ACTIVE_INST_ANY"), "active_inst_valu": share("SQ_ACTIVE_INST_VALU"), "active_inst_vmem": share("SQ_ACTIVE_INST_VMEM"),
             "active_inst_sca": share("SQ_ACTIVE_INST_SCA"), "active_inst_lds": share("SQ_ACTIVE_INST_LDS"), "wait_inst_lds": share("SQ_WAIT_INST_LDS"),
             "valu_wave_instructions": per("SQ_INSTS_VALU"), "salu_instructions": per("SQ_INSTS_SALU"),
             "l2_hit": (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss else None,
             "tcp_pending_stall_cycles": per("TCP_PENDING_STALL_CYCLES_sum"),
             "icache_miss_rate": (per("SQC_ICACHE_MISSES") / per("SQC_ICACHE_REQ")) if per("SQC_ICACHE_REQ") else None,
             "busy_cycles": per("SQ_BUSY_CYCLES"), "gui_active": per("GRBM_GUI_ACTIVE")}
        js["kernels"][k] = e
        rows.append((dur_us * n, k, e))

    def f(x, pct=True):
        return "-" if x is None else ("%.1f" % (100 * x) if pct else "%.3g" % x)

    lines = ["| kernel | launches | avg us | parked (WAIT_ANY) % | ready, not issued (WAIT_INST_ANY) % | issuing (ACTIVE_INST_ANY) % | of it VALU % | VMEM % | scalar % | LDS % | L2 hit % | I-cache miss % |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for _, k, e in sorted(rows, reverse=True, key=lambda r: r[0]):
        lines.append("| `%s` | %d | %.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            k, e["launches"], e["avg_us"], f(e["wait_any"]), f(e["wait_inst_any"]), f(e["active_inst_any"]), f(e["active_inst_valu"]), f(e["active_inst_vmem"]),
            f(e["active_inst_sca"]), f(e["active_inst_lds"]), f(e["l2_hit"]), f(e["icache_miss_rate"])))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    try:
        from zokrates_amd.build import csrc_hash
        js["csrc_hash"] = csrc_hash()
    except Exception:
        pass
    text = "\n".join(lines)
    print(text)
    open(out, "w").write(text + "\n")
    json.dump(js, open(re.sub(r"\.md$", "", out) + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
