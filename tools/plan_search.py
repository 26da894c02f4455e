#!/usr/bin/env python3
"""Local search over stream plans (ZKHIP_PIPES, core.cuh make_pipe_streams): which dispatcher each stream of a resident prover sits on, scored on
FOUR workloads at once (dense 2^20 BN254, stdlib SHA-256 2^20, the Poseidon chain on BLS12-381 2^18, GM17 2^20) against streams made in order of
first use.  One bench.py process per (plan, workload).  usage: plan_search.py <minutes> [seed]   (run on the GPU box; prints one JSON line per plan)"""
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = {"dense": [], "sha": ["--kind", "sha256", "--log-domain", "20"], "poseidon": ["--curve", "bls12_381", "--log-domain", "18", "--kind", "poseidon"],
             "gm17": ["--scheme", "gm17"]}
STREAMS = ["M", "N", "O", "n", "G0", "Z0", "H0", "G1", "Z1", "H1", "G2", "Z2", "H2"]
REPLICA = {"M": 0, "N": 2, "O": 1, "n": 3, "G0": 1, "Z0": 3, "H0": 0, "G1": 2, "Z1": 3, "H1": 0, "G2": 1, "Z2": 2, "H2": 3}


def plan_string(p):
    return ",".join("%s=%d" % (k, p[k]) for k in STREAMS) if p else "-"


def run(plan, name):
    env = dict(os.environ, ZKHIP_BENCH_CHILD="1", ZKHIP_PIPES=plan_string(plan))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "6", "--witnesses", "2", "--cpu-seconds", "0", "--e2e", "0", "--serial-proofs", "0",
           "--repeats", "2", "--oracle", "none", "--configs", "0", "--bind", "2", "--pipe-plan", "1"] + WORKLOADS[name]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=150).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        ms = sorted(d["repeats"]["ms_per_step"])
        return 1000.0 / ms[0], d["single_proof_ms"]
    except Exception as e:
        return 0.0, 0.0


def evaluate(plan):
    return {w: run(plan, w) for w in WORKLOADS}


LONE_WEIGHT = float(os.environ.get("PLAN_LONE_WEIGHT", "0"))      # > 0: the lone proof's latency counts as well (geometric mean of base / plan, to this power)


def score(res, base):
    s, l = 1.0, 1.0
    for w in WORKLOADS:
        s *= (res[w][0] / base[w][0]) if base[w][0] else 1.0
        l *= (base[w][1] / res[w][1]) if res[w][1] and base[w][1] else 1.0
    return s ** (1.0 / len(WORKLOADS)) * (l ** (1.0 / len(WORKLOADS))) ** LONE_WEIGHT


def sizes(p):
    c = [0, 0, 0, 0]
    for v in p.values():
        c[v] += 1
    return c


def neighbour(p, rnd):
    q = dict(p)
    for _ in range(50):
        if rnd.random() < 0.5:
            a, b = rnd.sample(STREAMS, 2)
            if q[a] != q[b]:
                q[a], q[b] = q[b], q[a]
                return q
        else:
            a = rnd.choice(STREAMS)
            t = rnd.randrange(4)
            if t != q[a] and sizes(q)[t] < 4:
                q[a] = t
                return q
    return q


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 10
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t_end = time.time() + 60 * minutes
    base = evaluate(None)
    b2 = evaluate(None)      # two baselines, averaged: one run's noise (2 %) would tilt every score of the search
    base = {w: ((base[w][0] + b2[w][0]) / 2, (base[w][1] + b2[w][1]) / 2) for w in WORKLOADS}
    print(json.dumps({"plan": "-", "res": base}), flush=True)
    cur = dict(REPLICA)
    if os.environ.get("PLAN_START"):
        cur = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ["PLAN_START"].split(",")}
    cur_res = evaluate(cur)
    cur_s = score(cur_res, base)
    print(json.dumps({"plan": plan_string(cur), "score": round(cur_s, 4), "res": cur_res}), flush=True)
    best, best_s = dict(cur), cur_s
    while time.time() < t_end:
        cand = neighbour(cur, rnd)
        res = evaluate(cand)
        s = score(res, base)
        print(json.dumps({"plan": plan_string(cand), "score": round(s, 4), "res": res, "accepted": s > cur_s + 0.002}), flush=True)
        if s > cur_s + 0.002:
            cur, cur_s = cand, s
            if s > best_s:
                best, best_s = dict(cand), s
    again = evaluate(best)
    base2 = evaluate(None)
    print(json.dumps({"best": plan_string(best), "score_first": round(best_s, 4), "score_again_vs_first_baseline": round(score(again, base), 4),
                      "score_again_vs_second_baseline": round(score(again, base2), 4), "res_again": again, "baseline_again": base2}), flush=True)


if __name__ == "__main__":
    main()
