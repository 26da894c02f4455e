#!/usr/bin/env python3
"""bench.py — Groth16 proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-domain 20] [--curve bn128] [--kind dense] [--scheme g16]

A "step" is one Groth16 proof (sparse mat-vec, 7 NTTs, 5 MSMs, assembly) of the synthetic 2^20-constraint BN254
circuit of BASELINE.json configs[1].  The proving key, the constraint system and the assignments are
resident in HBM when the timed region starts; every step uses a different (witness, r, s).  The timed region is ONE
call of `zkhip_prove_g16_resident_batch` over the K steps — the library keeps two proofs in flight (steady-state
proofs/sec, the headline metric); `single_proof_ms` is the latency of an isolated `zkhip_prove_g16_resident` call.  With N > 1 every rank proves its own K proofs on its own GPU with a full copy of
the key (independent proofs: no data-path collective; "weak" scaling) and `value` is N*K / max-over-ranks time.

`--scheme gm17` runs BASELINE.json configs[4] instead (the same circuit through the GM17 prover: R1CS -> SAP on the
device, 5 NTTs over the twice-larger domain, 5 MSMs over the extended assignment); the default line is Groth16.

Besides the contract fields the JSON line carries
  roofline     — the dominant kernel's algorithmic bytes per launch / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline — the C++ restatement of zokrates_ark/ark-groth16 0.3.0 (oracle/, kind "port") timed on this
                 box's host cores on the same circuit, key and witness (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import importlib  # noqa: E402

_pkg = os.environ.get("ZKHIP_PKG", "zokrates_amd")   # development hook: A/B two builds of the library on the same box
native, parallel, synth = (importlib.import_module(_pkg + "." + m) for m in ("native", "parallel", "synth"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
SHARDED_LEG_TIMEOUT_S = 120


def make_proving_key(ctx, cs, circ, curve_id, scheme="g16"):
    """Setup for the synthetic circuit with fixed toxic waste, on the GPU (zkhip_setup_g16 / zkhip_setup_gm17)."""
    tox = synth.toxic_waste(curve_id, 0xC0FFEE)
    if scheme == "gm17":
        return native.setup_gm17(ctx, cs, (tox[0], tox[1], tox[2], tox[4]))   # alpha, beta, gamma, t
    return native.setup_g16(ctx, cs, tox)


def cpu_baseline(circ, pk_bytes, z, budget_s, gm17=False):
    """Times the CPU port of the reference path on the host cores: same circuit, key, assignment."""
    from oracle import cpu   # test infrastructure; used here only as the timed CPU baseline
    threads = cpu.hw_threads()
    oc = cpu.Circuit.from_csr(circ.curve_id, circ.n, circ.l, circ.w, circ.mats())
    opk = (cpu.Gm17ProvingKey if gm17 else cpu.ProvingKey).parse(circ.curve_id, pk_bytes)
    t0 = time.time()
    done, proofs = 0, []
    while True:
        if gm17:
            raw, _ = cpu.gm17_prove(oc, opk, z, 1000 + done, 3000 + done, 2000 + done, threads)
        else:
            raw, _ = cpu.prove(oc, opk, z, 1000 + done, 2000 + done, threads)
        proofs.append(raw)
        done += 1
        el = time.time() - t0
        if el >= budget_s or el + el / done > 2.5 * budget_s or done >= 64:
            break
    return {"value": done / el, "unit": "proofs/s", "cores": threads, "kind": "port",
            "sample": f"{done} proof(s) of the same 2^{int(np.log2(circ.N))} circuit in {el:.1f} s, "
                      f"C++ restatement of {'ark-gm17' if gm17 else 'ark-groth16'} 0.3.0 (Pippenger c=0.69*log2(n)+2, radix-2 FFT), "
                      f"{threads} threads",
            "ms_per_proof": 1000.0 * el / done}, proofs[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-domain", type=int, default=20)
    ap.add_argument("--curve", default="bn128")
    ap.add_argument("--kind", default="dense", choices=["dense", "sha", "poseidon"])
    ap.add_argument("--scheme", default="g16", choices=["g16", "gm17"])
    ap.add_argument("--witnesses", type=int, default=2, help="distinct assignments kept resident and cycled")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    args = ap.parse_args()

    ranks = parallel.Ranks()          # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; nccl = RCCL
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    barrier_sync = ranks.barrier

    curve_id = synth.CURVE_IDS[args.curve]
    ctx = native.Context(int(os.environ.get("ZKHIP_BENCH_DEVICE", local_rank)))   # env override: test hook (all ranks on one GPU)
    if args.kind == "poseidon":   # BASELINE.json configs[3]: the stdlib Poseidon hash chain, depth 1024 at a 2^18 domain
        poseidon = importlib.import_module(_pkg + ".poseidon")
        depth = 1024 << (args.log_domain - 18) if args.log_domain >= 18 else max(1, ((1 << args.log_domain) - 4) // 243)
        circ = poseidon.chain(curve_id, depth)
    else:
        circ = synth.circuit(curve_id, args.log_domain, kind=args.kind)
    cs = native.ConstraintSystem(ctx, curve_id, circ.n, circ.l, circ.w, circ.mats())
    t0 = time.time()
    gm17 = args.scheme == "gm17"
    pk_bytes = make_proving_key(ctx, cs, circ, curve_id, args.scheme)
    t_setup = time.time() - t0
    t0 = time.time()
    pk = native.ProvingKey(ctx, curve_id, pk_bytes, scheme=args.scheme)
    t_pkload = time.time() - t0
    nw = max(1, min(args.witnesses, args.steps + args.warmup))
    zs = [circ.assignment(ranks.witness_seed(i)) for i in range(nw)]
    t0 = time.time()
    resident = [native.Assignment(ctx, cs, z) for z in zs]
    t_h2d = (time.time() - t0) / nw

    def rs(i):
        if gm17:   # d1, d2, r
            return 0x1111111111111111 * (i + 1) + rank, 0x3333333333333333 * (i + 2) + rank, 0x2222222222222222 * (i + 3) + rank
        return 0x1111111111111111 * (i + 1) + rank, 0x2222222222222222 * (i + 3) + rank

    if gm17:
        prove_one = lambda a, rnd: native.prove_gm17(ctx, pk, cs, a, *rnd, want_timings=True)
        prove_many = lambda aa, rnds: native.prove_gm17_resident_batch(ctx, pk, cs, aa, rnds)
    else:
        prove_one = lambda a, rnd: native.prove_g16_resident(ctx, pk, cs, a, *rnd, want_timings=True)
        prove_many = lambda aa, rnds: native.prove_g16_resident_batch(ctx, pk, cs, aa, rnds)
    single = []
    for i in range(args.warmup):
        _, tm1 = prove_one(resident[i % nw], rs(i))
        single.append(tm1["total_ms"])
    if args.warmup:   # warm the pipelined path too (second slot's workspaces)
        prove_many([resident[i % nw] for i in range(2)], [rs(100 + i) for i in range(2)])
    steps = [args.warmup + i for i in range(args.steps)]
    barrier_sync()
    t_begin = time.perf_counter()
    proofs, acc = prove_many([resident[j % nw] for j in steps], [rs(j) for j in steps])
    barrier_sync()
    elapsed = time.perf_counter() - t_begin
    elapsed = ranks.max_over_ranks(elapsed)
    # isolated single-proof latency (not part of the timed region)
    for i in range(3):
        _, tm1 = prove_one(resident[i % nw], rs(200 + i))
        single.append(tm1["total_ms"])
    single_ms = min(single)

    avg = {k: v / args.steps for k, v in acc.items()}
    # ---- roofline of the dominant kernel (HIP events on the library's stream, inside the timed region)
    m, N = circ.m, circ.N
    if gm17:   # the MSMs run over the SAP variables and the SAP domain
        m, N = pk.m, pk.hlen - 1
    fq = native.FQ_BYTES[curve_id]
    kernels = {
        # algorithmic bytes per launch: bases read once + the 32-B scalars they pair with (SURVEY.md §8d (iv)+(v))
        "msm_accumulate<G2> (b_g2_query)": (avg["kernel_msm_accum_g2_ms"], (m + 2) * (4 * fq + 32), 1),
        "msm_accumulate<G1> (a/b_g1/l/h_query)": (avg["kernel_msm_accum_g1_ms"], (3 * (m + 2) + N) * (2 * fq + 32), 4),
    }
    name, (ms, bytes_all, launches) = max(kernels.items(), key=lambda kv: kv[1][0])
    achieved = bytes_all / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) — measured offline, same workload
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.log_domain == 20 and args.curve == "bn128" and args.kind == "dense" and not gm17:
        with open(pmc_path) as f:
            traffic = json.load(f).get("G2" if "G2" in name else "G1", {}).get("traffic_bytes_per_launch")
    # the honest bound of this kernel: mixed additions per second against the multiplier-limited rate of the same
    # kernel on synthetic data (tools/accum_bench.hip); W signed windows, one mixed addition per non-zero digit
    shape_w = {20: 16}.get(args.log_domain)
    compute = None
    if shape_w and args.curve == "bn128":
        madds = (((m + 2) if "G2" in name else (3 * (m + 2) + N)) * shape_w)
        peak = 5.29e9 if "G2" in name else 13.75e9
        compute = {"unit": "mixed additions/s", "achieved": madds / (ms * 1e-3), "peak": peak, "frac": madds / (ms * 1e-3) / peak,
                   "peak_source": "tools/accum_bench.hip on MI355X (same kernel, synthetic sorted lists)"}
    roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_launch": bytes_all / launches, "ms_per_launch": ms / launches, "launches_per_proof": launches,
                "compute_bound": compute,
                "note": "bucket accumulation is bound by integer-multiply issue (Montgomery products), not by HBM; durations are "
                        "HIP-event intervals on the MSM streams inside the timed region, where the five MSMs overlap"}
    b_alg = gm17_algorithmic_bytes(circ, fq, m, N) if gm17 else proof_algorithmic_bytes(circ, fq)
    workload = (f"synthetic R1CS {args.kind}, n = 2^{args.log_domain} - 2 constraints, {args.curve} GM17 (SAP: {m} variables, domain {N}), "
                f"5 NTTs + 5 MSMs per proof") if gm17 else (
        (f"Poseidon hash chain depth {circ.depth} (t = 3, 243 constraints per hash), n = {circ.n} constraints (QAP domain 2^{args.log_domain}), "
         if args.kind == "poseidon" else
         f"synthetic R1CS {args.kind}, n = 2^{args.log_domain} - 2 constraints (QAP domain 2^{args.log_domain}), ")
        + f"{args.curve} Groth16, 7 NTTs + 5 MSMs per proof")
    out = {
        "metric": "gm17_proofs_per_sec" if gm17 else "groth16_proofs_per_sec", "value": world * args.steps / elapsed, "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": workload, "curve": args.curve, "constraints": circ.n,
                   "variables": m, "domain": N, "parallelism": f"{world} independent prover(s), full key per GPU"},
        "single_proof_ms": single_ms, "phases_ms": avg,
        "whole_proof_hbm": {"algorithmic_bytes": b_alg, "achieved_GBs": b_alg / (elapsed / args.steps) / 1e9,
                            "frac": b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        "roofline": roofline,
        "host_ms": {"setup_gpu": 1000 * t_setup, "pk_load": 1000 * t_pkload, "assignment_h2d": 1000 * t_h2d},
        "device": ctx.describe(),
    }
    if world > 1:
        # latency mode: ONE proof sharded over all ranks (1/world of the bases per GPU, RCCL all-gather of the partial
        # records); reported next to the throughput metric, never instead of it.  A watchdog guarantees that the
        # throughput line is printed even if this optional leg hangs in a collective (e.g. one rank failed asymmetrically).
        import threading

        def give_up():
            if rank == 0:
                out["sharded_single_proof"] = {"error": "timed out after %d s" % SHARDED_LEG_TIMEOUT_S}
                out["cpu_baseline"] = None
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(SHARDED_LEG_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            shard = native.ProvingKey(ctx, curve_id, pk_bytes, rank=rank, world=world, scheme=args.scheme)
            z_common = native.Assignment(ctx, cs, circ.assignment(0x5EED7777))
            times = []
            for i in range(5):
                barrier_sync()
                t0 = time.perf_counter()
                proof = parallel.prove_sharded(ranks, ctx, shard, cs, z_common, 4242 + i, 777 + i, d1_d2=(31 + i, 59))
                barrier_sync()
                times.append(ranks.max_over_ranks(time.perf_counter() - t0))
            whole = prove_one(z_common, (31 + 4, 59, 4242 + 4) if gm17 else (4242 + 4, 777 + 4))[0]
            out["sharded_single_proof"] = {"ms": 1000.0 * min(times[1:]), "ranks": world, "identical_to_unsharded": bool(proof == whole),
                                           "exchange": "all-gather of one %d-byte record per rank" % native.partial_size(ctx, curve_id)}
        except Exception as e:  # the throughput line must survive a failure of the optional leg
            out["sharded_single_proof"] = {"error": repr(e)}
        watchdog.cancel()
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        base, cpu_proof = cpu_baseline(circ, pk_bytes, zs[0], args.cpu_seconds, gm17)
        out["cpu_baseline"] = base
        # same inputs -> byte-identical proof (the CPU leg doubles as a full-size parity check)
        rnd0 = (1000, 3000, 2000) if gm17 else (1000, 2000)
        gpu_proof = prove_one(resident[0], rnd0)[0]
        batch_proof = prove_many([resident[0]] * 3, [rnd0] * 3)[0]
        out["cpu_baseline"]["gpu_proof_identical"] = bool(gpu_proof == cpu_proof and all(p == cpu_proof for p in batch_proof))
        out["speedup_vs_cpu_baseline"] = out["value"] / base["value"]
    elif rank == 0:
        out["cpu_baseline"] = None   # N > 1 or --cpu-seconds 0
    if rank == 0 and world == 1:
        out["box_probe"] = box_probe()
    if rank == 0:
        print(json.dumps(out), flush=True)
    ranks.close()


def box_probe():
    """Latency figures of THIS box (tools/box_probe, a stand-alone HIP micro-benchmark built next to libzkhip; run after
    the timed region): the pool's boxes come in two kinds that differ 2-3.6x on the latency-bound fold kernels
    (DESIGN.md §8), and these numbers travel with the bench line so that a result can be attributed.  None if the
    binary is absent or fails."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "box_probe")
    if not os.access(exe, os.X_OK):
        return None
    try:
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=30).stdout
        grab = lambda pat: float(re.search(pat, txt).group(1))
        return {"lone_wave_mad_ns": grab(r"lone wave [\d.]+ ms \(([\d.]+) ns/iter\)"), "lds_hop_ns": grab(r"lds chain: lone wave [\d.]+ ms \(([\d.]+) ns/hop\)"),
                "global_hop_ns_256KiB": grab(r"over\s+256 KiB: ([\d.]+) ns/hop"), "global_hop_ns_256MiB": grab(r"over 262144 KiB: ([\d.]+) ns/hop")}
    except Exception:
        return None


def proof_algorithmic_bytes(circ, fq):
    """Compulsory HBM traffic of one proof: every input read once, every output written once per logical stage
    (SURVEY.md §8d): mat-vec, 7 transforms, MSM bases, MSM scalars."""
    F, n, m, N, w, l = 32, circ.n, circ.m, circ.N, circ.w, circ.l
    nnz = sum(int(mat[0][-1]) for mat in circ.mats())
    matvec = nnz * (F + 4) + 3 * (n + 1) * 8 + m * F + 3 * N * F
    transforms = 7 * 2 * N * F
    bases = (N - 1) * 2 * fq + w * 2 * fq + 2 * m * 2 * fq + m * 4 * fq
    scalars = N * F + w * F + 3 * m * F
    return matvec + transforms + bases + scalars


def gm17_algorithmic_bytes(circ, fq, M, D):
    """The same accounting for GM17: SAP rows (matrices + z read, two D-vectors and the n + l - 1 extension written),
    5 transforms over D, the five base sets (4 over the M SAP variables, one over D) and their scalars."""
    F, n, m = 32, circ.n, circ.m
    nnz = sum(int(mat[0][-1]) for mat in circ.mats())
    rows = nnz * (F + 4) + 3 * (n + 1) * 8 + m * F + 2 * D * F + (M - m) * F
    transforms = 5 * 2 * D * F
    bases = 3 * M * 2 * fq + M * 4 * fq + D * 2 * fq
    scalars = 4 * M * F + D * F
    return rows + transforms + bases + scalars


if __name__ == "__main__":
    main()
