#!/usr/bin/env python3
"""bench.py — Groth16 proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-domain 20] [--curve bn128] [--kind dense] [--scheme g16]

A "step" is one Groth16 proof (sparse mat-vec, 6 NTTs, 5 MSMs, assembly) of the synthetic 2^20-constraint BN254
circuit of BASELINE.json configs[1].  The proving key, the constraint system and the assignments are
resident in HBM when the timed region starts; EVERY step proves a different witness (--witnesses defaults to --steps,
all resident: 32 x 32 MiB) with its own (r, s).  The timed region is ONE call of `zkhip_prove_g16_resident_batch` over
the K steps — the library keeps three proofs in flight (steady-state proofs/sec, the headline metric); the measurement runs
in a child process that a parent supervises (a GPU fault kills the process that owns the queue: see supervise()).
`single_proof_ms` is the latency of an isolated proof of a resident assignment (the minimum of seventeen; `single_proof_ms_stats` has the
median and the spread), `single_proof_from_host_ms` the same
from an assignment in host memory to the proof bytes in host memory (SURVEY.md §8d's definition; the PCIe leg is
never part of `value`).  With N > 1 every rank proves its own K proofs on its own GPU with a full copy of the key
(independent proofs: no data-path collective; "weak" scaling) and `value` is N*K / max-over-ranks time.

Launch contract.  `python bench.py --gpus N` on its own starts the N ranks itself (one process per GPU, `nccl` = RCCL,
rendezvous on 127.0.0.1) and relays rank 0's line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N` it is one of the ranks the launcher started (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).
`n_gpus` is the number of ranks that RAN; fewer visible GPUs than N, or --gpus disagreeing with WORLD_SIZE, is an error
(a JSON line with "error", non-zero exit status), never a silently smaller measurement.

`--scheme gm17` runs BASELINE.json configs[4] instead (the same circuit through the GM17 prover: R1CS -> SAP on the
device, 5 NTTs over the twice-larger domain, 5 MSMs over the extended assignment); the default line is Groth16.

Besides the contract fields the JSON line carries
  roofline      — the dominant kernel (bucket accumulation): algorithmic bytes per launch / HIP-event duration vs 8 TB/s,
                  inside the timed region (`frac`, five MSM streams and two proofs overlapping) and from a short
                  one-stream leg after it (`frac_serial`, un-overlapped kernel time); `traffic` is the HBM traffic per
                  launch from the rocprofv3 PMC passes committed under profiles/ (`traffic_source` names the file)
  roofline_ntt  — the same for the NTT passes, the kernels the HBM roofline is meaningful for
  cpu_baseline  — the C++ restatement of zokrates_ark/ark-groth16 0.3.0 (oracle/, kind "port") timed on this
                  box's host cores on the same circuit, key and witness (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
_pkg = os.environ.get("ZKHIP_PKG", "zokrates_amd")   # development hook: A/B two builds of the library on the same box


CONFIG_LEGS = [
    # (key, BASELINE.json config, bench.py arguments, timed steps: the thin circuits' proofs take 3-5 ms, and a region of 16 of them is a tenth pipeline
    #  fill and drain — the Poseidon leg read 180-184 proofs/s where 32-step regions read 195-201: 48 steps cost a quarter of a second)
    ("gm17_2e20", "configs[4]: GM17 on the 2^20-constraint BN254 circuit", ["--scheme", "gm17"], 16),
    ("poseidon_chain_bls12_381_2e18", "configs[3]: stdlib Poseidon hash chain (depth 1024), BLS12-381", ["--curve", "bls12_381", "--log-domain", "18", "--kind", "poseidon"], 48),
    ("sha256_stdlib_2e20", "configs[0] at the size of configs[1]: stdlib sha256/512bitPacked.zok side by side up to a 2^20 domain", ["--kind", "sha256", "--log-domain", "20"], 48),
    ("dense_2e22_and_8_members", "configs[2]: 2^22 constraints, one GPU whole and as 8 members of one proof", ["--log-domain", "22", "--members", "8"], 16),
]


def config_legs(args, budget_s=None):
    """BASELINE.json's other configurations as short runs of THIS script (a process each: a leg that dies or stalls costs its own entry,
    not the line): 16 or 48 timed steps (CONFIG_LEGS) after 4 warm-up steps, the key bound as a resident prover would, the device proof — single, batched,
    bound — held to the oracle's closed form.  Compact records; `wall_s` is the leg's whole process."""
    import subprocess
    budget_s = budget_s or float(os.environ.get("ZKHIP_BENCH_CONFIGS_BUDGET_S", "95"))
    t_all = time.time()
    res = {}
    legs = CONFIG_LEGS
    if os.environ.get("ZKHIP_BENCH_TEST_LEGS"):      # tests/test_bench_cli.py: the same four legs at toy size (the emulator build)
        legs = [(k, w, {"gm17_2e20": ["--scheme", "gm17", "--log-domain", "5"], "poseidon_chain_bls12_381_2e18": ["--curve", "bls12_381", "--log-domain", "8", "--kind", "poseidon"],
                        "sha256_stdlib_2e20": ["--kind", "sha", "--log-domain", "6"], "dense_2e22_and_8_members": ["--log-domain", "6", "--members", "2"]}[k]
                , 16) for k, w, _, _ in CONFIG_LEGS]
        budget_s = 1200
    for key, what, extra, leg_steps in legs:
        left = budget_s - (time.time() - t_all)
        if left < 15:
            res[key] = {"config": what, "skipped": "time budget of the configs block (%.0f s) spent" % budget_s}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(leg_steps), "--warmup", "4", "--witnesses", "2", "--cpu-seconds", "0", "--e2e", "0",
               "--serial-proofs", "0", "--repeats", "2", "--oracle", "trapdoor", "--configs", "0"] + extra
        env = dict(os.environ, ZKHIP_BENCH_CHILD="1", ZKHIP_BENCH_LEG="1", ZKHIP_BENCH_STAGES="")
        env.pop("ZKHIP_BENCH_STAGES")
        t0 = time.time()
        rec = {"config": what}
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=left if os.environ.get("ZKHIP_BENCH_TEST_LEGS") else min(left, 75.0))
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                rec["error"] = "exit status %d: %s" % (p.returncode, (p.stderr or "")[-300:])
            else:
                d = json.loads(lines[-1])
                rec.update({"proofs_per_s": d["value"], "ms_per_step": d["ms_per_step"], "single_proof_ms": d["single_proof_ms"],
                            "single_proof_median_ms": (d.get("single_proof_ms_stats") or {}).get("median"),
                            "identical_to_oracle": d.get("identical_to_oracle"), "oracle": d.get("oracle"),
                            "key_bound": d["bound_key"].get("bound"), "bind_ms": d["bound_key"].get("bind_ms"),
                            "proofs_per_s_unbound": d.get("value_unbound"), "single_proof_unbound_ms": d.get("single_proof_unbound_ms"),
                            "steps": d["steps"], "constraints": d["config"]["constraints"], "domain": d["config"]["domain"], "curve": d["config"]["curve"]})
                mm = d.get("multi_single_proof")
                if mm:
                    rec["members"] = {k: mm.get(k) for k in ("members", "distinct_gpus", "ms", "ms_unbound", "identical_to_unsharded", "identical_to_oracle", "key_bound",
                                                               "bind_ms", "kernel_ntt_ms", "error") if k in mm}
        except subprocess.TimeoutExpired:
            rec["error"] = "timed out"
        except Exception as e:
            rec["error"] = repr(e)[:200]
        rec["wall_s"] = round(time.time() - t0, 1)
        res[key] = rec
    return res


CLI_DEFERRED = "deferred to the supervising process"


def cli_run(state):
    """The `generate-proof` processes of cli_end_to_end (below), over the files it left in state["directory"]; removes the directory.  Runs in the
    measuring process of an unsupervised run, in the supervising process otherwise (no HIP runtime there, the device released)."""
    import shutil
    import subprocess
    d, scheme = state["directory"], state["scheme"]
    res = {"directory": d, "write_input_files_ms": state["write_input_files_ms"], "file_bytes": dict(state["file_bytes"])}
    try:
        paths = {k: os.path.join(d, k) for k in ("out", "witness", "proving.key", "proof.json", "cache")}
        want = open(os.path.join(d, "expected_proof.json")).read()
        ing = None
        # the native executable of the compiled host layer (csrc/host: C++ over the C ABI), and the Python shim
        native_exe = os.path.join(ROOT, _pkg, "zkhip-cli")
        if os.environ.get("ZKHIP_LIBRARY", "").endswith("libzkhip_emu.so"):      # (tests: the emulator build of the same executable)
            native_exe = os.path.join(ROOT, "tests", "_emu", "zkhip-cli-emu")

        t_leg = time.perf_counter()

        gap_s = float(os.environ.get("ZKHIP_BENCH_CLI_GAP_S", "0"))

        def run(name, extra, exe=None):
            if time.perf_counter() - t_leg > 150:          # the leg must never hold the throughput line back for long
                res[name] = {"skipped": "time budget of this leg (150 s) spent"}
                return
            if gap_s > 0:                                  # (experiment: does a process pay for the previous one's teardown in the driver?)
                time.sleep(gap_s)
            if os.path.exists(paths["proof.json"]):
                os.remove(paths["proof.json"])
            cmd = ([exe] if exe else [sys.executable, "-m", _pkg + ".cli"]) + [
                "generate-proof", "-i", paths["out"], "-w", paths["witness"], "-p", paths["proving.key"],
                "-j", paths["proof.json"], "-s", scheme, "--entropy", "bench", "--timings"] + extra
            t0 = time.perf_counter()
            p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=120)
            wall = 1000.0 * (time.perf_counter() - t0)
            rec = {"process_wall_ms": wall}
            if p.returncode != 0:
                rec["error"] = (p.stderr or p.stdout)[-400:]
            else:
                for line in p.stdout.splitlines():
                    if line.startswith("timings "):
                        rec.update(json.loads(line[8:]))
                rec["proof_json_identical_to_resident_prover"] = open(paths["proof.json"]).read() == want
                if "--verify" in extra:
                    rec["verified"] = "verified against the verification key" in p.stdout
            res[name] = rec

        # three runs of the compiled executable (csrc/host: the drop-in's own front end): from the proving.key; the first run with a key
        # cache (writes the device-layout image); from that image.  ZKHIP_BENCH_CLI_ALL=1 adds round 5's other legs (the self check by
        # the compiled verifier, the full tables, the Python shim's three) — tests/test_bench_cli.py runs them at toy size.
        have_native = os.access(native_exe, os.X_OK)
        if have_native:
            paths["cache_native"] = os.path.join(d, "cache_native")
            run("native_from_proving_key", [], native_exe)
            run("native_first_run_with_key_cache", ["--key-cache", paths["cache_native"]], native_exe)
            # (the figure a `generate-proof` user sees: three processes, the one with the median wall clock reported and all three listed —
            # HIP start alone moves between 0.11 and 0.27 s from one process to the next on one box: profiles/r7k_bench_driver_command.json)
            runs = []
            for _ in range(3):
                run("native_from_key_image", ["--key-cache", paths["cache_native"]], native_exe)
                if "process_wall_ms" not in res["native_from_key_image"] or "error" in res["native_from_key_image"]:
                    break
                runs.append(res["native_from_key_image"])
            if runs:
                runs.sort(key=lambda r: r["process_wall_ms"])
                res["native_from_key_image"] = dict(runs[len(runs) // 2], process_wall_ms_runs=[round(r["process_wall_ms"], 1) for r in runs])
            res["file_bytes"]["key_image"] = sum(os.path.getsize(os.path.join(paths["cache_native"], f)) for f in os.listdir(paths["cache_native"]))
            ing = res["native_from_proving_key"].get("parse_program_ms")
        if os.environ.get("ZKHIP_BENCH_CLI_ALL") or not have_native:
            if have_native:
                run("native_from_key_image_with_verify", ["--key-cache", paths["cache_native"], "--verify"], native_exe)
                run("native_from_proving_key_full_tables", ["--full-tables"], native_exe)
            run("from_proving_key", [])
            run("first_run_with_key_cache", ["--key-cache", paths["cache"]])
            run("from_key_image", ["--key-cache", paths["cache"]])
            if not have_native:
                res["file_bytes"]["key_image"] = sum(os.path.getsize(os.path.join(paths["cache"], f)) for f in os.listdir(paths["cache"]))
                ing = res["from_proving_key"].get("parse_program_ms")
        if ing:
            res["ingest_constraints_per_s"] = state["constraints"] / (ing * 1e-3)
    except Exception as e:   # the throughput line must survive a failure of this leg
        res["error"] = repr(e)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


CONFIGS_DEFERRED = "deferred to the supervising process"


def supervise():
    """N = 1 runs measure in a CHILD process and the parent — which never loads the HIP runtime — relays its line.  A GPU
    memory fault does not raise an error, it aborts the process that owns the queue (round 2's driver run ended that way,
    3.2 s in, with nothing on stdout).  The child prints stage marks on stderr (ZKHIP_BENCH_STAGES); if it dies, the parent
    keeps what it left behind (exit status, last completed stage, the runtime's message) and measures once more in a fresh
    process: the line that is printed then carries the failed attempt under "attempts" instead of hiding it, and if the
    second attempt dies too a line with "error" and value null is printed and the exit status is 1."""
    import subprocess
    import tempfile
    attempts = []
    for attempt in range(2):
        env = dict(os.environ, ZKHIP_BENCH_CHILD="1", ZKHIP_BENCH_STAGES="1", ZKHIP_BENCH_ATTEMPT=str(attempt))
        if attempt:
            env["ZKHIP_BENCH_BIND"] = "0"      # the second attempt measures with the key as loaded (whatever killed the first, the
                                               # binding's kernels are the youngest code on the path)
        with tempfile.TemporaryFile() as err:
            proc = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], stdout=subprocess.PIPE, stderr=err, env=env)
            err.seek(0)
            etxt = err.read().decode(errors="replace")
        sys.stderr.write(etxt)
        sys.stderr.flush()
        out_lines = proc.stdout.decode(errors="replace").splitlines()
        json_lines = [l for l in out_lines if l.startswith("{")]
        for l in out_lines:                      # (anything else the child or a library wrote to stdout: kept, but off the JSON channel)
            if not l.startswith("{"):
                print(l, file=sys.stderr)
        stages = [l for l in etxt.splitlines() if l.startswith("[bench]")]
        other = "\n".join(l for l in etxt.splitlines() if not l.startswith("[bench]"))
        rec = {"exit_status": proc.returncode, "signal": -proc.returncode if proc.returncode < 0 else None,
               "last_stage": stages[-1].split(None, 3)[-1] if stages else None}
        if proc.returncode == 0 and json_lines:
            doc = json.loads(json_lines[-1])
            doc["attempts"] = attempts + [rec]
            cli = doc.get("cli_end_to_end_ms")
            if isinstance(cli, dict) and CLI_DEFERRED in cli:      # (the `generate-proof` processes: from here too, the device released)
                doc["cli_end_to_end_ms"] = cli_run(cli[CLI_DEFERRED])
            if doc.get("configs") == CONFIGS_DEFERRED:
                # BASELINE.json's other configurations run from HERE, after the measuring process has gone: as its children — beside a process that
                # still holds the device, even with its context closed — the legs read 3-5 % below the same commands run alone (GM17 81.3 against
                # 85.4 proofs/s, the Poseidon chain 188 against 192: profiles/r8h_*).  This process never loads the HIP runtime.
                del doc["configs"]
                doc["configs"] = config_legs(None)      # LAST key of the line on purpose: it survives a reader that keeps the tail
            print(json.dumps(doc), flush=True)
            return 0
        rec["stderr_tail"] = other[-800:]
        attempts.append(rec)
    print(json.dumps({"metric": "groth16_proofs_per_sec", "value": None, "unit": "proofs/s", "n_gpus": 1, "higher_is_better": True,
                      "error": "the measuring process died twice (see attempts)", "attempts": attempts}), flush=True)
    return 1


def requested_gpus(argv):
    """--gpus N as given on the command line (both `--gpus N` and `--gpus=N`); 1 when absent."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def visible_gpus():
    """GPUs the library sees, counted in a throw-away process (the launching parent never loads the HIP runtime)."""
    import subprocess
    code = ("import importlib, os, sys; sys.path.insert(0, %r); "
            "print(importlib.import_module(os.environ.get('ZKHIP_PKG', 'zokrates_amd') + '.native').default_library().device_count())"
            % os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        raise RuntimeError("cannot count the GPUs: " + (p.stderr or p.stdout)[-400:])
    return int(p.stdout.strip().splitlines()[-1])


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here — one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set as `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N` would set them, `nccl` (= RCCL) underneath — relay rank 0's JSON line and fail loudly (one JSON line
    with "error", exit status 1) when the box has fewer than N GPUs or a rank dies.  `n_gpus` of the line is the number of ranks
    that ran, never the number asked for."""
    import socket
    import subprocess
    import tempfile

    def fail(msg, extra=None):
        doc = {"metric": "groth16_proofs_per_sec", "value": None, "unit": "proofs/s", "n_gpus": 0, "requested_gpus": n,
               "higher_is_better": True, "error": msg}
        doc.update(extra or {})
        print(json.dumps(doc), flush=True)
        print("bench.py: " + msg, file=sys.stderr, flush=True)
        return 1

    if not os.environ.get("ZKHIP_BENCH_DEVICE"):          # (test hook: every rank on the one device of the emulator build)
        try:
            have = visible_gpus()
        except Exception as e:
            return fail(str(e))
        if have < n:
            return fail("--gpus %d asked for, %d GPU(s) visible: refusing to measure fewer GPUs than the line would claim" % (n, have),
                        {"visible_gpus": have})
    with socket.socket() as sk:                            # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, errs = [], []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKHIP_BENCH_LAUNCHED="self")
        err = tempfile.TemporaryFile()
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=err))
    # a rank that dies before the rendezvous would leave the others waiting in it: watch all of them, end all on the first failure
    budget = float(os.environ.get("ZKHIP_BENCH_LAUNCH_TIMEOUT_S", "1800"))
    t0, failed, t_done = time.time(), None, None
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with status %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if codes[0] == 0:
            # rank 0 is done (its line is on the pipe): the others only have to leave; one that does not — stuck behind an optional
            # leg rank 0 abandoned — is ended after a grace period instead of holding the finished measurement back
            t_done = t_done or time.time()
            if time.time() - t_done > 30:
                for p in procs[1:]:
                    if p.poll() is None:
                        p.kill()
                break
        if time.time() - t0 > budget:
            failed = "ranks still running after %.0f s" % budget
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    tails = []
    for rank, err in enumerate(errs):
        err.seek(0)
        txt = err.read().decode(errors="replace")
        err.close()
        sys.stderr.write(txt)
        tails.append(txt[-600:])
    sys.stderr.flush()
    lines = [l for l in (out0[0].decode(errors="replace") if out0 else "").splitlines() if l.startswith("{")]
    if failed or not lines:
        return fail(failed or "rank 0 printed no result line", {"stderr_tails": tails})
    print(lines[-1], flush=True)
    return 0


if __name__ == "__main__" and not any(a in ("-h", "--help") for a in sys.argv[1:]):
    _want = requested_gpus(sys.argv[1:])
    _world_env = os.environ.get("WORLD_SIZE")
    _explicit = any(a == "--gpus" or a.startswith("--gpus=") for a in sys.argv[1:])
    if _want < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if _world_env is not None and _explicit and _want != int(_world_env):
        # under a launcher (torch.distributed.run) the ranks exist already: a line that would claim another number of GPUs than
        # the launcher started is refused, not silently re-labelled
        print("bench.py: --gpus %d but WORLD_SIZE=%s: start as many ranks as --gpus says" % (_want, _world_env), file=sys.stderr)
        sys.exit(2)
    if _world_env is None and _want > 1:
        sys.exit(launch_ranks(_want))
    if int(_world_env or "1") == 1 and not os.environ.get("ZKHIP_BENCH_CHILD"):
        sys.exit(supervise())

import numpy as np

sys.path.insert(0, ROOT)

import importlib  # noqa: E402

native, parallel, synth = (importlib.import_module(_pkg + "." + m) for m in ("native", "parallel", "synth"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
SHARDED_LEG_TIMEOUT_S = int(os.environ.get("ZKHIP_BENCH_LEG_TIMEOUT_S", "120"))   # the optional latency legs, together
PMC_TRAFFIC_FILE = os.path.join("profiles", "pmc_traffic.json")


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


def make_witnesses(circ, seeds):
    """Distinct satisfying assignments (host, pure Python: a 2^20-step multiplication chain each), generated by forked
    workers BEFORE the HIP runtime and torch.distributed start.  The workers are plain os.fork children that answer over a
    pipe and leave with os._exit: no SIGTERM (multiprocessing.Pool ends its workers with one) and no exit handlers — under
    `rocprofv3 --pmc` the profiler's handler for either in a forked child never returned and the parent waited in wait4
    until the pass's timeout (gpurun_out r3y: two 900 s passes lost that way)."""
    workers = min(len(seeds), os.cpu_count() or 1, 32)
    if workers <= 1:
        return [circ.assignment(s) for s in seeds]
    import pickle
    jobs = []
    for k in range(workers):
        mine = list(range(k, len(seeds), workers))
        rd, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            status = 1
            try:
                os.close(rd)
                blob = pickle.dumps([circ.assignment(seeds[i]) for i in mine], protocol=pickle.HIGHEST_PROTOCOL)
                with os.fdopen(wr, "wb") as f:
                    f.write(blob)
                status = 0
            finally:
                os._exit(status)
        os.close(wr)
        jobs.append((pid, rd, mine))
    out = [None] * len(seeds)
    failed = []
    for pid, rd, mine in jobs:
        with os.fdopen(rd, "rb") as f:
            blob = f.read()
        _, st = os.waitpid(pid, 0)
        if st != 0 or not blob:
            failed.append((pid, st))
            continue
        for i, z in zip(mine, pickle.loads(blob)):
            out[i] = z
    if failed:
        raise RuntimeError(f"witness workers failed: {failed}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-domain", type=int, default=20)
    ap.add_argument("--curve", default="bn128")
    ap.add_argument("--kind", default="dense", choices=["dense", "sha", "sha256", "poseidon"])
    ap.add_argument("--scheme", default="g16", choices=["g16", "gm17"])
    ap.add_argument("--witnesses", type=int, default=0, help="distinct assignments kept resident (0 = one per timed step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--serial-proofs", type=int, default=3, help="proofs of the one-stream leg after the timed region (0 = skip)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions in all: `value` is the FIRST (the contract's K steps after W warm-up steps); the others repeat it and "
                         "`repeats` reports every region, the median and the spread")
    ap.add_argument("--constraints", type=int, default=0, help="exact constraint count (default 2^log_domain - 2, i.e. a domain of exactly 2^log_domain)")
    ap.add_argument("--e2e", type=int, default=1,
                    help="after the timed region, time the reference-shaped flow `generate-proof` (program + witness + proving.key files -> "
                         "proof.json, one fresh process per proof) through zokrates_amd.cli (0 = skip)")
    ap.add_argument("--members", type=int, default=0,
                    help="after the timed region, prove ONE proof across this many members inside the library (zkhip_prove_*_multi); "
                         "members take the visible GPUs in turn, sharing them when there are fewer (0 = skip; N > 1 ranks: rank 0 drives all GPUs)")
    ap.add_argument("--bind", type=int, default=1,
                    help="Groth16: bind the resident key to the resident constraint system before the timed region (zkhip_pk_bind_r1cs: four "
                         "transforms per proof instead of six, no C mat-vec, same proof bytes — checked here against an unbound proof; the line "
                         "also times one region with the key unbound again).  0 = the key as loaded; 2 = bound, without the unbound region")
    ap.add_argument("--pipe-plan", type=int, default=-1,
                    help="the stream plan of resident provers (ZKHIP_TUNE_PIPE_PLAN: the context's sixteen streams made at its first proof, each on a "
                         "chosen one of the GPU's four dispatchers; the layout comes from a local search over four workloads, tools/plan_search.py).  "
                         "Against streams in order of first use (profiles/r7g_*, r7h_*: the last two lines of the search are the plan and no plan measured back to back): dense 2^20 level in batches and a lone proof 0.25-0.3 ms "
                         "sooner, stdlib SHA-256 +8-9 %%, the Poseidon chain on BLS12-381 +3 %%, GM17 level.  -1 (default) and 1 = on, except where --members "
                         "puts more contexts into the process; 0 = off")
    ap.add_argument("--oracle", default="auto", choices=["auto", "algorithmic", "trapdoor", "none"],
                    help="what the device proof is held to: `algorithmic` = the cpu_baseline leg's proof (the C++ restatement of ark's prover, "
                         "needs --cpu-seconds > 0), `trapdoor` = the closed form from the setup's toxic waste (oracle/c, independent of every "
                         "transform and MSM; seconds at any size), auto = algorithmic when the CPU leg runs (N = 1), trapdoor on rank 0 for N > 1, else none")
    ap.add_argument("--configs", type=int, default=-1,
                    help="after everything else, run BASELINE.json's other configurations as short legs (GM17 2^20, Poseidon chain on BLS12-381, "
                         "stdlib SHA-256 2^20, 2^22 on one GPU and as 8 members), each checked against the oracle's closed form, and append them as "
                         "the LAST key of the line (`configs`).  -1 = only for the default workload on one GPU; 0 = never; 1 = always")
    args = ap.parse_args()
    # the host's explicit choice of HIP hardware queues (zkhip_init; the library never touches the environment itself): 16 for a process
    # with ONE resident prover, 8 as soon as it will hold several contexts (--members: every queue reserves scratch for the largest frame)
    # (ranks that SHARE a device — the ZKHIP_BENCH_DEVICE test hook — keep 8 queues each and no stream plan: two provers' 32 queues on one GPU
    # are time-sliced, 51 proofs/s where 8 + 8 give 100: profiles/r7o_two_ranks_one_gpu.txt)
    shared_gpu = os.environ.get("ZKHIP_BENCH_DEVICE") is not None and int(os.environ.get("WORLD_SIZE", "1") or 1) > 1
    native.default_library().init(8 if (args.members or shared_gpu) else 16)

    timeline = {}

    def mark(stage):
        """Seconds since the process started at which `stage` was complete (part of the JSON line; ZKHIP_BENCH_STAGES=1
        also prints each mark on stderr as it happens, so that a run that dies leaves the last completed stage behind)."""
        timeline[stage] = round(time.time() - T_PROCESS_START, 3)
        if os.environ.get("ZKHIP_BENCH_STAGES"):
            print("[bench] %8.3f s  %s" % (timeline[stage], stage), file=sys.stderr, flush=True)

    mark("imports")
    curve_id = synth.CURVE_IDS[args.curve]
    per_hash = None
    if args.kind == "poseidon":   # BASELINE.json configs[3]: the stdlib Poseidon hash chain, depth 1024 at a 2^18 domain
        poseidon = importlib.import_module(_pkg + ".poseidon")
        depth = 1024 << (args.log_domain - 18) if args.log_domain >= 18 else max(1, ((1 << args.log_domain) - 4) // 243)
        circ = poseidon.chain(curve_id, depth)
    elif args.kind == "sha256":   # BASELINE.json configs[0]: stdlib sha256/512bitPacked.zok, as many calls side by side as the domain holds
        sha = importlib.import_module(_pkg + ".sha256_circuit")
        per_hash = len(sha.template()[0])
        circ = sha.circuit(curve_id, max(1, (1 << args.log_domain) // (per_hash + 7)))
        args.log_domain = int(np.log2(circ.N))
    else:
        circ = synth.circuit(curve_id, args.log_domain, n=args.constraints or None, kind=args.kind)
        args.log_domain = int(np.log2(circ.N))
    mark("circuit_built")
    env_rank = int(os.environ.get("RANK", "0"))
    nw = max(1, min(args.witnesses or args.steps, args.steps + args.warmup))
    t0 = time.time()
    zs = make_witnesses(circ, [0x5EED0000 + env_rank * 1000 + i for i in range(nw)])
    t_witness = time.time() - t0
    mark("witnesses_generated")

    ranks = parallel.Ranks()          # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; nccl = RCCL
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    barrier_sync = ranks.barrier

    # the rank's GPU: LOCAL_RANK, unless the launcher already narrowed this process's view to one device
    # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) — ZKHIP_BENCH_DEVICE: test hook (all ranks on one GPU)
    ndev = native.default_library().device_count()
    if os.environ.get("ZKHIP_BENCH_DEVICE") is not None:
        device = int(os.environ["ZKHIP_BENCH_DEVICE"])
    elif local_rank < ndev:
        device = local_rank
    elif ndev == 1 and (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")):
        device = 0                   # the launcher gave this rank a view of exactly one GPU
    else:
        raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no GPU of its own: %d visible — one GPU per rank, or fewer ranks"
                         % (ranks.rank, local_rank, ndev))
    pci = native.default_library().device_pci_bus_id(device)
    placement = numa_placement(pci, pin=ranks.world > 1)      # N > 1: every rank's host threads on the NUMA node of its GPU
    ctx = native.Context(device)
    # (every resident single-GPU prover of this script asks for the plan: round 7's is level or better on all four workloads)
    stream_plan = args.pipe_plan != 0 and not args.members and not shared_gpu and os.environ.get("ZKHIP_PIPES", "1") not in ("0", "-")
    if stream_plan:
        ctx.tune("pipe_plan", 1)
    mark("context_created")
    if os.environ.get("ZKHIP_BENCH_TEST_DIE") in (os.environ.get("ZKHIP_BENCH_ATTEMPT", "-"), "*"):
        os.abort()      # tests/test_bench_cli.py: what a GPU memory fault does to the measuring process
    cs = native.ConstraintSystem(ctx, curve_id, circ.n, circ.l, circ.w, circ.mats())
    mark("r1cs_resident")
    t0 = time.time()
    gm17 = args.scheme == "gm17"
    pk_bytes = make_proving_key(ctx, cs, circ, curve_id, args.scheme)
    t_setup = time.time() - t0
    mark("setup_done")
    t0 = time.time()
    pk = native.ProvingKey(ctx, curve_id, pk_bytes, scheme=args.scheme)
    t_pkload = time.time() - t0
    mark("key_resident")
    t0 = time.time()
    resident = [native.Assignment(ctx, cs, z) for z in zs]
    t_h2d = (time.time() - t0) / nw
    mark("assignments_resident")

    def rs(i):
        if gm17:   # d1, d2, r
            return 0x1111111111111111 * (i + 1) + rank, 0x3333333333333333 * (i + 2) + rank, 0x2222222222222222 * (i + 3) + rank
        return 0x1111111111111111 * (i + 1) + rank, 0x2222222222222222 * (i + 3) + rank

    if gm17:
        prove_one = lambda a, rnd: native.prove_gm17(ctx, pk, cs, a, *rnd, want_timings=True)
        prove_many = lambda aa, rnds: native.prove_gm17_resident_batch(ctx, pk, cs, aa, rnds)
    else:
        prove_one = lambda a, rnd: (native.prove_g16_resident if isinstance(a, native.Assignment) else native.prove_g16)(ctx, pk, cs, a, *rnd, want_timings=True)
        prove_many = lambda aa, rnds: native.prove_g16_resident_batch(ctx, pk, cs, aa, rnds)
    # The resident prover's key, bound to its system (include/zkhip.h: zkhip_pk_bind_r1cs).  Self-checking: the bound key must
    # reproduce an unbound proof byte for byte, or the run goes on with the key as loaded and says so.
    bound, bound_rnd0 = {"bound": False}, None
    rnd0 = (1000, 3000, 2000) if gm17 else (1000, 2000)      # the randomness of the proof that is held against the oracle
    if args.bind and os.environ.get("ZKHIP_BENCH_BIND") == "0":
        bound["note"] = "second attempt after a measuring process that died: the key is left as loaded"
    elif args.bind:
        try:
            ref_proof = prove_one(resident[0], rs(0))[0]
            t0 = time.time()
            pk.bind(cs)
            bound["bind_ms"] = 1000 * (time.time() - t0)
            same = prove_one(resident[0], rs(0))[0] == ref_proof
            bound["proof_identical_to_unbound"] = bool(same)
            bound["bound"] = bool(same)
            if same:
                bound_rnd0 = prove_one(resident[0], rnd0)[0]   # (held against the oracle's proof below)
            else:
                pk.unbind()
        except native.ZkhipError as e:
            bound["error"] = str(e)
            try:
                pk.unbind()
            except native.ZkhipError:
                pass
        mark("key_bound")
    single = []
    # the samplers of the timed regions are MADE here, before the warm-up (the clock probe is a second context: tens of milliseconds to create), and only
    # started after it: with them made in between, the chip sat idle for ~30 ms after the warm-up and the first region — the one `value` is — ran 1-2 %
    # behind the two that follow it back to back (114.4 against 116.6 / 117.1 proofs/s in profiles/r7k_*, the same pattern in every line of the round)
    # (after the context's first proof, though: the stream plan's queues are dealt to the dispatchers in creation order, and the probe's context must not
    # make its own before them — with the key bound a proof has run already, without (--bind 0) the isolated warm-up proof below is the first)
    if args.warmup:
        _, tm1 = prove_one(resident[0], rs(0))
        single.append(tm1["total_ms"])
    sampler = LoadSampler(pci)
    idle_reading = sampler.read_once()
    clock = ClockProbe(device)            # a second context's one-wave probe: the shader clock the proving kernels actually run at
    if args.warmup:
        # The W warm-up steps run the way the timed steps do — through the pipelined batch call — so that the timed region starts
        # on a chip in the state it will be measured in (clocks and power ramp over the first tens of milliseconds of load: with
        # W isolated proofs as warm-up the first of three identical regions was 1-2.5 % slower than the third,
        # profiles/r5c_bench_driver_command.json).  Every proof slot the library can keep in flight (ZK_NSLOTS = 4) allocates its
        # workspaces the first time it is used, so never fewer than 4 proofs; one isolated proof first (above; its latency is reported).
        nwarm = max(args.warmup, 4)
        prove_many([resident[i % nw] for i in range(nwarm)], [rs(100 + i) for i in range(nwarm)])
    steps = [args.warmup + i for i in range(args.steps)]
    mark("warmup_done")
    sampler.start()
    clock.start()
    barrier_sync()
    t_begin = time.perf_counter()
    proofs, acc = prove_many([resident[j % nw] for j in steps], [rs(j) for j in steps])
    barrier_sync()
    elapsed_local = time.perf_counter() - t_begin
    mark("timed_region_done")
    elapsed = ranks.max_over_ranks(elapsed_local)
    # the same region again (`value` stays the first one): how far a number taken over K x 10 ms can be trusted
    region_ms = [1000.0 * elapsed / args.steps]
    for rep in range(1, max(1, args.repeats)):
        barrier_sync()
        t0 = time.perf_counter()
        prove_many([resident[j % nw] for j in steps], [rs(j + 1000 * rep) for j in steps])
        barrier_sync()
        region_ms.append(1000.0 * ranks.max_over_ranks(time.perf_counter() - t0) / args.steps)
    sampler.stop()
    clock_regions = clock.window(t_begin, time.perf_counter())
    mark("repeats_done")
    per_rank = gather_per_rank(ranks, device, pci, placement, args.steps, elapsed_local)
    # isolated single-proof latency (not part of the timed region): resident assignment, then from host memory
    # (a lone proof's latency moves by +-0.4 ms from proof to proof — which streams' kernels meet on a dispatcher — and a minimum of three
    # moved by 0.6 ms between processes of one build (profiles/r7r_*, r7s_*): sixteen proofs, the minimum as before and the spread beside it)
    lone = []
    for i in range(16):
        _, tm1 = prove_one(resident[i % nw], rs(200 + i))
        lone.append(tm1["total_ms"])
    single.extend(lone)
    single_ms = min(single)
    lone.sort()
    single_stats = {"proofs": len(lone), "min": lone[0], "median": lone[len(lone) // 2], "p90": lone[(9 * len(lone)) // 10], "max": lone[-1]}
    from_host = []
    for i in range(3):
        t0 = time.perf_counter()
        prove_one(zs[i % nw], rs(300 + i))
        from_host.append(1000.0 * (time.perf_counter() - t0))
    # one-stream leg: the same proofs with every kernel on ONE stream -> un-overlapped kernel durations
    serial = None
    clock_serial = None
    if args.serial_proofs > 0:
        ctx.tune("serial", 1)
        t_s0 = time.perf_counter()
        tms = [prove_one(resident[i % nw], rs(400 + i))[1] for i in range(args.serial_proofs + 1)][1:]
        clock_serial = clock.window(t_s0, time.perf_counter())
        ctx.tune("serial", 0)
        serial = {k: sum(t[k] for t in tms) / len(tms) for k in tms[0]}
    clock.stop()

    # the same region once more with the key UNBOUND (six transforms, three mat-vecs): the same-box figure the binding is worth
    if bound["bound"] and args.bind == 1 and args.repeats > 1:          # (--bind 2: bound throughout — a trace of the bound pipeline only)
        pk.unbind()
        prove_many([resident[i % nw] for i in range(4)], [rs(500 + i) for i in range(4)])
        barrier_sync()
        t0 = time.perf_counter()
        prove_many([resident[j % nw] for j in steps], [rs(j + 7000) for j in steps])
        barrier_sync()
        bound["unbound_ms_per_step"] = 1000.0 * ranks.max_over_ranks(time.perf_counter() - t0) / args.steps
        bound["unbound_single_proof_ms"] = min(prove_one(resident[i % nw], rs(600 + i))[1]["total_ms"] for i in range(3))
        bound["note"] = ("`value` is measured with the key bound to the constraint system (Groth16: 4 transforms + 2 mat-vecs per proof; GM17: 2 transforms); "
                         "unbound_* repeat the region and the isolated proof with the key as loaded (6 transforms + 3 mat-vecs, GM17 4; the reference's "
                         "schedules have 7 and 5)")
        mark("unbound_region_done")

    value_first = world * args.steps / elapsed
    if bound["bound"]:
        value_bound, value_unbound = value_first, (world * 1000.0 / bound["unbound_ms_per_step"]) if bound.get("unbound_ms_per_step") else None
        key_state = "bound to the constraint system (zkhip_pk_bind_r1cs: one-time bind_ms outside the timed region, two extra tables resident)"
    else:
        value_bound, value_unbound = None, value_first
        key_state = "as loaded" + (" — FALLBACK: " + bound["note"] if os.environ.get("ZKHIP_BENCH_BIND") == "0" and args.bind else "")
    avg = {k: v / args.steps for k, v in acc.items()}
    # ---- roofline of the dominant kernel (HIP events on the library's streams)
    m, N = circ.m, circ.N
    if gm17:   # the MSMs run over the SAP variables and the SAP domain
        m, N = pk.m, pk.hlen - 1
    fq = native.FQ_BYTES[curve_id]
    kernels = {
        # algorithmic bytes per proof: bases read once + the 32-B scalars they pair with (SURVEY.md §8d (iv)+(v)); launches
        "msm_accumulate<G2> (b_g2_query)": ("kernel_msm_accum_g2_ms", (m + 2) * (4 * fq + 32), 1),
        # (a, b_g1 and l share the sorted assignment and run as ONE launch, h_query as another; ZKHIP_FUSE_Z=0: four)
        "msm_accumulate<G1> (a/b_g1/l/h_query)": ("kernel_msm_accum_g1_ms", (3 * (m + 2) + N) * (2 * fq + 32), 4 if os.environ.get("ZKHIP_FUSE_Z") == "0" else 2),
    }
    name, (key, bytes_all, launches) = max(kernels.items(), key=lambda kv: avg[kv[1][0]])
    ms = avg[key]
    achieved = bytes_all / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) — measured offline, same workload
    traffic, traffic_ntt, traffic_cal = None, None, None
    default_workload = args.log_domain == 20 and args.curve == "bn128" and args.kind == "dense" and not gm17
    evidence = offline_evidence() if default_workload else {"traffic": None, "valu": None, "stale": False}
    pmc = evidence["traffic"]
    if pmc:
        ent = pmc.get("G2" if "G2" in name else "G1", {})
        traffic = ent.get("traffic_bytes_per_launch")
        traffic_cal = {"fetch_factor": ent.get("fetch_factor"), "pattern": ent.get("pattern"), "calibration": pmc.get("calibration")}
        if traffic and "G1" in name:
            # the file averages over the accumulation launches of its run; re-average if that run cut a proof's G1 work
            # into a different number of launches than this build (the per-proof total is what was measured)
            lpp = ent.get("launches_per_proof") or (ent["launches_fetch_pass"] / pmc["NTT"]["proofs"] if pmc.get("NTT", {}).get("proofs") else launches)
            traffic = int(traffic * lpp / launches)
        traffic_ntt = pmc.get("NTT", {}).get("traffic_bytes_per_pass")
    # The honest bound of this kernel is VALU ISSUE (Montgomery products are multiply-adds; one wave64 instruction per 4 cycles and SIMD):
    #   peak = the kernel's own VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU of this build, profiles/pmc_valu.json)
    #          issued by 1 024 SIMDs at the shader clock MEASURED IN THIS RUN beside the kernels (zkhip_ctx_clock_probe)
    # expressed as mixed additions per second (one per non-zero signed digit, W digits per scalar).
    W = msm_windows(synth.FR_BITS[curve_id], m + 2) if args.log_domain >= 15 else None
    compute = None
    pv = evidence["valu"]
    if W and args.curve == "bn128" and args.kind == "dense":   # (sparse / boolean witnesses drop their zero digits: no fixed addition count)
        madds = (((m + 2) if "G2" in name else (3 * (m + 2) + N)) * W)
        group = "G2" if "G2" in name else "G1"
        ghz = clock_regions.get("mean_ghz") or (pv or {}).get(group, {}).get("clock_ghz")
        ghz_serial = (clock_serial or {}).get("mean_ghz") or ghz
        ent = (pv or {}).get(group)
        static = STATIC_VALU_PER_ADDITION.get(group)
        # wave-instructions of one proof's launches of this kernel: counted (fresh counter file) or, failing that, the static count of the
        # hot loop (profiles/r5_accum_isa_mix.txt) times the additions
        instr = ent["valu_wave_instructions_per_launch"] * launches if ent else (madds * static / 64.0 if static else None)
        compute = {"unit": "mixed additions/s", "achieved": madds / (ms * 1e-3), "windows": W,
                   "valu_wave_instructions_per_proof": instr,
                   "instructions_source": ("rocprofv3 SQ_INSTS_VALU of this build (profiles/pmc_valu.json: OFFLINE, fingerprinted)" if ent else
                                           "static count of the hot loop x additions (no fresh counter file for this build)"),
                   "shader_clock_ghz": ghz, "shader_clock_ghz_serial_leg": ghz_serial,
                   "clock_source": "zkhip_ctx_clock_probe: one wavefront beside the kernels, shader cycles / wall clock (LIVE)" if clock_regions.get("mean_ghz") else
                                   "derived from the counter pass (no live probe)",
                   "kernel_on_synthetic_lists": {"additions_per_s": 5.57e9 if "G2" in name else 13.4e9, "source": "tools/accum_bench.hip, profiles/r5a_accum_bench.txt (round 5's `peak`: the kernel against itself)"}}
        if instr and ghz:
            t_issue_ms = instr / (1024 * ghz * 1e9 / 4) * 1e3                  # per proof, this kernel alone at the issue limit
            compute["peak"] = madds / (t_issue_ms * 1e-3)
            compute["peak_source"] = "VALU issue limit: 1024 SIMDs x shader clock / 4 cycles per wave64 instruction / the kernel's instructions per proof"
            compute["frac"] = compute["achieved"] / compute["peak"]
            if serial and serial[key] > 0 and ghz_serial:
                compute["frac_serial"] = (instr / (1024 * ghz_serial * 1e9 / 4) * 1e3) / serial[key]
        if world == 1 and not gm17 and pv:
            compute["pipeline_issue_bound"] = pipeline_issue_bound(pv, 1000.0 * elapsed / args.steps, 8 if bound["bound"] else 12, ghz)
    roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_ratio": (traffic / (bytes_all / launches)) if traffic else None,
                "traffic_calibration": traffic_cal,
                "traffic_source": ("offline rocprofv3 --pmc passes of this workload, " + PMC_TRAFFIC_FILE) if traffic else None,
                "frac_serial": (bytes_all / (serial[key] * 1e-3) / 1e9 / HBM_PEAK_GBS) if serial and serial[key] > 0 else None,
                "bytes_per_launch": bytes_all / launches, "ms_per_launch": ms / launches,
                "ms_per_launch_serial": serial[key] / launches if serial else None, "launches_per_proof": launches,
                "compute_bound": compute,
                "offline_evidence": {k: evidence.get(k) for k in ("stale", "why", "csrc_hash", "files") if evidence.get(k) is not None},
                "note": "bucket accumulation is bound by integer-multiply issue (Montgomery products), not by HBM; `frac` uses HIP-event "
                        "intervals on the MSM streams inside the timed region (five MSMs and two proofs overlap), `frac_serial` the same "
                        "kernels alone on one stream after it"}
    # the NTT passes: 2 * N * 32 B per pass and vector; Groth16 runs 12 pass-vectors per proof (6 transforms: c
    # needs only its coefficients; the reference runs 7), GM17 8 (4 transforms; the reference 5); the interval also
    # holds the pointwise quotient kernel
    passes = 8 if (gm17 or bound["bound"]) else 12
    if N <= 1 << 10:
        passes //= 2
    ntt_bytes = passes * 2 * N * 32
    roofline_ntt = {"bound": "hbm", "kernel": "ntt_cols / ntt_rows (%d pass-vectors of 2^%d elements per proof)" % (passes, int(np.log2(N))),
                    "achieved": ntt_bytes / (avg["kernel_ntt_ms"] * 1e-3) / 1e9 if avg.get("kernel_ntt_ms", 0) > 0 else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic_ntt, "traffic_ratio": (traffic_ntt / (2.0 * N * 32)) if traffic_ntt else None,
                    "traffic_source": ("offline rocprofv3 --pmc passes of this workload, " + PMC_TRAFFIC_FILE) if traffic_ntt else None,
                    "bytes_per_pass": 2 * N * 32}
    if roofline_ntt["achieved"]:
        roofline_ntt["frac"] = roofline_ntt["achieved"] / HBM_PEAK_GBS
        roofline_ntt["us_per_pass"] = 1000.0 * avg["kernel_ntt_ms"] / passes
    if serial and serial.get("kernel_ntt_ms", 0) > 0:
        roofline_ntt["frac_serial"] = ntt_bytes / (serial["kernel_ntt_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        roofline_ntt["us_per_pass_serial"] = 1000.0 * serial["kernel_ntt_ms"] / passes
    b_alg = gm17_algorithmic_bytes(circ, fq, m, N) if gm17 else proof_algorithmic_bytes(circ, fq)
    workload = (f"synthetic R1CS {args.kind}, n = 2^{args.log_domain} - 2 constraints, {args.curve} GM17 (SAP: {m} variables, domain {N}), "
                f"4 NTTs + 5 MSMs per proof") if gm17 else (
        (f"Poseidon hash chain depth {circ.depth} (t = 3, 243 constraints per hash), n = {circ.n} constraints (QAP domain 2^{args.log_domain}), "
         if args.kind == "poseidon" else
         f"{circ.hashes} x a RESTATEMENT of stdlib sha256/512bitPacked.zok (zokrates_amd/sha256_circuit.py walks the program through the reference's "
         f"uint-optimizer / flattener / redefinition rules: {per_hash} constraints per call by that restatement, never compared with a compiled `out` — "
         f"no compiler here; the function is pinned on hashlib and the reference's known answer, the R1CS shape is not), "
         f"n = {circ.n} constraints (QAP domain 2^{args.log_domain}), "
         if args.kind == "sha256" else
         f"synthetic R1CS {args.kind}, n = {circ.n} constraints (QAP domain 2^{args.log_domain})"
         + (" [stand-in for BASELINE configs[0], stdlib sha256/512bitPacked.zok: the ZoKrates compiler cannot run here, so the wire "
            "statistics of a SHA-256 circuit (90 % boolean) are generated directly], " if args.kind == "sha" else ", "))
        + f"{args.curve} Groth16, " + ("4 NTTs + 5 MSMs per proof (key bound to the constraint system)" if bound["bound"] else "6 NTTs + 5 MSMs per proof"))
    out = {
        "metric": "gm17_proofs_per_sec" if gm17 else "groth16_proofs_per_sec", "value": value_first, "unit": "proofs/s",
        "value_is": "value_bound" if bound["bound"] else "value_unbound", "value_bound": value_bound, "value_unbound": value_unbound, "key_state": key_state,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": workload, "curve": args.curve, "constraints": circ.n,
                   "variables": m, "domain": N, "distinct_witnesses": nw,
                   "parallelism": f"{world} independent prover(s), full key per GPU",
                   "process_group": ranks.describe(),
                   "launcher": ("bench.py --gpus N started the ranks itself" if os.environ.get("ZKHIP_BENCH_LAUNCHED") == "self" else
                                "external launcher (torch.distributed.run)" if world > 1 else "single process")},
        "single_proof_ms": single_ms, "single_proof_unbound_ms": bound.get("unbound_single_proof_ms"),
        "single_proof_ms_stats": single_stats, "single_proof_from_host_ms": min(from_host), "phases_ms": avg, "phases_ms_serial": serial,
        "whole_proof_hbm": {"algorithmic_bytes": b_alg, "achieved_GBs": b_alg / (elapsed / args.steps) / 1e9,
                            "frac": b_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        "roofline": roofline, "roofline_ntt": roofline_ntt,
        "host_ms": {"setup_gpu": 1000 * t_setup, "pk_load": 1000 * t_pkload, "assignment_h2d": 1000 * t_h2d,
                    "witness_generation_total": 1000 * t_witness},
        "timeline_s": timeline,
        "device": ctx.describe(),
        "repeats": {"regions": len(region_ms), "ms_per_step": region_ms, "median_ms_per_step": float(np.median(region_ms)),
                    "spread": (max(region_ms) - min(region_ms)) / float(np.median(region_ms)),
                    "value_of_median": world * 1000.0 / float(np.median(region_ms)),
                    "note": "`value` / `ms_per_step` are the FIRST region; every region is K steps over the same K resident witnesses"},
        "under_load": sampler.summary(idle_reading),
        "shader_clock": {"timed_regions": clock_regions, "serial_leg": clock_serial,
                         "source": "zkhip_ctx_clock_probe on a second context of the same device, 2 ms readings back to back (shader cycles / wall clock of one sleeping wavefront)"},
        "per_rank": per_rank,
        "bound_key": bound,
        "stream_plan": {"on": stream_plan, "plan": os.environ.get("ZKHIP_PIPES") or ("M=2,N=3,O=1,n=3,G0=0,Z0=1,H0=0,G1=3,Z1=1,H1=0,G2=3,Z2=2,H2=1" if stream_plan else None),
                        "note": "the prover's streams placed on the GPU's four dispatchers (zkhip_ctx_tune PIPE_PLAN, core.cuh make_pipe_streams): same "
                                "kernels, same proofs; the plan comes from a local search scored on four workloads, batches and lone proofs (tools/plan_search.py, "
                                "profiles/r7f_*, r7h_*); against streams made in order of first use (profiles/r7g_*, r7h_*): the dense circuit level in batches and a lone "
                                "proof 0.25-0.3 ms sooner, stdlib SHA-256 +8-9 %, Poseidon / BLS12-381 +3 %, GM17 level — every leg of this line asks for it"},
    }
    # ---- optional legs (latency mode): a watchdog guarantees that the throughput line is printed even if one of them hangs
    # (a collective after an asymmetric failure; the in-library path on hardware this container cannot test)
    import threading
    members = args.members or (world if world > 1 else 0)

    def give_up():
        if rank == 0:
            note = {"error": "timed out after %d s" % SHARDED_LEG_TIMEOUT_S}
            if world > 1:
                out.setdefault("sharded_single_proof", note)
            if members >= 1:
                out.setdefault("multi_single_proof", note)
            out["cpu_baseline"] = None
            print(json.dumps(out), flush=True)
        os._exit(0)

    watchdog = threading.Timer(SHARDED_LEG_TIMEOUT_S, give_up)
    watchdog.daemon = True
    if world > 1 or members >= 1:
        watchdog.start()
    if world > 1:
        # ONE proof sharded over all ranks (1/world of the bases per GPU, RCCL all-gather of the partial records);
        # reported next to the throughput metric, never instead of it
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
    if os.environ.get("ZKHIP_BENCH_TEST_STALL"):      # tests/test_bench_cli.py: an optional leg that never returns
        time.sleep(3600)
    if rank == 0 and members >= 1:
        out["multi_single_proof"] = multi_leg(ctx, circ, curve_id, pk_bytes, zs[0], members, gm17, prove_one,
                                              oracle=(lambda rnd: trapdoor_proof(circ, curve_id, zs[0], rnd, gm17)) if args.oracle == "trapdoor" else None)
    if world > 1:
        ranks.host_barrier()      # the other ranks idle (on the host: no collective kernel parked on their GPUs) while rank 0
                                  # drives every GPU through the library
    watchdog.cancel()
    # (N > 1 has no cpu_baseline leg: rank 0 holds its proof to the closed form instead — a second of host arithmetic after the timed region)
    oracle_kind = args.oracle if args.oracle != "auto" else ("algorithmic" if (world == 1 and args.cpu_seconds > 0) else "trapdoor" if world > 1 else "none")
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        base, cpu_proof = cpu_baseline(circ, pk_bytes, zs[0], args.cpu_seconds, gm17)
        out["cpu_baseline"] = base
        # same inputs -> byte-identical proof (the CPU leg doubles as a full-size parity check)
        gpu_proof = prove_one(resident[0], rnd0)[0]
        batch_proof = prove_many([resident[0]] * 3, [rnd0] * 3)[0]
        out["cpu_baseline"]["gpu_proof_identical"] = bool(gpu_proof == cpu_proof and all(p == cpu_proof for p in batch_proof))
        if bound_rnd0 is not None:
            out["cpu_baseline"]["gpu_bound_key_proof_identical"] = bool(bound_rnd0 == cpu_proof)
        # like for like: the CPU baseline runs the reference's schedule on the key as loaded, so the ratio is quoted on the UNBOUND region
        # when this run has one (the bound figure amortises a one-time bind of `bind_ms` over the prover's lifetime)
        out["speedup_vs_cpu_baseline"] = (out.get("value_unbound") or out["value"]) / base["value"]
        out["speedup_vs_cpu_baseline_is"] = "value_unbound / cpu_baseline.value" if out.get("value_unbound") else "value / cpu_baseline.value (no unbound region in this run)"
        if oracle_kind == "algorithmic":
            out["identical_to_oracle"] = bool(out["cpu_baseline"]["gpu_proof_identical"] and (bound_rnd0 is None or bound_rnd0 == cpu_proof))
            out["oracle"] = "algorithmic restatement of ark's prover (oracle/c), the cpu_baseline leg's proof"
    elif rank == 0:
        out["cpu_baseline"] = None   # N > 1 or --cpu-seconds 0
    if rank == 0 and oracle_kind == "trapdoor":
        # the closed form from the toxic waste of this run's own setup (oracle/c orc_trapdoor / orc_gm17_trapdoor): no transform, no MSM
        try:
            t0 = time.time()
            want = trapdoor_proof(circ, curve_id, zs[0], rnd0, gm17)
            gpu_proof = prove_one(resident[0], rnd0)[0]
            batch_proof = prove_many([resident[0]] * 3, [rnd0] * 3)[0]
            ok = gpu_proof == want and all(p == want for p in batch_proof) and (bound_rnd0 is None or bound_rnd0 == want)
            out["identical_to_oracle"] = bool(ok)
            out["oracle"] = "closed-form trapdoor proof (oracle/c), %.1f s" % (time.time() - t0)
        except Exception as e:
            out["identical_to_oracle"] = None
            out["oracle"] = "trapdoor check failed to run: %r" % (e,)
    if rank == 0 and world == 1 and args.e2e:
        out["cli_end_to_end_ms"] = cli_end_to_end(circ, curve_id, pk_bytes, zs[0], args.scheme, ctx, pk, cs)
    if rank == 0 and world == 1:
        out["box_probe"] = box_probe()
        out["rocm_smi"] = rocm_smi()
    want_configs = args.configs == 1 or (args.configs == -1 and default_workload and world == 1 and not os.environ.get("ZKHIP_BENCH_LEG"))
    if rank == 0 and want_configs and os.environ.get("ZKHIP_BENCH_ATTEMPT") is not None:
        out["configs"] = CONFIGS_DEFERRED    # (a supervised run: the parent runs the legs once this process has released the device)
    elif rank == 0 and want_configs:
        del resident[:]                      # this process's share of the device: the legs are processes of their own
        for obj in (pk, cs, ctx):            # (the context too: its sixteen idle hardware queues beside a leg's own cost the leg 5-8 % — legs read
            try:                             # 177-185 / 295-306 / 81-83 proofs/s where the same commands alone read 195-201 / 314-330 / 86-87: r7q, r8f)
                obj.close()
            except Exception:
                pass
        out["configs"] = config_legs(args)   # LAST key of the line on purpose: it survives a reader that keeps the tail
    if rank == 0:
        print(json.dumps(out), flush=True)
    ranks.close()


def trapdoor_proof(circ, curve_id, z, rnd, gm17):
    """The proof the toxic waste of make_proving_key's setup determines in closed form (oracle/c: three fixed-base products of
    field expressions — no transform, no MSM; test infrastructure, used here as the checker only)."""
    from oracle import cpu
    oc = cpu.Circuit.from_csr(curve_id, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(curve_id, 0xC0FFEE)
    if gm17:
        tb = b"".join(int(v).to_bytes(32, "little") for v in (tox[0], tox[1], tox[2], tox[4]))
        return cpu.gm17_trapdoor(oc, tb, z, rnd[0], rnd[2])
    tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
    return cpu.trapdoor(oc, tb, z, rnd[0], rnd[1])


def cli_end_to_end(circ, curve_id, pk_bytes, z, scheme, ctx, pk, cs):
    """The reference's own use of the path: `zokrates generate-proof` reads the compiled program, the witness and the proving
    key from files and writes proof.json, ONE proof per process (/root/reference/zokrates_cli/src/ops/generate_proof.rs:152-202).
    Here: the same three files for the benchmark circuit (ZoKrates' `out` / `witness` formats, ark's proving.key) in a RAM-backed
    directory, `python -m zokrates_amd.cli generate-proof` as a fresh process per run — from the proving.key, and from the
    device-layout key image a first run leaves behind (level 0 of the base tables) —
    wall clock of the process and the split it reports (the program is decoded on host threads while the key is uploaded).
    The steady-state numbers above are what a resident prover service gets; this is what the CLI user gets."""
    import shutil
    import subprocess
    import tempfile
    formats, rng = (importlib.import_module(_pkg + "." + m) for m in ("formats", "rng"))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="zkhip_e2e_", dir=base)
    res = {"directory": d}
    try:
        t0 = time.perf_counter()
        ids = np.arange(circ.m, dtype=np.int64)
        prog_bytes = native.write_program(curve_id, circ.n, circ.m, circ.mats(), ids=ids, args=[(j, False) for j in range(1, circ.l)])
        chk = native.Program(prog_bytes)       # the reader allocates columns as ark's generate_constraints does; the key was made for circ's
        in_order = bool((chk.variable_order() == ids).all())
        del chk
        if not in_order:
            res["error"] = ("the generated system's columns are not in generate_constraints order: the key of this run would not fit the program "
                            "file (kinds dense, poseidon and sha256 are; the statistical 'sha' stand-in is not)")
            return res
        paths = {k: os.path.join(d, k) for k in ("out", "witness", "proving.key", "proof.json", "cache")}
        prog_bytes.tofile(paths["out"])
        native.write_witness(ids, z).tofile(paths["witness"])
        np.asarray(pk_bytes, dtype=np.uint8).tofile(paths["proving.key"])
        res["write_input_files_ms"] = 1000.0 * (time.perf_counter() - t0)
        res["file_bytes"] = {k: os.path.getsize(paths[k]) for k in ("out", "witness", "proving.key")}
        # the proof the CLI must write: the same (r, s) drawn from the same entropy, proved in this process
        gen = rng.rng_from_entropy("bench")
        if scheme == "gm17":
            rnd = tuple(rng.fr_rand(gen, curve_id) for _ in range(3))
            raw = native.prove_gm17(ctx, pk, cs, z, *rnd)
        else:
            rnd = tuple(rng.fr_rand(gen, curve_id) for _ in range(2))
            raw = native.prove_g16(ctx, pk, cs, z, *rnd)
        inputs = [int.from_bytes(z[32 * j:32 * j + 32].tobytes(), "little") for j in range(1, circ.l)]
        want = formats.proof_json(curve_id, raw, inputs, scheme=scheme)

        with open(os.path.join(d, "expected_proof.json"), "w") as fh:
            fh.write(want)
        state = {"directory": d, "scheme": scheme, "constraints": int(circ.n), "write_input_files_ms": res["write_input_files_ms"], "file_bytes": res["file_bytes"]}
    except Exception as e:   # the throughput line must survive a failure of this leg
        shutil.rmtree(d, ignore_errors=True)
        return {"directory": d, "error": repr(e)}
    # A supervised run hands the files to the SUPERVISING process, which starts the `generate-proof` processes once this one has released the device
    # (beside a process that holds it a leg pays for that: the configuration legs read 3-5 % low as this process's children, profiles/r8h_*).
    if os.environ.get("ZKHIP_BENCH_ATTEMPT") is not None:
        return {CLI_DEFERRED: state}
    return cli_run(state)


def multi_leg(ctx, circ, curve_id, pk_bytes, z, members, gm17, prove_one, oracle=None):
    """ONE proof across `members` members inside the library (zkhip_ctx_create_multi / zkhip_prove_*_multi: a host thread per
    member, canonical partial records gathered in host memory — no Python, no collective library): the latency mode a caller
    behind the reference's trait gets.  First with the members' shards as loaded, then bound to the system (zkhip_multi_bind: one
    member computes the bound bases, every member installs its index ranges) — `ms` is the bound figure when the binding took."""
    try:
        ndev = ctx.lib.device_count()
        devices = [k % ndev for k in range(members)]
        multi = native.Multi(devices, ctx.lib)
        if len(set(devices)) == len(devices):          # one GPU per member: the exchange step runs over RCCL (xGMI)
            try:
                multi.use_rccl(True)
            except native.ZkhipError:
                pass                                   # (no loadable librccl: the host exchange stays in use)
        multi.load_constraint_system(curve_id, circ.n, circ.l, circ.w, circ.mats())
        t0 = time.time()
        multi.load_proving_key(curve_id, pk_bytes, scheme="gm17" if gm17 else "g16")
        t_load = time.time() - t0
        rnd = (31, 59, 4242) if gm17 else (4242, 777)
        whole = prove_one(z, rnd)[0]

        def timed(count=5):
            times, phases, proof = [], None, None
            for i in range(count):
                t0 = time.perf_counter()
                proof, phases = (multi.prove_gm17 if gm17 else multi.prove_g16)(z, *rnd, want_timings=True)
                times.append(1000.0 * (time.perf_counter() - t0))
            return min(times[1:]), phases, proof
        ms_unbound, phases_unbound, proof = timed()
        same = proof == whole
        rec = {"members": members, "devices": devices, "distinct_gpus": len(set(devices)), "key_load_ms": 1000.0 * t_load,
               "ms_unbound": ms_unbound, "ms": ms_unbound, "key_bound": False}
        try:
            t0 = time.time()
            multi.bind(pk_bytes)
            rec["bind_ms"] = 1000.0 * (time.time() - t0)
            ms_bound, phases, proof_b = timed()
            same = same and proof_b == whole
            rec.update({"ms": ms_bound, "key_bound": True, "slowest_member_phases_ms": phases, "kernel_ntt_ms": phases.get("kernel_ntt_ms"),
                        "kernel_ntt_ms_unbound": phases_unbound.get("kernel_ntt_ms")})
            multi.unbind()
        except native.ZkhipError as e:
            rec["bind_error"] = str(e)
            rec["slowest_member_phases_ms"] = phases_unbound
        rec["identical_to_unsharded"] = bool(same)
        if oracle is not None:
            rec["identical_to_oracle"] = bool(same and whole == oracle(rnd))
        replicas = None
        if not gm17:
            # throughput mode of the same members: whole key on each, independent proofs dealt over them
            multi.load_proving_key_replicas(curve_id, pk_bytes)
            count = 8 * members
            zs, rss = [z] * count, [(4242 + i, 777 + i) for i in range(count)]
            multi.prove_g16_batch(zs[:members], rss[:members])
            t0 = time.perf_counter()
            proofs, _ = multi.prove_g16_batch(zs, rss)
            dt = time.perf_counter() - t0
            replicas = {"proofs": count, "proofs_per_s": count / dt, "first_identical_to_unsharded": bool(proofs[0] == whole)}
        rec["replicas_batch"] = replicas
        rec["exchange"] = multi.exchange()
        multi.close()
        return rec
    except Exception as e:   # the throughput line must survive a failure of the optional leg
        return {"error": repr(e)}


def msm_windows(scalar_bits, n):
    """Windows per scalar of a resident key's MSMs — the rule of csrc/core.cuh msm_shape(table = true): the fewest windows the widest
    admissible width (17 bits, or log2 n + 1) gives.  254-bit scalars: 15 (17-bit windows); 255-bit: 16."""
    lg = max(int(n), 1).bit_length() - 1
    cmax = max(2, min(17, lg + 1))
    return (scalar_bits + 1 + cmax - 1) // cmax


def numa_placement(pci, pin):
    """The NUMA node of the GPU at PCI address `pci` (/sys/bus/pci/devices/<pci>/numa_node) and, with `pin`, this process bound to
    that node's CPUs (os.sched_setaffinity: the staging memcpy of an assignment is the host work that scales with the circuit).
    Returns what happened, for the line's per_rank list."""
    rec = {"pci": pci, "numa_node": None, "pinned": False}
    try:
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % pci).read().strip())
        rec["numa_node"] = node
        if pin and node >= 0 and not os.environ.get("ZKHIP_BENCH_NO_PIN"):
            cpus = set()
            for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
                rec["pinned"] = True
        rec["cpus_allowed"] = len(os.sched_getaffinity(0))
    except Exception as e:      # no such file (emulator, container without sysfs): nothing to pin to
        rec["note"] = repr(e)[:120]
    return rec


def gather_per_rank(ranks, device, pci, placement, steps, elapsed_local):
    """One record per rank — rank, device ordinal, PCI address, NUMA node, its OWN proofs/s and ms per step of the first timed
    region (the line's `value` uses the slowest rank's time) — gathered over the process group: with N ranks the list has N entries."""
    mine = np.array([ranks.rank, device, placement.get("numa_node") if placement.get("numa_node") is not None else -2,
                     1.0 if placement.get("pinned") else 0.0, elapsed_local], dtype=np.float64)
    label = (pci or "?").encode()[:16].ljust(16, b" ")
    blob = np.frombuffer(mine.tobytes() + label, dtype=np.uint8)
    out = []
    for rec in ranks.all_gather_bytes(blob):
        raw = bytes(np.asarray(rec, dtype=np.uint8))
        v = np.frombuffer(raw[:40], dtype=np.float64)
        out.append({"rank": int(v[0]), "device": int(v[1]), "pci": raw[40:56].decode(errors="replace").strip(),
                    "numa_node": None if v[2] == -2 else int(v[2]), "pinned_to_numa_node": bool(v[3]),
                    "value": steps / v[4] if v[4] > 0 else None, "ms_per_step": 1000.0 * v[4] / steps})
    return out


class LoadSampler:
    """Clocks, power and temperatures of the rank's GPU WHILE the timed regions run: a host thread reads the amdgpu hwmon files of
    the device (found through its PCI address) every 25 ms.  The prover's calls release the GIL, so the thread runs
    beside them; a read is a few tens of microseconds of host time.  rocm-smi after the run only ever showed an idle chip."""
    FILES = (("power_w", ("power1_average", "power1_input"), 1e-6), ("sclk_mhz", ("freq1_input",), 1e-6), ("mclk_mhz", ("freq2_input",), 1e-6),
             ("temp_edge_c", ("temp1_input",), 1e-3), ("temp_junction_c", ("temp2_input",), 1e-3), ("temp_mem_c", ("temp3_input",), 1e-3))

    def __init__(self, pci, period_s=0.025):      # (every 4 ms, as until session r8k, the reads themselves cost the first region ~1 %: profiles/r8k_*)
        import glob
        import threading
        self.period = float(os.environ.get("ZKHIP_BENCH_SAMPLER_PERIOD_S", period_s))      # (experiment hook; <= 0: no sampling)
        self.paths = {}
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        base = "/sys/bus/pci/devices/%s" % pci if pci else None
        if base and os.path.isdir(base):
            for hw in sorted(glob.glob(base + "/hwmon/hwmon*")):
                for key, names, scale in self.FILES:
                    for nme in names:
                        f = os.path.join(hw, nme)
                        if key not in self.paths and os.path.exists(f):
                            self.paths[key] = (f, scale)
            busy = os.path.join(base, "gpu_busy_percent")
            if os.path.exists(busy):
                self.paths["gpu_busy_percent"] = (busy, 1.0)

    def read_once(self):
        rec = {}
        for key, (f, scale) in self.paths.items():
            try:
                with open(f) as fh:
                    rec[key] = float(fh.read().strip()) * scale
            except Exception:
                pass
        return rec

    def start(self):
        import threading
        if not self.paths or self.period <= 0:
            return

        def loop():
            while not self._stop.is_set():
                self.samples.append(self.read_once())
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)

    def summary(self, idle):
        if not self.paths:
            return {"source": None, "note": "no amdgpu hwmon files for this device (no sysfs view of the GPU here)"}
        out = {"source": "amdgpu hwmon (sysfs) of the rank's GPU, sampled every %.0f ms on a host thread across the timed regions" % (1000 * self.period),
               "samples": len(self.samples), "idle_before": idle}
        for key in self.paths:
            vals = [smp[key] for smp in self.samples if key in smp]
            if vals:
                out[key] = {"min": min(vals), "mean": sum(vals) / len(vals), "max": max(vals)}
        return out


def offline_evidence(root=ROOT, pkg=None):
    """The committed counter files (profiles/pmc_traffic.json, pmc_valu.json: rocprofv3 PMC passes, taken offline) — but only if
    they were taken from THIS build: each file carries the fingerprint of the sources it ran on (zokrates_amd.build.csrc_hash,
    stamped by tools/pmc_traffic.py / pmc_valu.py on the GPU box).  A file without a fingerprint or with another one is stale:
    its figures are left out of the line and "stale" says why — a kernel change that forgets to refresh the passes can no
    longer print old evidence next to a new time."""
    build = importlib.import_module((pkg or _pkg) + ".build")
    try:
        now = build.csrc_hash(root if root != ROOT else None)
    except Exception as e:
        return {"traffic": None, "valu": None, "stale": True, "why": "cannot fingerprint the sources: %r" % (e,)}
    out = {"traffic": None, "valu": None, "stale": False, "csrc_hash": now, "files": {}}
    for key, rel in (("traffic", PMC_TRAFFIC_FILE), ("valu", os.path.join("profiles", "pmc_valu.json"))):
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                doc = json.load(f)
        except Exception:
            continue
        have = doc.get("csrc_hash")
        out["files"][rel] = have
        if have == now:
            out[key] = doc
        else:
            out["stale"] = True
    if out["stale"]:
        out["why"] = "counter file(s) taken from another build (csrc_hash differs): refresh with tools/gpu_final.sh / gpu_pmc_valu.sh"
    return out


def pipeline_issue_bound(pv, ms_per_step, pass_vectors=12, clock_ghz=None):
    """The proof rate against the issue limit of the pipeline's own instruction stream: the VALU wavefront instructions the
    committed counter pass (profiles/pmc_valu.json) counted for the kernels that fill the machine — two G1 accumulation launches,
    one G2, four column and four row transform launches per Groth16 proof (two thirds of the latter over a bound key); sort, fold and
    mat-vec add < 3 % —, divided by what 1024 SIMDs issue at one instruction per 4 cycles at `clock_ghz`: the shader clock measured
    LIVE beside the kernels (zkhip_ctx_clock_probe), or — without a probe — the clock the counter pass derived.  OFFLINE instruction
    counts, this run's time and clock.  None if the file does not hold what is needed."""
    try:
        # (the counter passes run with the key as loaded, `--bind 0`: 12 pass-vectors in four column and four row launches; a proof
        # over a bound key runs `pass_vectors` = 8 of the same)
        per_proof = {"G1": 2, "G2": 1, "NTT_cols": 4 * pass_vectors / 12.0, "NTT_rows": 4 * pass_vectors / 12.0}
        instr = sum(n * pv[k]["valu_wave_instructions_per_launch"] for k, n in per_proof.items())
        clock = clock_ghz or pv["G1"]["clock_ghz"]
        ms = instr / (1024 * clock * 1e9 / 4) * 1e3
        return {"valu_wave_instructions_per_proof": instr, "clock_ghz_under_load": clock, "clock_is_live": bool(clock_ghz),
                "ms_per_proof_at_issue_limit": ms,
                "frac_of_ms_per_step": ms / ms_per_step if ms_per_step > 0 else None,
                "source": "OFFLINE instruction counts (profiles/pmc_valu.json), this run's ms_per_step and shader clock"}
    except Exception:
        return None


# instructions of the accumulation's hot loop per sorted entry (tools/isa_mix.py on the build of profiles/r5_accum_isa_mix.txt): the
# fallback of compute_bound when no counter file of THIS build is at hand
STATIC_VALU_PER_ADDITION = {"G1": 2427, "G2": 6506}


class ClockProbe:
    """The shader clock under the bench's own load: a SECOND context on the rank's device runs zkhip_ctx_clock_probe — one wavefront
    that compares the shader-cycle counter with the constant-rate wall clock over 2 ms, asleep in between — in a host thread, back to
    back, while the first context proves (the calls release the GIL).  window(t0, t1) averages the readings that ended inside."""

    def __init__(self, device, period_us=2000):
        import threading
        self.samples = []
        self.period_us = period_us
        self._stop = threading.Event()
        self._thread = None
        self.ctx = None
        self.error = None
        if os.environ.get("ZKHIP_BENCH_NO_PROBE"):      # (experiment: what does the probe's context cost the proofs it runs beside?)
            self.error = "ZKHIP_BENCH_NO_PROBE"
            return
        try:
            self.ctx = native.Context(device)
            self.ctx.clock_probe(200)
        except Exception as e:      # (an older library, the emulator: no probe, no figure)
            self.error = repr(e)[:160]
            self.ctx = None

    def start(self):
        import threading
        if self.ctx is None:
            return

        def loop():
            while not self._stop.is_set():
                try:
                    ghz = self.ctx.clock_probe(self.period_us)
                except Exception as e:
                    self.error = repr(e)[:160]
                    return
                self.samples.append((time.perf_counter(), ghz))
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def window(self, t0, t1):
        vals = [g for t, g in self.samples if t0 <= t <= t1 and g > 0.3]      # (< 0.3 GHz: a counter that does not tick at the shader clock)
        if not vals:
            return {"mean_ghz": None, "samples": 0, "note": self.error or "no reading inside the window"}
        return {"mean_ghz": sum(vals) / len(vals), "min_ghz": min(vals), "max_ghz": max(vals), "samples": len(vals)}

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)
        if self.ctx is not None:
            try:
                self.ctx.close()
            except Exception:
                pass
            self.ctx = None


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


def rocm_smi():
    """Clocks, performance level, partition modes and power cap of GPU 0 as rocm-smi reports them (attribution of
    box-to-box differences); None if the tool is absent."""
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showperflevel", "--showcomputepartition", "--showmemorypartition",
                              "--showmaxpower", "--json"], capture_output=True, text=True, timeout=30).stdout
        doc = json.loads(txt[txt.index("{"):])
        card = next(iter(doc.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "performance level", "partition", "power")):
                keep[k] = v
        return keep or None
    except Exception:
        return None


def proof_algorithmic_bytes(circ, fq):
    """Compulsory HBM traffic of one proof: every input read once, every output written once per logical stage
    (SURVEY.md §8d): mat-vec, the transforms (6 here, 7 in the reference), MSM bases, MSM scalars."""
    F, n, m, N, w, l = 32, circ.n, circ.m, circ.N, circ.w, circ.l
    nnz = sum(int(mat[0][-1]) for mat in circ.mats())
    matvec = nnz * (F + 4) + 3 * (n + 1) * 8 + m * F + 3 * N * F
    transforms = 6 * 2 * N * F          # (the reference's witness_map runs 7; this one needs 6)
    bases = (N - 1) * 2 * fq + w * 2 * fq + 2 * m * 2 * fq + m * 4 * fq
    scalars = N * F + w * F + 3 * m * F
    return matvec + transforms + bases + scalars


def gm17_algorithmic_bytes(circ, fq, M, D):
    """The same accounting for GM17: SAP rows (matrices + z read, two D-vectors and the n + l - 1 extension written),
    5 transforms over D, the five base sets (4 over the M SAP variables, one over D) and their scalars."""
    F, n, m = 32, circ.n, circ.m
    nnz = sum(int(mat[0][-1]) for mat in circ.mats())
    rows = nnz * (F + 4) + 3 * (n + 1) * 8 + m * F + 2 * D * F + (M - m) * F
    transforms = 4 * 2 * D * F          # (ark-gm17 runs 5)
    bases = 3 * M * 2 * fq + M * 4 * fq + D * 2 * fq
    scalars = 4 * M * F + D * F
    return rows + transforms + bases + scalars


if __name__ == "__main__":
    main()
