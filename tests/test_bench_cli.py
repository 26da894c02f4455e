"""bench.py itself on CPU: the real script, pointed at the TEST-ONLY emulator build through the ZKHIP_LIBRARY hook, at toy
sizes — the JSON contract (one line, rank 0 only, the required keys, roofline + cpu_baseline objects), the three
workloads / two schemes, and the N = 2 path over gloo including the sharded single-proof leg.  Numbers are meaningless
here; shapes and parity flags are not."""
import json
import os
import subprocess
import sys

import pytest

from emu_util import EMU_LIB, emu_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"]


def run_bench(args, extra_env=None, timeout=900, expect_rc=0):
    emu_library()                                            # make sure the emulator build exists
    env = dict(os.environ, ZKHIP_LIBRARY=EMU_LIB)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == expect_rc, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return lines


@pytest.mark.parametrize("args,metric", [
    (["--log-domain", "5"], "groth16_proofs_per_sec"),
    (["--log-domain", "5", "--scheme", "gm17"], "gm17_proofs_per_sec"),
    (["--log-domain", "8", "--kind", "poseidon", "--curve", "bls12_381"], "groth16_proofs_per_sec"),
])
def test_single_rank_contract(args, metric):
    with_cli_leg = "--scheme" not in args and "--kind" not in args      # the CLI-shaped leg (seven subprocesses) once is enough
    lines = run_bench(args + ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2"] + ([] if with_cli_leg else ["--e2e", "0"]),
                      extra_env={"ZKHIP_BENCH_CLI_ALL": "1"})
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == metric and d["unit"] == "proofs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["gpu_proof_identical"] is True
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1e-6 * 1000
    rp = d["repeats"]                                   # the timed region three times over; `value` is the first
    assert rp["regions"] == 3 and len(rp["ms_per_step"]) == 3 and abs(rp["ms_per_step"][0] - d["ms_per_step"]) < 1e-9 and rp["spread"] >= 0
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["rank"] == 0 and abs(d["per_rank"][0]["ms_per_step"] - d["ms_per_step"]) < 1e-6
    assert "under_load" in d                            # (no sysfs view of a GPU here: the sampler says so instead of inventing numbers)
    b = d["bound_key"]                                  # either scheme runs with the key bound to its system, checked against an unbound proof and the CPU's
    assert b["bound"] is True and b["proof_identical_to_unbound"] is True and b["bind_ms"] > 0 and b["unbound_ms_per_step"] > 0, b
    assert c["gpu_bound_key_proof_identical"] is True
    if metric == "groth16_proofs_per_sec":
        assert "4 NTTs" in d["config"]["workload"]
    # which figure `value` is, both figures side by side, the CPU ratio on the like-for-like (unbound) schedule
    assert d["value_is"] == "value_bound" and abs(d["value_bound"] - d["value"]) < 1e-9 and d["value_unbound"] > 0 and "bound" in d["key_state"]
    assert abs(d["speedup_vs_cpu_baseline"] - d["value_unbound"] / c["value"]) < 1e-6 * d["speedup_vs_cpu_baseline"]
    assert d["identical_to_oracle"] is True and "algorithmic" in d["oracle"]
    assert "configs" not in d                           # (the other BASELINE configurations ride on the default workload only)
    if not with_cli_leg:
        return
    e = d["cli_end_to_end_ms"]                         # the reference-shaped flow: files -> proof.json, one process per proof
    assert "error" not in e, e
    for run in ("native_from_proving_key", "native_from_key_image", "from_proving_key", "from_key_image"):
        assert e[run]["proof_json_identical_to_resident_prover"] is True and e[run]["process_wall_ms"] > 0, (run, e[run])
    assert e["from_proving_key"]["key_source"] == "proving.key" and e["from_key_image"]["key_source"] == "image"
    chk = e["native_from_key_image_with_verify"]         # the proof of the run checked by the compiled verifier before the process reports success
    assert chk["verified"] is True and chk["verify_ms"] > 0 and chk["proof_json_identical_to_resident_prover"] is True, chk


@pytest.mark.parametrize("scheme,port", [("g16", "29541")])      # (the sharded GM17 proof over two processes: tests/test_multiprocess.py)
def test_two_ranks_real_bench_script(scheme, port):
    world = 2
    procs = []
    for rank in range(world):
        env = dict(os.environ, ZKHIP_LIBRARY=EMU_LIB, ZKHIP_DIST_BACKEND="gloo", ZKHIP_BENCH_DEVICE="0", RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-domain", "5", "--steps", "2", "--warmup", "1", "--scheme", scheme],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # ONE line, from rank 0
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["identical_to_oracle"] is True and "closed-form" in d["oracle"], d.get("oracle")     # N > 1: rank 0 holds its proof to the closed form
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1000.0)) < 1e-6 * d["value"]          # whole-job aggregate over both ranks
    s = d["sharded_single_proof"]
    assert s.get("identical_to_unsharded") is True and s["ranks"] == 2, s
    mm = d["multi_single_proof"]                      # the in-library path: rank 0 drives one member per rank's device
    assert mm.get("identical_to_unsharded") is True and mm["members"] == 2, mm
    if scheme == "g16":                               # and its throughput mode (whole key per member, proofs dealt over the members)
        assert mm["replicas_batch"]["first_identical_to_unsharded"] is True and mm["replicas_batch"]["proofs"] == 16, mm


def test_gpus_flag_alone_starts_the_ranks():
    """`python bench.py --gpus 2` with NO rank environment (the shape of the driver's N = 1 command): bench.py
    starts the two ranks itself and the line says n_gpus = 2 — here over gloo on the emulator, both ranks on its one device."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZKHIP_LIBRARY=EMU_LIB, ZKHIP_DIST_BACKEND="gloo", ZKHIP_BENCH_DEVICE="0")
    emu_library()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-domain", "5", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "2 rank(s)" in d["config"]["process_group"] and "itself" in d["config"]["launcher"]
    pr = d["per_rank"]                                  # one record per rank that ran, each with its own rate
    assert len(pr) == 2 and sorted(r["rank"] for r in pr) == [0, 1] and all(r["value"] > 0 and r["device"] == 0 for r in pr)
    assert max(r["ms_per_step"] for r in pr) <= d["ms_per_step"] * (1 + 1e-6)      # the line's time is the slowest rank's
    assert d["sharded_single_proof"]["identical_to_unsharded"] is True and d["multi_single_proof"]["identical_to_unsharded"] is True
    for k in REQUIRED:
        assert k in d, k


def test_more_gpus_asked_for_than_visible_fails_loudly():
    """--gpus 3 on a box with one device: an error line and exit status 1, not a one-GPU measurement labelled 3 (or 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ZKHIP_BENCH_DEVICE")}
    env.update(ZKHIP_LIBRARY=EMU_LIB)
    emu_library()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--log-domain", "5"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 1
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] is None and d["n_gpus"] == 0 and d["requested_gpus"] == 3 and d["visible_gpus"] == 1 and "refusing" in d["error"]
    # and under a launcher: --gpus must agree with the ranks that exist
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr and not p.stdout.strip()


def test_stalled_optional_leg_still_prints_the_line():
    """An optional latency leg that never returns (a hung collective, an untested multi-GPU path): the watchdog prints the
    throughput line with the leg marked as timed out and the process exits 0 — the driver's bench run must not lose its line."""
    lines = run_bench(["--log-domain", "5", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--members", "2"],
                      extra_env={"ZKHIP_BENCH_TEST_STALL": "1", "ZKHIP_BENCH_LEG_TIMEOUT_S": "3"}, timeout=300)
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and "timed out" in d["multi_single_proof"]["error"] and d["cpu_baseline"] is None
    for k in REQUIRED:
        assert k in d, k


def test_measuring_process_that_dies_is_reported_and_retried():
    """A GPU memory fault aborts the process that owns the queue (the driver's round-2 bench run: rc 134, nothing on stdout).
    bench.py measures in a child: a child that dies once is measured again and the line carries the failed attempt; a child
    that dies twice still leaves ONE JSON line (value null, "error", both attempts) and a non-zero exit status."""
    args = ["--log-domain", "5", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]
    lines = run_bench(args, extra_env={"ZKHIP_BENCH_TEST_DIE": "0"})
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and len(d["attempts"]) == 2
    assert d["attempts"][0]["signal"] == 6 and d["attempts"][0]["last_stage"] == "context_created" and d["attempts"][1]["exit_status"] == 0
    assert d["bound_key"]["bound"] is False and "second attempt" in d["bound_key"]["note"]      # the retry leaves the key as loaded
    assert "6 NTTs" in d["config"]["workload"]
    assert d["value_is"] == "value_unbound" and d["value_bound"] is None and "FALLBACK" in d["key_state"]      # ... and the line says so at top level
    lines = run_bench(args, extra_env={"ZKHIP_BENCH_TEST_DIE": "*"}, expect_rc=1)
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and "error" in d and [a["signal"] for a in d["attempts"]] == [6, 6]


def _bench_module():
    import importlib.util
    env_before = os.environ.get("ZKHIP_BENCH_CHILD")
    os.environ["ZKHIP_BENCH_CHILD"] = "1"           # importing bench.py must not start its supervising parent
    try:
        spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
    finally:
        if env_before is None:
            del os.environ["ZKHIP_BENCH_CHILD"]
        else:
            os.environ["ZKHIP_BENCH_CHILD"] = env_before
    return bench


def test_offline_evidence_is_tied_to_the_build(tmp_path):
    """The counter figures bench.py prints (roofline.traffic, VALU issue utilisation, the pipeline's issue bound) come from files
    committed under profiles/; each carries the fingerprint of the sources its rocprofv3 pass ran on.  Edit a kernel file and the
    figures are gone from the line and `stale` says so."""
    import shutil
    from zokrates_amd.build import csrc_hash
    bench = _bench_module()
    root = tmp_path / "tree"
    (root / "zokrates_amd" / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    (root / "profiles").mkdir()
    src = os.path.join(ROOT, "zokrates_amd", "csrc")
    for n in os.listdir(src):
        if os.path.isfile(os.path.join(src, n)):
            shutil.copy(os.path.join(src, n), root / "zokrates_amd" / "csrc" / n)
    shutil.copy(os.path.join(ROOT, "include", "zkhip.h"), root / "include" / "zkhip.h")
    assert csrc_hash(str(root)) == csrc_hash()                    # the same bytes, the same fingerprint, wherever the tree lies
    for n in ("pmc_traffic.json", "pmc_valu.json"):
        doc = json.load(open(os.path.join(ROOT, "profiles", n)))
        doc["csrc_hash"] = csrc_hash(str(root))
        json.dump(doc, open(root / "profiles" / n, "w"))
    ev = bench.offline_evidence(root=str(root))
    assert ev["stale"] is False and ev["traffic"]["G1"]["traffic_bytes_per_launch"] > 0 and ev["valu"]["G1"]["issue_utilisation"] > 0
    # the file readers hold no kernel and launch none: an edit there moves no counter and leaves the fingerprint alone
    with open(root / "zokrates_amd" / "csrc" / "ingest.hip", "a") as f:
        f.write("// a reader edit\n")
    assert bench.offline_evidence(root=str(root))["stale"] is False
    assert csrc_hash(str(root), with_host_only=True) != csrc_hash(with_host_only=True)
    with open(root / "zokrates_amd" / "csrc" / "kernels_msm.cuh", "a") as f:
        f.write("// a kernel edit\n")
    ev = bench.offline_evidence(root=str(root))
    assert ev["stale"] is True and ev["traffic"] is None and ev["valu"] is None and "another build" in ev["why"]
    # one refreshed file, one forgotten: the fresh one is used, the line still says stale
    doc = json.load(open(root / "profiles" / "pmc_valu.json"))
    doc["csrc_hash"] = csrc_hash(str(root))
    json.dump(doc, open(root / "profiles" / "pmc_valu.json", "w"))
    ev = bench.offline_evidence(root=str(root))
    assert ev["stale"] is True and ev["traffic"] is None and ev["valu"] is not None


def test_pipeline_issue_bound_from_the_committed_counter_file():
    """bench.pipeline_issue_bound: the committed VALU counter pass turned into the issue-limited time of one proof."""
    bench = _bench_module()
    pv = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
    b = bench.pipeline_issue_bound(pv, 10.57)
    assert 4.0e9 < b["valu_wave_instructions_per_proof"] < 6.0e9 and 8.0 < b["ms_per_proof_at_issue_limit"] < 11.0
    assert abs(b["frac_of_ms_per_step"] - b["ms_per_proof_at_issue_limit"] / 10.57) < 1e-12
    assert bench.pipeline_issue_bound({}, 10.0) is None and bench.pipeline_issue_bound({"G1": {}}, 10.0) is None


def test_bench_window_count_mirrors_the_library_rule():
    """bench.py prices the accumulation in mixed additions = scalars x windows; the window count is the library's rule for resident
    keys (csrc/core.cuh msm_shape): 17-bit windows for 254-bit scalars (15), 16 for 255-bit ones, log2 n + 1 bits for small keys."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_for_test", os.path.join(ROOT, "bench.py"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    ns = {}
    start = src.index("def msm_windows(")
    exec(src[start:src.index("def numa_placement(")], ns)
    assert ns["msm_windows"](254, (1 << 20) + 2) == 15 and ns["msm_windows"](255, (1 << 20) + 2) == 16
    assert ns["msm_windows"](254, 1 << 10) == (254 + 1 + 10) // 11 and ns["msm_windows"](254, 1 << 24) == 15


def test_config_legs_of_the_default_line():
    """The default line carries BASELINE.json's other configurations as its LAST key: short runs of the same script, each bound and
    held to the oracle's closed form.  Here at toy size through the hook that overrides the legs' arguments."""
    lines = run_bench(["--log-domain", "5", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--e2e", "0", "--configs", "1"],
                      extra_env={"ZKHIP_BENCH_TEST_LEGS": "1"}, timeout=1500)
    d = json.loads(lines[0])
    assert list(d.keys())[-2:] == ["configs", "attempts"] or list(d.keys())[-1] == "configs", list(d.keys())[-3:]
    cfg = d["configs"]
    assert set(cfg) == {"gm17_2e20", "poseidon_chain_bls12_381_2e18", "sha256_stdlib_2e20", "dense_2e22_and_8_members"}
    for k, rec in cfg.items():
        assert "error" not in rec and "skipped" not in rec, (k, rec)
        assert rec["identical_to_oracle"] is True and rec["proofs_per_s"] > 0 and rec["single_proof_ms"] > 0 and rec["key_bound"] is True, (k, rec)
    mm = cfg["dense_2e22_and_8_members"]["members"]
    assert mm["members"] == 2 and mm["identical_to_unsharded"] is True and mm["identical_to_oracle"] is True and mm["key_bound"] is True, mm
