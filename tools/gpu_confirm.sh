#!/bin/bash
# the driver's round-end order at the head commit: pytest -m gpu, smoke(), the bench command — and the sha256 program through the
# CLI-shaped legs: bash tools/gpu_confirm.sh <tag>
set -u
tag=${1:-confirm}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 1200 python -m pytest tests/ -x -q -m gpu > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$out/pytest_gpu.log"
tail -4 "$out/pytest_gpu.log"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -1 "$out/smoke.log"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"; echo "bench rc=$?"
timeout 400 python bench.py --kind sha256 --cpu-seconds 0 --steps 16 > "$out/bench_sha256_with_cli_legs.json" 2> "$out/bench_sha256.err"; echo "sha256 rc=$?"
python - "$out" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_driver_command.json").readline())
print("driver command:", round(d["value"],2), "proofs/s", round(d["single_proof_ms"],2), "ms single; cpu", round(d["cpu_baseline"]["value"],3), d["cpu_baseline"].get("gpu_proof_identical"), "| evidence stale:", d["roofline"]["offline_evidence"]["stale"], "| native CLI wall ms", {k: round(v["process_wall_ms"]) for k,v in d["cli_end_to_end_ms"].items() if isinstance(v,dict) and k.startswith("native")})
d=json.loads(open(sys.argv[1]+"/bench_sha256_with_cli_legs.json").readline())
print("sha256:", round(d["value"],1), "proofs/s", round(d["single_proof_ms"],2), "ms single")
for k,v in d["cli_end_to_end_ms"].items():
    if isinstance(v,dict): print(" ", k, {kk:(round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("process_wall_ms","parse_program_ms","key_load_ms","hip_init_ms","prove_ms","proof_json_identical_to_resident_prover","verified","error")})
    else: print(" ", k, v)
PY
