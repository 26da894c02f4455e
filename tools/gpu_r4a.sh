#!/bin/bash
# round 4, first GPU session: the large-domain tests first (new code), then the whole -m gpu suite, then the bench line.
set -u
tag=${1:-r4a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; tail -5 "$out/smoke.log"; exit 0; }
tail -1 "$out/smoke.log"
timeout 900 python -m pytest tests/test_gpu_large_domains.py -m gpu -x -q --durations=8 > "$out/pytest_large.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_large.log"
tail -14 "$out/pytest_large.log"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_large_domains.py > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -14 "$out/pytest_gpu.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$out/bench_driver_command.json").read().strip().splitlines()[-1])
    print("proofs/s", d["value"], "ms/step", d["ms_per_step"], "single", d["single_proof_ms"], "serial", d.get("phases_ms_serial"))
    e = d.get("cli_end_to_end_ms", {})
    for k, v in e.items():
        if isinstance(v, dict) and "process_wall_ms" in v:
            print(k, {q: v.get(q) for q in ("process_wall_ms", "hip_init_ms", "key_load_ms", "parse_program_ms", "prove_ms")})
except Exception as ex:
    print("no bench line:", ex)
PY
