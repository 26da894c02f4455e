#!/bin/bash
# round 4: CLI-shaped numbers after the key-load work + the large-domain bench lines (VERDICT r3 item 2 "bench lines for both")
set -u
tag=${1:-r4e}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; exit 0; }
tail -1 "$out/smoke.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"; echo "bench rc=$?"
python - "$out/bench_driver_command.json" <<'PY'
import json,sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("proofs/s", d["value"], "ms/step", d["ms_per_step"], "single", d["single_proof_ms"], "host_ms", d["host_ms"])
for k, v in d.get("cli_end_to_end_ms", {}).items():
    if isinstance(v, dict) and "process_wall_ms" in v:
        print(k, {q: v.get(q) for q in ("process_wall_ms", "hip_init_ms", "key_load_ms", "parse_program_ms", "prove_ms", "tables", "host_threads", "numa_node", "proof_json_identical_to_resident_prover")})
PY
# large domains: Groth16 over the literal n = 2^22 (domain 2^23), GM17 over n = 2^22 - 2 (SAP domain 2^23)
timeout 900 python bench.py --constraints 4194304 --log-domain 23 --steps 8 --warmup 2 --witnesses 2 --cpu-seconds 0 --e2e 0 > "$out/bench_g16_n2e22_domain2e23.json" 2> "$out/bench_g16_2e23.err"; echo "g16 2^23 rc=$?"
timeout 900 python bench.py --scheme gm17 --log-domain 22 --steps 6 --warmup 2 --witnesses 2 --cpu-seconds 0 --e2e 0 > "$out/bench_gm17_n2e22_sap2e23.json" 2> "$out/bench_gm17_2e23.err"; echo "gm17 rc=$?"
for f in bench_g16_n2e22_domain2e23 bench_gm17_n2e22_sap2e23; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["metric"], round(d["value"], 2), "proofs/s", round(d["ms_per_step"], 2), "ms/step; single", round(d["single_proof_ms"], 2), d["config"]["workload"][:90])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -3 "$out"/bench_g16_2e23.err "$out"/bench_gm17_2e23.err
