#!/bin/bash
# round 4, late session: the NTT up to the two-adicity, the SHA-256 circuit (test + bench lines): bash tools/gpu_r4n.sh <tag>
set -u
tag=${1:-r4n}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; exit 0; }
free -g | head -2
timeout 900 python -m pytest tests/test_sha256_circuit.py tests/test_gpu_large_domains.py -x -q -m gpu --durations=10 -k "sha256 or two_adicity" > "$out/pytest_new.log" 2>&1
echo "pytest rc=$?"; tail -15 "$out/pytest_new.log"
for ld in 20 18; do
  timeout 600 python bench.py --kind sha256 --log-domain $ld --steps 20 --warmup 3 --e2e 0 > "$out/bench_sha256_2e$ld.json" 2> "$out/bench_sha256_2e$ld.err"
  echo "bench sha256 2^$ld rc=$?"; tail -2 "$out/bench_sha256_2e$ld.err"
done
timeout 300 python bench.py --kind sha --steps 20 --warmup 3 --e2e 0 --cpu-seconds 0 > "$out/bench_sha_standin.json" 2>> "$out/bench.err"
timeout 300 python bench.py --steps 20 --warmup 3 --e2e 0 --cpu-seconds 0 > "$out/bench_dense.json" 2>> "$out/bench.err"
for f in "$out"/bench_*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single; identical', d.get('cpu_baseline',{}).get('gpu_proof_identical'), 'cpu', d.get('cpu_baseline',{}).get('value'), '| serial', {k: round(v,3) for k,v in s.items()})
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
