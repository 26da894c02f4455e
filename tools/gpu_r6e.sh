#!/bin/bash
# round 6, session e: the transform passes with batched loads / prefetched factors / entry-major plans: parity and time.
set -u
tag=${1:-r6e}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "transform parity"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_domains.py tests/test_gm17.py -m gpu -q -x -p no:cacheprovider -k "ntt or witness or large or full_size or prove_matches or gm17" > "$out/pytest_ntt.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_ntt.log"; tail -3 "$out/pytest_ntt.log"
step "the driver's flags, no side legs"
for i in 1 2; do
timeout 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --e2e 0 --configs 0 > "$out/bench_$i.json" 2> "$out/bench_$i.err"
python - "$out/bench_$i.json" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print('  value', round(d['value'],2), 'unbound', d['value_unbound'] and round(d['value_unbound'],2), '| single', round(d['single_proof_ms'],2), '| ntt serial us/pass', round(d['roofline_ntt']['us_per_pass_serial'],2), 'frac_serial', round(d['roofline_ntt']['frac_serial'],4), '| pipelined us/pass', round(d['roofline_ntt']['us_per_pass'],1))
        print('  serial', {k: round(v,3) for k,v in d['phases_ms_serial'].items() if 'kernel' in k or k=='total_ms'})
PY
done
step "done"
