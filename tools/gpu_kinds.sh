#!/bin/bash
# one bench line per workload kind on one box (no CPU leg, no CLI legs): bash tools/gpu_kinds.sh <tag>
set -u
tag=${1:-kinds}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; exit 0; }
run() { name=$1; shift; timeout 400 python bench.py --cpu-seconds 0 --e2e 0 --steps 32 "$@" > "$out/bench_$name.json" 2>> "$out/bench.err"; }
run dense
run sha256 --kind sha256
run sha256_2e18 --kind sha256 --log-domain 18
run sha_standin --kind sha
run poseidon_bls --kind poseidon --curve bls12_381 --log-domain 18
run gm17 --scheme gm17
for f in "$out"/bench_*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); s=d.get('phases_ms_serial') or {}
        print('%-24s' % sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s', round(d['single_proof_ms'],2),'ms single | serial', {k: round(v,2) for k,v in s.items() if k in ('ntt_ms','msm_h_ms','msm_z_ms','total_ms','kernel_ntt_ms','kernel_msm_accum_g1_ms','kernel_msm_accum_g2_ms')})
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
