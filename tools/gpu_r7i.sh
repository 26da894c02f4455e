#!/bin/bash
# round 6, session r7i: the staging ring filled by helper threads started once per transfer (devrt.h dev_h2d; ZKHIP_COPY_THREADS 1 / 4 / 2 / 8, alternating):
# parity of the uploads first (the full-size and large-domain tests move 32 MiB - 1 GiB through the ring), then the proof from host memory
set -u
tag=${1:-r7i}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "parity (uploads through the threaded ring)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_domains.py tests/test_ingest.py tests/test_gpu_bound.py -m gpu -q -x -p no:cacheprovider > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -3 "$out/pytest.log"
step "from host memory, dense 2^20"
for t in 1 4 1 4 2 8; do
  ZKHIP_COPY_THREADS=$t timeout 120 python3 bench.py --steps 8 --warmup 3 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 1 --oracle none --configs 0 --bind 2 2> /dev/null |
    python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  copy threads $t: single resident %.2f ms, from host %.2f ms, pk_load %.0f ms, r1cs+key timeline %s' % (d['single_proof_ms'], d['single_proof_from_host_ms'], d['host_ms']['pk_load'], {k: d['timeline_s'][k] for k in ('context_created','r1cs_resident','setup_done','key_resident','assignments_resident')}))"
done
step "done"
