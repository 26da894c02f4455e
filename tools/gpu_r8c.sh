#!/bin/bash
# round 6, session r8c: the staging ring's DMA issued per 2 MiB / 512 KiB piece of a slot instead of per 8 MiB slot (ZKHIP_COPY_UNIT_KB): a lone proof whose
# assignment comes from host memory (32 MiB), 24 per process, alternating; parity of the uploads first
set -u
tag=${1:-r8c}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_domains.py tests/test_ingest.py tests/test_gpu_bound.py tests/test_gpu_soak.py -m gpu -q -x -p no:cacheprovider > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -3 "$out/pytest.log"
ZKHIP_COPY_UNIT_KB=512 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_512.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_512.log"; tail -2 "$out/pytest_512.log"
for kb in 8192 2048 512 8192 2048 512 1024; do
  FROM_HOST=1 ZKHIP_COPY_UNIT_KB=$kb timeout 300 python3 tools/lone_stats.py dense 24 2> /dev/null | tee -a "$out/lone_stats.jsonl" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  from host, DMA per %4d KiB: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f' % ($kb, d['min'], d['p25'], d['median'], d['p75'], d['max']))"
done
timeout 300 python3 tools/lone_stats.py dense 24 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  resident                   : min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f' % (d['min'], d['p25'], d['median'], d['p75'], d['max']))"
