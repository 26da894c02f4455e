#!/bin/bash
# round 6, session n: the whole -m gpu suite in one process at GPU_MAX_HW_QUEUES=16 (round 5's build aborted there: per-queue scratch for
# 0.8-2.9 KB frames, profiles/r5_q16_suite_abort.txt), then at the default 8.
set -u
tag=${1:-r6n}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; exit 0; }
tail -1 "$out/smoke.log"
GPU_MAX_HW_QUEUES=16 timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 -p no:cacheprovider > "$out/pytest_gpu_q16.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu_q16.log"
tail -14 "$out/pytest_gpu_q16.log"
