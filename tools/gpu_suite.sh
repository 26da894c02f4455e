#!/bin/bash
# the -m gpu suite alone (plus smoke), for re-verification after a change: bash tools/gpu_suite.sh <tag>
set -u
tag=${1:-suite}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED: giving the box back"; tail -5 "$out/smoke.log"; exit 0; }
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
tail -1 "$out/smoke.log"; tail -14 "$out/pytest_gpu.log"
