#!/bin/bash
# round 6, session 7d: the transform passes' short butterfly rounds synchronised per wavefront (ntt_wave_sync 1) against workgroup barriers (0):
# parity against the oracle first (every size, both curves, three-pass domains), then the pass times alone, alternating in one process.
set -u
tag=${1:-r7d}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_domains.py tests/test_gm17.py tests/test_gpu_bound.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_ntt.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_ntt.log"; tail -3 "$out/pytest_ntt.log"
python tools/ntt_probe.py ntt_wave_sync 0 1 > "$out/wave_sync.txt" 2>&1; cat "$out/wave_sync.txt" | cut -c1-200
ROUNDS=2 timeout 120 python tools/lone_ab.py 16 ntt_wave_sync 0 1 | grep '^{' | cut -c1-220
