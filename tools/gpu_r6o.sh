#!/bin/bash
# round 6, session o: the transform passes — first round on the way in (ntt_fuse_first), loose quotient digits (the default build against
# libzkhip_tightntt.so = -DZK_NTT_LOOSE=0), one column per workgroup of the cols pass — parity first, then pass times alone in one process each.
set -u
tag=${1:-r6o}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "transform parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_domains.py tests/test_gm17.py tests/test_gpu_bound.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_ntt.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_ntt.log"; tail -3 "$out/pytest_ntt.log"
step "first round on the way in: 0 / 1 (default build: loose digits)"
python tools/ntt_probe.py ntt_fuse_first 0 1 > "$out/fuse_first.txt" 2>&1; cat "$out/fuse_first.txt"
step "the same, tight digits (libzkhip_tightntt.so)"
ZKHIP_LIBRARY=$root/zokrates_amd/libzkhip_tightntt.so python tools/ntt_probe.py ntt_fuse_first 0 1 > "$out/fuse_first_tight.txt" 2>&1; cat "$out/fuse_first_tight.txt"
step "columns per workgroup of the cols pass: 2 / 1 / 4"
python tools/ntt_probe.py ntt_cols 2 1 4 > "$out/ntt_cols.txt" 2>&1; cat "$out/ntt_cols.txt"
step "done"
