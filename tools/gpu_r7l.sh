#!/bin/bash
# round 6, session r7l: the suite's new full-size checks against the ALGORITHMIC oracle (Groth16 2^20, GM17 2^20: the C++ restatement of ark's
# create_proof over the key bytes the device made) and the soak (tests/test_gpu_soak.py): first at its default size, then 5000 proofs
set -u
tag=${1:-r7l}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "full-size tests with the algorithmic oracle"
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_full_size_properties" "tests/test_gm17.py::test_gpu_gm17_config5_full_size" -m gpu -q -x -p no:cacheprovider --durations=4 > "$out/full_size.log" 2>&1; echo "pytest rc=$?" >> "$out/full_size.log"; tail -9 "$out/full_size.log"
step "soak, default size"
timeout 600 python -m pytest tests/test_gpu_soak.py -m gpu -q -x -s -p no:cacheprovider --durations=2 > "$out/soak_default.log" 2>&1; echo "pytest rc=$?" >> "$out/soak_default.log"; tail -6 "$out/soak_default.log"
step "soak, ${SOAK:-5000} proofs"
ZKHIP_SOAK_PROOFS=${SOAK:-5000} timeout 900 python -m pytest tests/test_gpu_soak.py -m gpu -q -x -s -p no:cacheprovider --durations=2 > "$out/soak_long.log" 2>&1; echo "pytest rc=$?" >> "$out/soak_long.log"; tail -6 "$out/soak_long.log"
step "lone proofs: the witness map held for the sort of the assignment (lone_sched bit 4), under the plan"
ZKHIP_PIPES=1 ROUNDS=3 timeout 200 python tools/lone_ab.py 16 lone_sched 0 4 6 > "$out/lone_sched_bit4.txt" 2>&1; cut -c1-400 "$out/lone_sched_bit4.txt" | grep -o '"round".*' | cut -c1-200
step "done"
