#!/bin/bash
# round 6, session d: the witness map split between members on the device, the paired transform tiles' traffic, counters of the bound pipeline.
set -u
tag=${1:-r6d}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
T0=$(date +%s)
step() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
step "GPU tests: binding, split, multi-device, stream jitter"
timeout 900 python -m pytest tests/test_gpu_bound.py tests/test_multi_device.py tests/test_stream_jitter.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > "$out/pytest_split.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_split.log"; tail -4 "$out/pytest_split.log"
step "2^20 as 8 members of one proof (one GPU): unbound / bound+split / bound without split"
python - > "$out/members8_2e20.txt" 2>&1 <<'PY'
import json, sys, time
sys.path.insert(0, '.')
from zokrates_amd import native, synth
native.default_library().init(8)
ctx = native.Context(0)
for lg in (20, 22):
    circ = synth.circuit(0, lg)
    cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
    raw = native.setup_g16(ctx, cs, synth.toxic_waste(0))
    z = circ.assignment(7)
    pk = native.ProvingKey(ctx, 0, raw)
    whole = native.prove_g16(ctx, pk, cs, z, 11, 13)
    pk.close(); cs.close()
    multi = native.Multi([0] * 8)
    multi.load_constraint_system(0, circ.n, circ.l, circ.w, circ.mats())
    multi.load_proving_key(0, raw)
    def timed(label):
        best, ph = 1e9, None
        for i in range(6):
            t0 = time.perf_counter()
            p, tm = multi.prove_g16(z, 11, 13, want_timings=True)
            dt = 1000 * (time.perf_counter() - t0)
            assert p == whole
            if i and dt < best: best, ph = dt, tm
        print(json.dumps({"log_domain": lg, "mode": label, "ms": round(best, 3), "split": multi.last_split(), "slowest_member": {k: round(v, 3) for k, v in ph.items() if k in ("ntt_ms", "kernel_ntt_ms", "msm_z_ms", "msm_h_ms", "kernel_msm_accum_g1_ms", "kernel_msm_accum_g2_ms")}}), flush=True)
    timed("shards as loaded")
    t0 = time.time(); multi.bind(raw); print("multi bind_ms", round(1000 * (time.time() - t0)))
    timed("bound, witness map split")
    multi.transform_split(False)
    timed("bound, whole map on every member")
    multi.close()
PY
cat "$out/members8_2e20.txt"
step "counter passes of the bound pipeline"
bash tools/gpu_pmc_r6.sh "$tag" 2>&1 | tail -8
step "done"
