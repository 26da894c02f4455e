#!/bin/bash
# round 6, session 7a: where the thin workloads' time goes — kernel traces of the pipelined bench legs (stdlib SHA-256 2^20, Poseidon chain on
# BLS12-381 2^18, GM17 2^20): kernel tables and how much of the steady-state window has an accumulation in flight.
set -u
tag=${1:-r7a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
leg() { name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --stats -d "$out/prof_$name" -o $name -- python "$root/bench.py" --steps 32 --warmup 6 --witnesses 2 --cpu-seconds 0 --e2e 0 --serial-proofs 0 --repeats 1 --oracle none --configs 0 --bind 2 "$@" > "$out/prof_$name.log" 2>&1
  db=$(find "$out/prof_$name" -name "*.db" | head -1)
  [ -n "$db" ] && python "$root/tools/rocpd_stats.py" "$db" "$out/${tag}_${name}_pipelined_kernel_stats.md" > /dev/null
  [ -n "$db" ] && python "$root/tools/timeline.py" "$db" 0.6 > "$out/${tag}_${name}_pipelined_timeline.txt" 2>&1
  echo "== $name"; grep -h '^{"metric"' "$out/prof_$name.log" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(' value', round(d['value'],1), 'single', round(d['single_proof_ms'],2))"
  head -6 "$out/${tag}_${name}_pipelined_timeline.txt"; sed -n 3,24p "$out/${tag}_${name}_pipelined_kernel_stats.md" | cut -c1-110
}
leg sha --kind sha256 --log-domain 20
leg poseidon --curve bls12_381 --log-domain 18 --kind poseidon
leg gm17 --scheme gm17
find "$out" -name "*.db" -size +8M -delete
