#!/bin/bash
# final verification of a round on the GPU box: smoke, the whole -m gpu suite, the PMC passes, the default bench line (with
# the CPU baseline), the other configurations of BASELINE.json.  Usage: [SKIP_SUITE=1] [SKIP_PMC=1] bash tools/gpu_final.sh <tag>
set -u
tag=${1:-final}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
git rev-parse HEAD > "$out/head.txt" 2>/dev/null || true
# the driver's own sequence first, on the cold box: the suite, smoke(), the bench command with the driver's flags
if [ -z "${SKIP_SUITE:-}" ]; then timeout 1500 python -m pytest tests/ -x -q -m gpu > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; fi
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench.err"
timeout 600 python bench.py --scheme gm17 --e2e 0 > "$out/bench_gm17.json" 2>> "$out/bench.err"
timeout 600 python bench.py --curve bls12_381 --log-domain 18 --kind poseidon --e2e 0 > "$out/bench_poseidon_bls12_381_2e18.json" 2>> "$out/bench.err"
timeout 600 python bench.py --kind sha --cpu-seconds 0 --e2e 0 > "$out/bench_sha_like.json" 2>> "$out/bench.err"
timeout 600 python bench.py --kind sha256 --e2e 0 > "$out/bench_sha256_stdlib_2e20.json" 2>> "$out/bench.err"
timeout 600 python bench.py --cpu-seconds 0 --constraints 1048576 --steps 16 --e2e 0 > "$out/bench_n2e20_literal_domain2e21.json" 2>> "$out/bench.err"
timeout 900 python bench.py --cpu-seconds 0 --log-domain 22 --steps 8 --members 8 --e2e 0 > "$out/bench_config3_2e22_members8.json" 2>> "$out/bench.err"
timeout 600 python bench.py --cpu-seconds 0 --members 8 --steps 16 --e2e 0 > "$out/bench_2e20_members8.json" 2>> "$out/bench.err"
# domains above 2^22 (three NTT passes): Groth16 over the literal n = 2^22 (domain 2^23), GM17 over n = 2^22 - 2 (SAP domain 2^23)
timeout 900 python bench.py --constraints 4194304 --log-domain 23 --steps 8 --warmup 2 --witnesses 2 --cpu-seconds 0 --e2e 0 > "$out/bench_g16_n2e22_domain2e23.json" 2>> "$out/bench.err"
timeout 900 python bench.py --scheme gm17 --log-domain 22 --steps 6 --warmup 2 --witnesses 2 --cpu-seconds 0 --e2e 0 > "$out/bench_gm17_n2e22_sap2e23.json" 2>> "$out/bench.err"
# and one size nobody asked for, to show there is no cap left: n = 2^24 - 2 (domain 2^24; a 6.4 GB key, 96 GiB of tables)
[ -n "${WITH_2E24:-}" ] && timeout 900 python bench.py --log-domain 24 --steps 4 --warmup 1 --witnesses 1 --serial-proofs 1 --cpu-seconds 0 --e2e 0 > "$out/bench_g16_domain2e24.json" 2>> "$out/bench.err"
# `bench.py --gpus 2` with no launcher: it starts the ranks itself (here both on this box's one GPU over gloo)
ZKHIP_DIST_BACKEND=gloo ZKHIP_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 16 --warmup 4 --e2e 0 > "$out/bench_gpus2_self_spawned_one_gpu.json" 2>> "$out/bench.err"
# the N > 1 code path of bench.py on real hardware: two ranks sharing this box's one GPU (gloo instead of RCCL, which wants
# one device per rank); rank 0 also drives the in-library multi leg
ZKHIP_DIST_BACKEND=gloo ZKHIP_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 \
  bench.py --gpus 2 --steps 16 --warmup 4 --e2e 0 > "$out/bench_two_ranks_one_gpu.json" 2>> "$out/bench.err"
# HBM traffic of the accumulation / transform kernels -> profiles/pmc_traffic.json (what the NEXT bench lines report as
# roofline.traffic): two PMC passes (their own runs, kernel trace only), one stream; a pass that does not finish in 200 s is
# given up.  Last, so that a profiler pass that hangs (one did: 900 s) cannot cost the bench lines.
[ -z "${SKIP_PMC:-}" ] && ( cd /tmp && export TMPDIR=/tmp ZKHIP_BENCH_CHILD=1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ZKHIP_SERIAL=1 timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d "$out/prof_pmc_$ctr" -o pmc -- python "$root/bench.py" --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_pmc_$ctr.log" 2>&1
    db=$(find "$out/prof_pmc_$ctr" -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && python "$root/tools/pmc_stats.py" "$db" "$out/${tag}_pmc_$ctr.md" > /dev/null
  done
  f=$(find "$out/prof_pmc_FETCH_SIZE" -name "*.db" 2>/dev/null | head -1); w=$(find "$out/prof_pmc_WRITE_SIZE" -name "*.db" 2>/dev/null | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then
    python "$root/tools/pmc_traffic.py" "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate runs), ZKHIP_SERIAL=1 python bench.py --bind 0 --steps 4 --warmup 1 --cpu-seconds 0 --serial-proofs 0; profiles/${tag}_pmc_FETCH_SIZE.md, ${tag}_pmc_WRITE_SIZE.md" > /dev/null \
      && cp "$out/pmc_traffic.json" "$root/profiles/pmc_traffic.json" && echo "pmc_traffic.json refreshed"
  else echo "PMC passes incomplete: profiles/pmc_traffic.json unchanged"; fi
  # VALU issue occupation of the same kernels -> pmc_valu.json (tools/pmc_valu.py)
  ZKHIP_SERIAL=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$out/prof_pmc_VALU" -o pmc -- \
    python "$root/bench.py" --bind 0 --cpu-seconds 0 --steps 4 --warmup 1 --serial-proofs 0 --e2e 0 > "$out/prof_pmc_VALU.log" 2>&1
  db=$(find "$out/prof_pmc_VALU" -name "*.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python "$root/tools/pmc_valu.py" "$db" "$out/${tag}_pmc_VALU.md" > /dev/null && cp "$out/${tag}_pmc_VALU.json" "$out/pmc_valu.json" && echo "pmc_valu.json written"
  find "$out" -name "*.db" -size +8M -delete )
cat "$out/smoke.log" | tail -1; [ -f "$out/pytest_gpu.log" ] && tail -12 "$out/pytest_gpu.log"
python - "$out/bench_two_ranks_one_gpu.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print('two ranks on one GPU:', round(d['value'],2), 'proofs/s aggregate, n_gpus', d['n_gpus'], '| sharded', d.get('sharded_single_proof'), '| multi', {k:v for k,v in (d.get('multi_single_proof') or {}).items() if k in ('ms','members','distinct_gpus','identical_to_unsharded','error')})
PY
for f in default gm17 poseidon_bls12_381_2e18 sha_like sha256_stdlib_2e20 n2e20_literal_domain2e21 config3_2e22_members8 2e20_members8 g16_n2e22_domain2e23 gm17_n2e22_sap2e23 g16_domain2e24 gpus2_self_spawned_one_gpu; do [ -f "$out/bench_$f.json" ] || continue; python - "$out/bench_$f.json" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    try:
        d=json.loads(line); c=d.get('cpu_baseline') or {}; m=d.get('multi_single_proof') or {}
        print(sys.argv[1].split('/')[-1][6:-5], round(d['value'],2), 'proofs/s |', round(d['single_proof_ms'],2),'ms single |', round(d['single_proof_from_host_ms'],2), 'from host | cpu', c.get('value'), c.get('gpu_proof_identical'), '| multi', m.get('ms'), m.get('identical_to_unsharded'), '| ntt frac_serial', (d.get('roofline_ntt') or {}).get('frac_serial'))
    except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 "$out/bench.err"
