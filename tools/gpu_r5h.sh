#!/bin/bash
# round 5, session H: the one-proof process after its streams became lazy (CLI legs, start profile), sort / NTT occupancy variants.
set -u
tag=${1:-r5h}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$out/smoke.log"; }
tail -1 "$out/smoke.log"
timeout 600 python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > "$out/bench_with_cli_legs.json" 2>> "$out/bench.err"
ZKHIP_BENCH_CLI_GAP_S=1.5 timeout 600 python bench.py --cpu-seconds 0 --steps 8 --warmup 4 --serial-proofs 0 --repeats 1 > "$out/bench_cli_spaced.json" 2>> "$out/bench.err"
for f in with_cli_legs cli_spaced; do python - "$out/bench_$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); e=d.get('cli_end_to_end_ms') or {}
print(sys.argv[1].split('/')[-1], round(d['value'],2), 'proofs/s')
for k,v in e.items():
    if isinstance(v,dict) and 'process_wall_ms' in v: print('   %-38s wall %4d  in-process %4d  hip_init %4d  key %4d  prove %5.1f  identical %s' % (k, v['process_wall_ms'], v.get('total_in_process_ms',0), v.get('hip_init_ms',0), v.get('key_load_ms',0), v.get('prove_ms',0), v.get('proof_json_identical_to_resident_prover')))
PY
done
B="timeout 300 python bench.py --cpu-seconds 0 --e2e 0 --steps 64 --warmup 8 --serial-proofs 2"
cfgs=(
 "default|libzkhip.so|"
 "sort1024|libzkhip_sort1024.so|"
 "nttwg3|libzkhip_nttwg3.so|"
 "nttwg2|libzkhip_nttwg2.so|"
 "nttcols1|libzkhip.so|ZKHIP_NTT_COLS=1"
 "nttcols4|libzkhip.so|ZKHIP_NTT_COLS=4"
)
for rep in 1 2; do
  for c in "${cfgs[@]}"; do
    IFS='|' read -r name lib envs <<< "$c"
    env $envs ZKHIP_LIBRARY=$root/zokrates_amd/$lib $B >> "$out/bench_$name.json" 2>> "$out/bench.err"
  done
done
python tools/ab_summary.py "$out" | grep -v "with_cli\|cli_spaced"
tail -2 "$out/bench.err"
