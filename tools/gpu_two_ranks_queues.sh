cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_two
for q in 16 8 12; do
  GPU_MAX_HW_QUEUES=$q ZKHIP_DIST_BACKEND=gloo ZKHIP_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 16 --warmup 4 --e2e 0 --serial-proofs 0 > gpurun_out/r5_two/q$q.json 2>> gpurun_out/r5_two/err.txt
  python - gpurun_out/r5_two/q$q.json $q <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print('queues', sys.argv[2], 'two ranks on one GPU:', round(d['value'],2), 'proofs/s aggregate;', [ (r['rank'], round(r['value'],1)) for r in d['per_rank']], d['repeats']['ms_per_step'])
PY
done
ZKHIP_BENCH_NO_PIN=1 GPU_MAX_HW_QUEUES=16 ZKHIP_DIST_BACKEND=gloo ZKHIP_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 16 --warmup 4 --e2e 0 --serial-proofs 0 > gpurun_out/r5_two/q16_nopin.json 2>> gpurun_out/r5_two/err.txt
python - gpurun_out/r5_two/q16_nopin.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print('queues 16, no NUMA pinning:', round(d['value'],2), d['repeats']['ms_per_step'])
PY
