#!/bin/bash
# round 6, session r7o: the N > 1 launch paths on the last build, two ranks sharing the box's one GPU (gloo for the barrier: RCCL refuses two ranks
# on one device) — bench.py starting its own ranks, and the driver's torch.distributed.run command
set -u
tag=${1:-r7o}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp ZKHIP_DIST_BACKEND=gloo ZKHIP_BENCH_DEVICE=0
show() { python3 - "$1" <<'PY'
import json,sys
ls=[l for l in open(sys.argv[1]) if l.startswith('{')]
if not ls: print('  NO LINE'); sys.exit()
d=json.loads(ls[-1]); print('  n_gpus', d['n_gpus'], 'value', round(d['value'],2), 'proofs/s aggregate; per rank', [(r['rank'], round(r['value'],1)) for r in d['per_rank']], 'scaling', d['scaling'], 'plan', d.get('stream_plan',{}).get('on'), 'identical', d.get('identical_to_oracle'))
PY
}
echo "bench.py --gpus 2 (its own ranks)"
timeout 600 python bench.py --gpus 2 --steps 16 --warmup 4 --e2e 0 --serial-proofs 0 > "$out/self_spawn.json" 2> "$out/self_spawn.err"; echo "  rc=$?"; show "$out/self_spawn.json"
echo "the driver's command for N = 2"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 > "$out/torchrun.json" 2> "$out/torchrun.err"; echo "  rc=$?"; show "$out/torchrun.json"
tail -3 "$out/torchrun.err" | cut -c1-300
