timeout 300 python bench.py > gpurun_out/bench_r1u.json 2> gpurun_out/bench_r1u.err; cut -c1-110 gpurun_out/bench_r1u.json
timeout 300 python bench.py --scheme gm17 > gpurun_out/bench_r1u_gm17.json 2>/dev/null; cut -c1-110 gpurun_out/bench_r1u_gm17.json
timeout 300 python bench.py --curve bls12_381 --log-domain 18 --kind poseidon > gpurun_out/bench_r1u_poseidon.json 2>/dev/null; cut -c1-110 gpurun_out/bench_r1u_poseidon.json
bash tools/profile_round.sh r1u > /dev/null 2>&1; head -12 gpurun_out/r1u_g16_kernel_stats.md | cut -c1-100
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
