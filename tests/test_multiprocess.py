"""bench.py's multi-process path on CPU: world_size 2 over gloo (the GPU run uses the same zokrates_amd.parallel code
with backend nccl = RCCL).  Independent proofs per rank, no data-path collective."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_gloo():
    world, steps = 2, 3
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker.py"), str(steps)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["scaling"] == "weak"
    assert line["ranks_ok"] == 2.0                      # every rank's proofs equal the oracle's
    assert line["sharded_ok"] == 2.0                    # the proof sharded over both ranks equals the oracle's on both
    assert line["sharded_gm17_ok"] == 2.0               # and so does the GM17 proof
    assert line["sharded_bound_split_ok"] == 2.0        # bound shards with the witness map split between the ranks: the same proof
    assert line["distinct_witnesses_per_rank"] == steps
    assert line["seed_sum"] == 2 * 0x5EED0000 + 1000    # ranks drew different witness seeds
    assert line["value"] > 0
    assert not any(l.startswith("{") for l in outs[1][0].splitlines())   # only rank 0 prints the result line
