"""Worker of tests/test_multiprocess.py: the N > 1 path of bench.py on CPU — one process per rank over gloo, every rank
proving its own witnesses (TEST-ONLY emulator build of libzkhip), barrier + MAX-over-ranks timing, rank 0 prints the line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from emu_util import emu_library  # noqa: E402
from oracle import cpu  # noqa: E402
from zokrates_amd import native, parallel, synth  # noqa: E402


def main():
    steps = int(sys.argv[1])
    ranks = parallel.Ranks(backend="gloo")
    ctx = native.Context(0, emu_library())
    circ = synth.circuit(0, 4, seed=0xA11CE)                       # same circuit and key on every rank
    cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
    tox = synth.toxic_waste(0)
    pk = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, tox))
    zs = [circ.assignment(ranks.witness_seed(i)) for i in range(steps)]
    resident = [native.Assignment(ctx, cs, z) for z in zs]
    rs = [(11 + ranks.rank + i, 7 * i + 3) for i in range(steps)]
    ranks.barrier()
    t0 = time.perf_counter()
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, resident, rs)
    ranks.barrier()
    elapsed = ranks.max_over_ranks(time.perf_counter() - t0)
    oc = cpu.Circuit.from_csr(0, circ.n, circ.l, circ.w, circ.mats())
    tb = b"".join(int(v).to_bytes(32, "little") for v in tox)
    ok = all(p == cpu.trapdoor(oc, tb, z, r, s) for p, z, (r, s) in zip(proofs, zs, rs))
    distinct = len({bytes(z.tobytes()) for z in zs})
    all_ok = ranks.sum_over_ranks(1.0 if ok else 0.0)
    seeds = ranks.sum_over_ranks(float(ranks.witness_seed(0)))
    # latency mode: ONE proof across the ranks (sharded key, all-gather of the partial records)
    shard = native.ProvingKey(ctx, 0, native.setup_g16(ctx, cs, tox), rank=ranks.rank, world=ranks.world)
    z_common = circ.assignment(424242)
    sharded = parallel.prove_sharded(ranks, ctx, shard, cs, z_common, 31337, 271828)
    sharded_ok = ranks.sum_over_ranks(1.0 if sharded == cpu.trapdoor(oc, tb, z_common, 31337, 271828) else 0.0)
    # ... and with the shards BOUND to the system and the witness map split between the ranks (even ranks transform a, odd ranks b,
    # partners swap their halves over the process group): the same proof
    raw16 = native.setup_g16(ctx, cs, tox)
    shard.bind_shard(cs, raw16)
    split = parallel.prove_sharded(ranks, ctx, shard, cs, z_common, 31337, 271828, transform_split=True)
    whole_map = parallel.prove_sharded(ranks, ctx, shard, cs, z_common, 31337, 271828, transform_split=False)
    split_ok = ranks.sum_over_ranks(1.0 if split == whole_map == cpu.trapdoor(oc, tb, z_common, 31337, 271828) else 0.0)
    # the same for the second scheme: GM17 key sharded over the ranks, one record each, combined on every rank
    t4 = (tox[0], tox[1], tox[2], tox[4])
    tb17 = b"".join(int(v).to_bytes(32, "little") for v in t4)
    shard17 = native.ProvingKey(ctx, 0, native.setup_gm17(ctx, cs, t4), rank=ranks.rank, world=ranks.world, scheme="gm17")
    sharded17 = parallel.prove_sharded(ranks, ctx, shard17, cs, z_common, 31337, None, d1_d2=(1414, 1732))
    sharded17_ok = ranks.sum_over_ranks(1.0 if sharded17 == cpu.gm17_trapdoor(oc, tb17, z_common, 1414, 31337) else 0.0)
    if ranks.rank == 0:
        print(json.dumps({"sharded_bound_split_ok": split_ok, "sharded_gm17_ok": sharded17_ok, "sharded_ok": sharded_ok, "n_gpus": ranks.world, "steps": steps, "value": ranks.world * steps / elapsed, "ranks_ok": all_ok,
                          "distinct_witnesses_per_rank": distinct, "seed_sum": seeds, "scaling": "weak"}), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
