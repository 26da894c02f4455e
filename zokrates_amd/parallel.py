"""One process per GPU: the rank plumbing of bench.py (and of a multi-GPU prover service).

Throughput mode: the Groth16 hot path shards by independent proofs — every rank holds the full proving key and proves
its own witnesses, no data-path collective, only a barrier around the timed region and a MAX reduction of the elapsed
time.  Latency mode (`prove_sharded`): ONE proof over all ranks — every rank holds 1/world of the bases, computes the
partial sums of the five MSMs, and the ranks all-gather one 768-byte record each.
`backend="nccl"` (= RCCL on ROCm) on GPUs; `backend="gloo"` lets the same code run in CPU tests.
"""
import os


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = None
        self._dist = None
        self._torch = None
        self._cpu_group = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            # ZKHIP_DIST_BACKEND=gloo: test hook for boxes with fewer GPUs than ranks (RCCL wants one device per rank)
            self.backend = backend or os.environ.get("ZKHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.backend == "nccl":
                # LOCAL_RANK, unless the launcher narrowed this process's view to exactly one device; RCCL wants one GPU per
                # rank, so a rank without a GPU of its own is an error, never a silent second tenant of GPU 0
                ndev = torch.cuda.device_count()
                narrowed = ndev == 1 and (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
                if self.local_rank >= ndev and not narrowed:
                    raise RuntimeError(f"rank {self.rank} (LOCAL_RANK {self.local_rank}) has no GPU of its own: {ndev} visible")
                dev = self.local_rank if self.local_rank < ndev else 0
                torch.cuda.set_device(dev)
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
                # a host-side group beside it: ranks that only WAIT (rank 0 drives every GPU through the library's own
                # multi-GPU path after the timed region) must not park a spinning collective kernel on their GPU meanwhile
                self._cpu_group = dist.new_group(backend="gloo")
            else:
                dist.init_process_group(self.backend)

    def describe(self):
        """What the N ranks talk over: for the record in the bench line (the driver's scaling run can check that N RCCL ranks
        really formed)."""
        if self._dist is None:
            return "single process, no process group"
        if self.backend == "nccl":
            try:
                ver = ".".join(str(v) for v in self._torch.cuda.nccl.version())
            except Exception:      # (a torch build without the query: the rank count is what matters)
                ver = "?"
            return f"torch.distributed nccl (= RCCL {ver}), {self._dist.get_world_size()} rank(s), one GPU each"
        return f"torch.distributed {self.backend}, {self._dist.get_world_size()} rank(s)"

    def barrier(self):
        """Barrier + device synchronisation (libzkhip calls return only after their streams have drained; the explicit
        torch.cuda.synchronize covers the collective itself)."""
        if self._dist is not None:
            self._dist.barrier()
            if self.backend == "nccl":
                self._torch.cuda.synchronize()

    def host_barrier(self):
        """Barrier on the host only (gloo beside nccl): for waits of unknown length during which the GPUs belong to someone else."""
        if self._dist is not None:
            self._dist.barrier(group=self._cpu_group) if self._cpu_group is not None else self._dist.barrier()

    def max_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = self._torch.tensor([float(value)], dtype=self._torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = self._torch.tensor([float(value)], dtype=self._torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def all_gather_bytes(self, record):
        """Every rank contributes one fixed-size uint8 record; returns the list of all ranks' records (rank order).
        On GPUs this is an RCCL all-gather of device tensors over xGMI; the records are a few hundred bytes."""
        import numpy as np
        record = np.ascontiguousarray(record, dtype=np.uint8)
        if self._dist is None:
            return [record]
        torch = self._torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        mine = torch.from_numpy(record.copy()).to(dev)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
        return [t.cpu().numpy() for t in out]

    def exchange_bytes(self, partner, record):
        """This rank's uint8 record to `partner`, the partner's back (a pairwise send / receive: RCCL over xGMI on GPUs, gloo in the
        CPU tests) — the halves of a split witness map are N x 32 bytes each, an all-gather would move world / 2 copies of them."""
        import numpy as np
        record = np.ascontiguousarray(record, dtype=np.uint8)
        if self._dist is None or partner == self.rank:
            return record
        torch, dist = self._torch, self._dist
        dev = "cuda" if self.backend == "nccl" else "cpu"
        mine = torch.from_numpy(record.copy()).to(dev)
        theirs = torch.empty_like(mine)
        ops = [dist.P2POp(dist.isend, mine, partner), dist.P2POp(dist.irecv, theirs, partner)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return theirs.cpu().numpy()

    def send_bytes(self, to, record):
        import numpy as np
        t = self._torch.from_numpy(np.ascontiguousarray(record, dtype=np.uint8).copy()).to("cuda" if self.backend == "nccl" else "cpu")
        self._dist.send(t, to)

    def recv_bytes(self, src, size):
        t = self._torch.empty(size, dtype=self._torch.uint8, device="cuda" if self.backend == "nccl" else "cpu")
        self._dist.recv(t, src)
        return t.cpu().numpy()

    def witness_seed(self, step):
        """Distinct witnesses per (rank, step): ranks never prove the same statement twice."""
        return 0x5EED0000 + self.rank * 1000 + step

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None


SPLIT_MIN_LOG = int(os.environ.get("ZKHIP_SPLIT_MIN_LOG", "18"))     # the library's own threshold (zkhip_ctx: split_min_log)


def prove_sharded(ranks, ctx, pk_shard, cs, z, r, s, d1_d2=None, transform_split=None):
    """One proof across all ranks.  `pk_shard` = native.ProvingKey(..., rank=ranks.rank, world=ranks.world); z and the
    blinding scalars are the same on every rank (Groth16: r, s; GM17: d1_d2 = (d1, d2) and r, `s` ignored).  Every rank
    returns the (identical) proof bytes."""
    from . import native
    if getattr(pk_shard, "scheme", "g16") == "gm17":
        d1, d2 = d1_d2
        parts = ranks.all_gather_bytes(native.prove_gm17_partial(ctx, pk_shard, cs, z, d1, d2, r))
        return native.combine_gm17(ctx, pk_shard, parts, d1, d2, r)
    split = transform_split if transform_split is not None else (ranks.world >= 2 and pk_shard.is_bound(cs) and (pk_shard.hlen + 1) >= (1 << SPLIT_MIN_LOG))
    if split and ranks.world >= 2:
        # the witness map split between the ranks (bound shards: a and b on the coset are all a proof needs of it): even ranks
        # transform a, odd ranks b, partners swap their halves
        half = ranks.rank & 1
        partner = ranks.rank ^ 1 if (ranks.rank ^ 1) < ranks.world else ranks.rank - 1
        mine = native.prove_g16_split_begin(ctx, pk_shard, cs, z, r, s, half)
        if (ranks.rank ^ 1) < ranks.world:
            theirs = ranks.exchange_bytes(partner, mine)
        else:      # the odd rank out (world is odd): it only receives — its partner has a partner of its own
            theirs = ranks.recv_bytes(partner, mine.size)
        if ranks.world % 2 == 1 and ranks.rank == ranks.world - 2:
            ranks.send_bytes(ranks.world - 1, mine)          # ... which also serves the rank without one
        part = native.prove_g16_split_end(ctx, pk_shard, cs, theirs)
    else:
        part = native.prove_g16_partial(ctx, pk_shard, cs, z, r, s)
    parts = ranks.all_gather_bytes(part)
    return native.combine_g16(ctx, pk_shard, parts, r, s)
