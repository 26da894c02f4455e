"""One process per GPU: the rank plumbing of bench.py (and of a multi-GPU prover service).

The Groth16 hot path shards by independent proofs: every rank holds the full proving key and proves its own witnesses,
so there is no data-path collective — only a barrier around the timed region and a MAX reduction of the elapsed time.
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
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            self.backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        """Barrier + device synchronisation (libzkhip calls return only after their streams have drained; the explicit
        torch.cuda.synchronize covers the collective itself)."""
        if self._dist is not None:
            self._dist.barrier()
            if self.backend == "nccl":
                self._torch.cuda.synchronize()

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

    def witness_seed(self, step):
        """Distinct witnesses per (rank, step): ranks never prove the same statement twice."""
        return 0x5EED0000 + self.rank * 1000 + step

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None
