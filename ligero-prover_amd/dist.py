"""Multi-GPU plumbing for the prover: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm, "gloo" for the CPU tests).  Traces (proof jobs) are independent objects, so the data path needs no
collective: jobs are dealt round-robin to ranks, each rank proves its own jobs on its own GPU, and only the
timing / bookkeeping is reduced (MAX of the wall time, SUM of the processed units)."""
import os


class Group:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def _tensor(self, values, dtype):
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        return torch.tensor(values, dtype=dtype, device=dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = self._tensor([float(x)], torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return int(x)
        import torch
        t = self._tensor([int(x)], torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def my_jobs(self, n_jobs):
        """round-robin deal of job ids: every job has exactly one owner"""
        return list(range(self.rank, n_jobs, self.world))

    def gather_digests(self, digest32):
        """all ranks learn every rank's 32-byte result digest (used to cross-check proofs in tests)"""
        if self.dist is None:
            return [bytes(digest32)]
        import torch
        mine = self._tensor(list(bytes(digest32)), torch.uint8)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [bytes(o.cpu().tolist()) for o in out]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
