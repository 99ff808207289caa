"""Multi-GPU plumbing for the prover: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm, "gloo" for the CPU tests).  Traces (proof jobs) are independent objects, so the data path needs no
collective: jobs are dealt round-robin to ranks, each rank proves its own jobs on its own GPU, and only the
timing / bookkeeping is reduced (MAX of the wall time, SUM of the processed units)."""
import os


class Group:
    def __init__(self, backend=None, force_init=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1 or force_init:
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

    # ---- collectives for the sharded prover (include/lig_hip.h: lig_comm)
    TRANSPORTS = ("rccl-stream", "rccl-sync", "ipc-stream", "ipc-sync", "torch", "host")

    def default_transport(self):
        if os.environ.get("LIG_COMM") == "ipc":
            return "ipc-stream"
        return "rccl-stream" if self.backend == "nccl" else "host"

    def make_comm(self, pkg, ctx, transport=None):
        """pkg = the ligero_prover_amd module, ctx = its Context on this rank's GPU.  Transports (bench.py tries them in order):
          rccl-stream  the product path -- the library's own RCCL communicator (csrc/comm_rccl.hip: grouped ncclSend/ncclRecv
                       and ncclAllGather enqueued on the context's HIP streams); the process group only carries the
                       128-byte unique id from rank 0 to the others
          rccl-sync    the same communicator through its host-synchronous forms (collective, then hipStreamSynchronize):
                       no stream-ordered overlap, no cross-stream events around RCCL kernels
          ipc-stream   process-to-process over mapped device memory (csrc/comm_ipc.hip), GPU-ordered on flags in shared memory:
                       one device shared by all ranks (tests) or one device per rank with peer access
          ipc-sync     its host-synchronous forms
          torch        host-synchronous callbacks over torch.distributed's OWN communicator on device memory, zero-copy
                       (all_to_all_single / all_gather_into_tensor of the backend "nccl" = RCCL inside torch)
          host         host-synchronous callbacks staged through host memory (gloo; the CPU-side tests) or plain copies (1 rank)
        Every communicator made here is ended by close(), newest first."""
        import ctypes as C
        import numpy as np
        import torch
        g = self
        transport = transport or self.default_transport()
        if transport not in self.TRANSPORTS:
            raise ValueError("unknown transport %r" % (transport,))
        self.transport = transport
        if not hasattr(self, "_comms"):
            self._comms = []

        def sync_only(comm):
            """a copy of `comm` without the stream-ordered forms: lig_shard_* then takes its host-synchronous branch.  The
            original stays the handle the library's *_comm_destroy recognises."""
            c2 = pkg.Comm()
            C.memmove(C.byref(c2), C.byref(comm), C.sizeof(pkg.Comm))
            c2.all_to_all_on = pkg.A2A_ON_FN()
            c2.all_gather_on = pkg.A2A_ON_FN()
            return c2

        if transport in ("ipc-stream", "ipc-sync"):
            # torch.distributed (any backend) is only the launcher's rendezvous.  Every communicator needs a fresh segment
            # name, the same on all ranks: launcher tag + rendezvous port + a per-group counter
            self._ipc_n = getattr(self, "_ipc_n", 0) + 1
            name = "/lig_ipc_%s_%s_%d" % (os.environ.get("LIG_COMM_TAG", "0"), os.environ.get("MASTER_PORT", "0"), self._ipc_n)
            comm = ctx.ipc_comm(name, g.rank, g.world)
            self._comms.append(("ipc", ctx, comm))
            return comm if transport == "ipc-stream" else sync_only(comm)
        if transport in ("rccl-stream", "rccl-sync"):
            uid = ctx.rccl_unique_id() if g.rank == 0 else bytes(128)
            if g.dist is not None:
                t = self._tensor(list(uid), torch.uint8)
                g.dist.broadcast(t, src=0)
                uid = bytes(t.cpu().tolist())
            comm = ctx.rccl_comm(uid, g.rank, g.world)
            self._comms.append(("rccl", ctx, comm))
            return comm if transport == "rccl-stream" else sync_only(comm)

        def host_of(ptr, nbytes):        # device -> host numpy copy through the library's own stream
            return ctx.download(C.c_void_p(ptr), (nbytes,), dtype=np.uint8)

        class DevBytes:                  # a raw device range as a zero-copy torch tensor (__cuda_array_interface__)
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

        def dev_tensor(ptr, nbytes):
            return torch.as_tensor(DevBytes(ptr, nbytes), device=torch.device("cuda", torch.cuda.current_device()))

        on_device = transport == "torch"
        if on_device and (g.dist is None or g.backend != "nccl") and g.world > 1:
            raise ValueError("transport 'torch' needs the nccl (= RCCL) process group")

        def all_to_all(user, send, recv, block):
            try:
                total = block * g.world
                if g.dist is None:
                    ctx.check(ctx.L.lig_copy(ctx.h, C.c_void_p(recv), C.c_void_p(send), total)); ctx.sync()
                elif on_device:              # the library drained its stream before the call; torch's stream is drained before we return
                    g.dist.all_to_all_single(dev_tensor(recv, total), dev_tensor(send, total))
                    torch.cuda.synchronize()
                else:                      # gloo has no all_to_all: gather everything, keep the blocks addressed to me
                    mine = torch.from_numpy(host_of(send, total))
                    parts = [torch.empty_like(mine) for _ in range(g.world)]
                    g.dist.all_gather(parts, mine)
                    out = torch.cat([p[g.rank * block:(g.rank + 1) * block] for p in parts]).numpy()
                    ctx.write(C.c_void_p(recv), out)
                return 0
            except Exception as e:           # never let an exception cross the C boundary
                print("lig comm all_to_all failed:", repr(e), flush=True)
                return 1

        def all_gather(user, send, recv, nbytes):
            try:
                if g.dist is None:
                    ctx.check(ctx.L.lig_copy(ctx.h, C.c_void_p(recv), C.c_void_p(send), nbytes)); ctx.sync()
                elif on_device:
                    g.dist.all_gather_into_tensor(dev_tensor(recv, nbytes * g.world), dev_tensor(send, nbytes))
                    torch.cuda.synchronize()
                else:
                    mine = torch.from_numpy(host_of(send, nbytes))
                    parts = [torch.empty_like(mine) for _ in range(g.world)]
                    g.dist.all_gather(parts, mine)
                    ctx.write(C.c_void_p(recv), torch.cat(parts).numpy())
                return 0
            except Exception as e:
                print("lig comm all_gather failed:", repr(e), flush=True)
                return 1

        comm = pkg.Comm()
        comm.user = None
        comm.all_to_all = pkg.A2A_FN(all_to_all)
        comm.all_gather = pkg.A2A_FN(all_gather)
        self._comms.append(("callbacks", ctx, (all_to_all, all_gather, comm)))     # keeps the ctypes thunks alive
        return comm

    def rccl_ranks(self):
        """ncclCommCount of the newest RCCL communicator made here (None: the transport in use is not RCCL)"""
        for kind, ctx, comm in reversed(getattr(self, "_comms", [])):
            if kind == "rccl":
                return ctx.rccl_comm_count(comm)
            break
        return None

    def close(self):
        for kind, ctx, comm in reversed(getattr(self, "_comms", [])):
            if kind == "ipc":
                ctx.ipc_comm_destroy(comm)
            elif kind == "rccl":
                ctx.rccl_comm_destroy(comm)
        self._comms = []
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
