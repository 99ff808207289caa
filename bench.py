#!/usr/bin/env python3
"""bench.py -- prover constraints/s of the MI355X-native Ligero hot path on synthetic BN254 traces.

  python bench.py --gpus N --steps K --warmup W [--workload full|encode] [--log2-constraints C]

One "step" = one pass of the hot path over one synthetic trace whose witness matrix is already resident in
HBM (generated on the GPU with the reference's AES-256-CTR field sampler, key SHA256("lig-synth"||le64(1))):
  full   : configs[2] of BASELINE.json (default) -- 2^24 constraints = 2098 rows of l=8000 (+3 mask rows):
           row forming (pads/masks from the encoding stream), RS-encode, column SHA-256, Merkle root, stage-2
           randomness rows (sampled + encoded on the GPU) and code/linear/quadratic accumulators, Fiat-Shamir
           seeds, column sampling, decommitment, prover self-check (3 decodes), column gather, protobuf envelope.
           The timed region ends with the proof bytes in host memory.
  encode : configs[1] -- 2^20 constraints = 132 rows, RS-encode only (INTT_k + NTT_4k)
N > 1: one process per GPU over RCCL.  Under torch.distributed.run the ranks exist already (WORLD_SIZE must equal --gpus);
a plain `python bench.py --gpus N` spawns the N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU each).
`value`: traces are independent objects, so every rank proves its own traces (weak scaling, no data-path collective); the
timed region is bracketed by barrier + torch.cuda.synchronize() and the MAX over ranks is reported.  For N > 1 the line also
carries `sharded` (+ `sharded_2p24`, `preflight`): configs[3] -- ONE 2^26-constraint trace row-sharded over the N ranks
(column-partitioned hash after a per-round all-to-all of codeword column slices, all-gathered leaves / partial sums / opened
columns), with its own barrier-bracketed timing, the number of ranks RCCL reports and the comparison with the oracle's pin.  That
leg runs rung by rung in FRESH child processes (sharded_ladder: stream-ordered RCCL -> host-synchronous RCCL -> torch's own
communicator -> process-to-process over mapped memory), so a failing or hanging collective cannot cost the weak figure.
N = 1 also reports, beside `value`: proof_wall_ms, quad_mix (half of the constraints quadratic), value_incl_h2d / incl_h2d
(witness from pinned host memory; .caller_rands: the randomness rows too; .narrow_format), the verifier, cpu_baseline.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes as C
import hashlib
import importlib.util
import json
import os
import sys
import time

# Before the HIP runtime initialises: enough hardware queues for every stream of this process.  The runtime folds the streams of a process
# onto GPU_MAX_HW_QUEUES (default 4) queues; the library needs a queue of its own for each proof's main stream and for the ONE side
# stream the proofs share (profiles/r06_stream_map_ab.md: 2.04e9 constraints/s, 1.83e9 when a main stream lands in a shared queue).  At
# N = 1 that is null + 2 main + side = 4; a rank of an N > 1 run also owns torch's / RCCL's streams.  With the shared side stream 8 queues
# cost nothing (2.04e9 with 4, 6 and 8).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.abspath(__file__))
L_, K_, N_, T_ = 8000, 8192, 32768, 192
NO_VERIFY = False


def load_pkg():
    path = os.path.join(ROOT, "ligero-prover_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("ligero_prover_amd", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ligero_prover_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def synth_key(seed=1):
    return hashlib.sha256(b"lig-synth" + int(seed).to_bytes(8, "little")).digest()


class EncodeWorkload:
    """configs[1]: R = ceil(C / l) message rows -> codewords"""
    name = "encode"

    def __init__(self, ctx, constraints):
        self.ctx = ctx
        self.constraints = constraints
        self.rows = -(-constraints // L_)
        self.msgs = ctx.malloc(self.rows * K_ * 32)
        self.cws = ctx.malloc(self.rows * N_ * 32)
        ctx.rng_fill(synth_key(), 0, self.msgs, self.rows * K_)
        ctx.sync()

    def step(self):
        self.ctx.encode_rows(self.msgs, self.cws, self.rows)

    def encodes_per_step(self):
        return self.rows

    def describe(self):
        return {"workload": "configs[1]: 2^%d-constraint synthetic BN254 witness, RS-encode only (INTT_k + NTT_4k)"
                            % (self.constraints.bit_length() - 1),
                "rows": self.rows, "l": L_, "k": K_, "n": N_}

    def close(self):
        pass


class FullWorkload:
    """configs[2]: full proof of a C-constraint trace (linear constraints, dense stage-2 randomness).
    inflight = M > 1: a step proves M such traces concurrently (one context = one set of HIP streams per trace, one
    host thread each), so that one proof's host-only phases (the sequential 3 MiB stage-2 seed hash, decommitment,
    envelope) are covered by the other proof's GPU work.  Default 1: a step is one proof, start to finish."""
    name = "full"

    def __init__(self, ctx, constraints, pkg=None, inflight=1, device=0, quad_percent=0):
        self.ctx = ctx
        self.constraints_per_trace = constraints
        self.inflight = inflight
        self.constraints = constraints * inflight
        # quad_percent > 0: that share of the constraints are quadratic slots x*y = z (three committed rows x, y, z per l of them,
        # each with its dense randomness row in this synthetic stream, plus the quadratic test: nonbatch_context.hpp:771-780),
        # the rest linear -- SURVEY.md 8(d)'s optional mix
        self.quad_percent = quad_percent
        self.n_quad = (constraints * quad_percent) // 100
        self.n_lin = constraints - self.n_quad
        self.ctxs = [ctx] + [pkg.Context(L_, K_, N_, device=device) for _ in range(inflight - 1)]
        self.traces = [c.synth_prepare(self.n_lin, self.n_quad, synth_seed=1, generated_at=0) for c in self.ctxs]
        self.rows = -(-self.n_lin // L_) + 3 * -(-self.n_quad // L_)
        self.last = None
        self.pool = None
        if inflight > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=inflight)
        for c in self.ctxs:
            c.sync()

    def _one(self, i):
        (addr, length), info = self.ctxs[i].synth_prove(self.traces[i], copy=False)     # proof bytes stay in the pinned buffer
        if not (info.valid_code and info.valid_linear and info.valid_quad):
            raise SystemExit("prover self-check failed")
        if i == 0:
            self.last_info = info
        return (addr, length, info.ms_stage1, info.ms_stage2, info.ms_stage3, info.ms_total)

    def step(self):
        self.run(1)

    def run(self, steps):
        """`steps` steps = steps x inflight proofs; with several proofs in flight every host thread proves `steps` traces
        back to back, so the proofs drift out of phase as they would in a proving service"""
        if self.pool is None:
            for _ in range(steps):
                self.last = self._one(0)
        else:                                   # ctypes releases the GIL for the duration of each call
            self.last = list(self.pool.map(lambda i: [self._one(i) for _ in range(steps)][-1], range(self.inflight)))[0]

    def single_proof_ms(self, reps=3):
        """wall time of one proof with nothing else in flight (latency figure next to the throughput figure)"""
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            self._one(0)
            dt = 1e3 * (time.perf_counter() - t0)
            best = dt if best is None or dt < best else best
        return best

    def encodes_per_step(self):
        return (2 * self.rows + 1) * self.inflight        # message rows + randomness rows (+ code mask on the fast path)

    def describe(self):
        lg = self.constraints_per_trace.bit_length() - 1
        d = {"workload": "%s2^%d-constraint trace, full proof (encode + column SHA-256 + Merkle + RLC checks + "
                         "sampling + envelope), witness matrix resident in HBM"
                         % ("configs[2]: " if lg == 24 else "configs[3] trace on one GPU: " if lg == 26 else "", lg),
             "rows": self.rows + 3, "l": L_, "k": K_, "n": N_, "sample_size": T_, "proofs_in_flight": self.inflight}
        if self.quad_percent:
            d.update(n_linear=self.n_lin, n_quad=self.n_quad,
                     workload=d["workload"].replace("full proof", "%d %% quadratic constraints (x*y = z slots, 3 committed rows per %d), full proof" % (self.quad_percent, L_)))
        if self.last:
            proof = C.string_at(self.last[0], self.last[1])               # after the timed region
            d.update(proof_bytes=len(proof), proof_sha256=hashlib.sha256(proof).hexdigest(),
                     stage_ms={"stage1": self.last[2], "stage2": self.last[3], "stage3": self.last[4]},
                     proof_latency_ms=self.last[5])
            # outside the timed region: the HIP verifier on that proof (informational; --no-verify skips it so that a
            # profiler run of this command sees the prover's kernel launches only)
            pkg = sys.modules["ligero_prover_amd"]
            job = pkg.Context.make_job(self.n_lin, self.n_quad, synth_seed=1, generated_at=0)
            info = self.last_info
            if not NO_VERIFY:
                self.ctx.synth_verify(job, None, proof)            # first call: its buffers are allocated right after the workloads freed theirs
                best = None
                for _ in range(3):                                 # the verifier derives the linear constant from the public statement
                    t0 = time.perf_counter()
                    v = self.ctx.synth_verify(job, None, proof)
                    wall = 1e3 * (time.perf_counter() - t0)
                    if best is None or v.ms_total < best[0]:
                        best = (v.ms_total, wall, bool(v.accept))
                d.update(verifier_accepts=best[2], verify_ms=best[0], verify_ms_with_python_copies=best[1])
            pin_path = os.path.join(ROOT, "tests", "golden", "full_pin_2p%d%s.json" % (lg, "_q%d" % self.quad_percent if self.quad_percent else ""))
            if os.path.exists(pin_path):                      # the oracle's reference-structured prover on this exact job
                with open(pin_path) as f:
                    pin = json.load(f)
                d["proof_equals_oracle_pin"] = bool(d["proof_sha256"] == pin["proof_sha256"] and bytes(info.root).hex() == pin["root"])
        return d

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
        for c, t in zip(self.ctxs, self.traces):
            c.trace_destroy(t)
        for c in self.ctxs[1:]:
            c.close()


class RowsFromHostWorkload:
    """the same full proof, but the witness matrix starts in (pinned) HOST memory and goes through the caller-rows entry
    (lig_rows_restart -> lig_rows_commit -> lig_rows_prove): the upload of every trace is inside the timed region, chunked
    on the copy stream under the encodes; with two traces in flight the upload of one also overlaps the proof of the
    other.  The dense randomness rows of the synthetic stream are produced on the device from the stage-1 seed
    (lig_rng_fill_rows), as its public definition allows.  This is the PCIe-inclusive figure (value_incl_h2d)."""
    name = "rows_from_host"

    def __init__(self, ctx, constraints, pkg, inflight, device, narrow_bytes=0, caller_rands=False):
        import numpy as np
        import torch
        self.pkg, self.inflight, self.narrow_bytes = pkg, inflight, narrow_bytes
        self.caller_rands, self.rands = caller_rands, None
        self.constraints_per_trace = constraints
        self.constraints = constraints * inflight
        R = -(-constraints // L_)
        per_row = np.full(R, L_, dtype=np.uint32)
        if constraints % L_:
            per_row[-1] = constraints % L_
        self.per_row = per_row
        self.ctxs = [ctx] + [pkg.Context(L_, K_, N_, device=device) for _ in range(inflight - 1)]
        # the witness matrix in pinned host memory (generated on the device once, outside the timed region)
        self.host = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
        d = ctx.malloc(R * K_ * 32)
        ctx.rng_fill_rows(synth_key(), 0, per_row, d)
        ctx.check(ctx.L.lig_read(ctx.h, C.c_void_p(self.host.data_ptr()), d, R * K_ * 32))
        ctx.free(d)
        kinds = np.full(R, pkg.ROW_KINDS["LINEAR"] | pkg.ROW_DRAW_PAD, dtype=np.uint8)
        self.widths = None
        if narrow_bytes:
            # a small-witness trace in the narrow row format (lig_rows_job.elem_bytes): the low narrow_bytes of every witness,
            # only the l data slots of every row are shipped (the library draws the pads): 8000 x 8 B instead of 8192 x 32 B per row
            w = narrow_bytes // 4
            packed = self.host[:, :L_, :w].contiguous()
            self.host = torch.empty(packed.shape, dtype=torch.int32, pin_memory=True)
            self.host.copy_(packed)
            self.widths = np.full(R, narrow_bytes, dtype=np.uint8)
        self.traces, self.keep = [], []
        for c in self.ctxs:
            t, keep = self._begin(c, kinds)
            self.traces.append(t)
            self.keep.append(keep)
        self.loaded = [True] * inflight          # lig_rows_begin started the first upload
        self.last = None
        if caller_rands:
            # what a real constraint generator does (nonbatch_context.hpp:654-780): after the commit it derives one dense 256-bit
            # randomness row per linear row from the stage-1 seed, in HOST memory.  Here: the dense rows of the synthetic stream (so the
            # proof is the pinned one), generated once -- the trace and therefore the seed are the same in every step -- and kept in
            # pinned host memory; every lig_rows_prove of the timed region ships them (another 550 MB per 2^24 constraints).
            c, t = self.ctxs[0], self.traces[0]
            _, seed = c.rows_commit(t)
            d = ctx.malloc(R * K_ * 32)
            ctx.rng_fill_rows(seed, 0, per_row, d)
            self.rands = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
            ctx.check(ctx.L.lig_read(ctx.h, C.c_void_p(self.rands.data_ptr()), d, R * K_ * 32))
            ctx.free(d)
            c.rows_prove(t, self.rands.data_ptr(), None, copy=False)
            self.loaded[0] = False
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=inflight)

    def _begin(self, c, kinds):
        job = self.pkg.RowsJob()
        job.rows = len(kinds)
        job.kinds = kinds.ctypes.data
        job.msgs = self.host.data_ptr()
        job.msgs_on_device = 0
        for i in range(32):
            job.encoding_seed[i] = i
            job.program_hash[i] = 0
        job.generated_at = 0
        job.version = b"1.5.0"
        job.set_public_args(None)
        if not self.caller_rands:
            job.dense_rands_per_row = self.per_row.ctypes.data      # the synthetic stream's dense coefficient rows: sampled on the device
        if self.widths is not None:
            job.elem_bytes = self.widths.ctypes.data
        t = C.c_void_p()
        c.check(c.L.lig_rows_begin(c.h, C.byref(job), C.byref(t)))
        return t, (kinds, job)

    def _loop(self, i, steps):
        """commit(s) -> restart(s+1) -> prove(s): the upload of the next trace runs under the proof of the current one"""
        c, t = self.ctxs[i], self.traces[i]
        host = C.c_void_p(self.host.data_ptr())
        out = None
        for s_ in range(steps):
            if not self.loaded[i]:
                c.check(c.L.lig_rows_restart(t, host, 0))
            c.rows_commit(t)
            self.loaded[i] = s_ + 1 < steps
            if self.loaded[i]:
                c.check(c.L.lig_rows_restart(t, host, 0))
            (addr, length), info = c.rows_prove(t, self.rands.data_ptr() if self.caller_rands else None, None, copy=False)
            if not (info.valid_code and info.valid_linear and info.valid_quad):
                raise SystemExit("prover self-check failed")
            out = (addr, length)
        return out

    def run(self, steps):
        self.last = list(self.pool.map(lambda i: self._loop(i, steps), range(self.inflight)))[0]

    def proof_sha256(self):
        return hashlib.sha256(C.string_at(self.last[0], self.last[1])).hexdigest()

    def close(self):
        self.pool.shutdown()
        for c, t in zip(self.ctxs, self.traces):
            c.trace_destroy(t)
        for c in self.ctxs[1:]:
            c.close()


class ShardedWorkload:
    """configs[3]-style: ONE C-constraint trace sharded over all ranks (row blocks per GPU, column-partitioned hash
    after one all-to-all, all-gathered leaves / partial accumulators / opened columns).  Strong scaling."""
    name = "sharded"

    def __init__(self, ctx, constraints, group, pkg, comm=None):
        self.ctx, self.constraints, self.group = ctx, constraints, group
        self.job = pkg.Context.make_job(constraints, 0, synth_seed=1, generated_at=0)
        self.comm = comm if comm is not None else group.make_comm(pkg, ctx)
        self.shard = ctx.shard_prepare(self.job, group.rank, group.world, self.comm)
        self.rows = -(-constraints // L_)
        self.rounds = pkg.shard_plan(self.job, L_, group.world)[0]
        self.last = None
        ctx.sync()

    def step(self):
        (addr, length), info = self.ctx.shard_prove(self.shard, copy=False)
        if not (info.valid_code and info.valid_linear and info.valid_quad):
            raise SystemExit("prover self-check failed")
        self.last = (addr, length, info.ms_stage1, info.ms_stage2, info.ms_stage3, info.ms_total)
        self.last_info = info

    def describe(self):
        d = {"workload": "configs[3]-style: ONE 2^%d-constraint trace row-sharded over the GPUs, full proof"
                         % (self.constraints.bit_length() - 1),
             "rows": self.rows + 3, "l": L_, "k": K_, "n": N_, "sample_size": T_}
        if self.last:
            proof = C.string_at(self.last[0], self.last[1])
            d.update(proof_bytes=len(proof), proof_sha256=hashlib.sha256(proof).hexdigest(),
                     stage_ms={"stage1": self.last[2], "stage2": self.last[3], "stage3": self.last[4]}, proof_latency_ms=self.last[5])
        return d

    def close(self):
        self.ctx.shard_destroy(self.shard)


class StubWorkload:
    """no GPU work at all: the launcher / process-group plumbing of this script under test on CPU (gloo), tests/test_bench_launcher.py"""
    name = "stub"
    constraints = 1000

    def step(self):
        time.sleep(0.001)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (one process per GPU) and relay rank 0's output"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()),
               LIG_BENCH_SPAWNED="1")
    # The pool's host driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's own P2P set-up (and any
    # hipIpcGetMemHandle) fails with "invalid argument".  The image exports it already; it is only set here when the caller's
    # environment lost it, never overridden, and the value the ranks ran with is recorded in the line (preflight).
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):          # one rank died: the others would wait in a collective forever
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.wait()
            break
        time.sleep(0.05)
    return max(abs(rc) for rc in rcs)


def run_stub(a, lig_dist):
    """--workload stub: barrier / MAX-over-ranks / single JSON line with no GPU in sight"""
    group = lig_dist.Group(a.backend or "gloo")
    wl = StubWorkload()
    for _ in range(a.warmup):
        wl.step()
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
    group.barrier()
    dt = group.max_over_ranks(time.perf_counter() - t0)
    total = group.sum_over_ranks(wl.constraints * a.steps)
    extra = {}
    if a.sharded_leg:
        results, attempts = sharded_ladder(a, group, group.rank, group.world, group.local_rank, workload="stub")
        extra = {"sharded": dict(results.get(str(a.sharded_log2).split(",")[-1]) or {"error": "no rung delivered"}, attempts=attempts)}
    if group.rank == 0:
        print(json.dumps({"metric": "prover constraints/sec", "value": total / dt, "unit": "constraints/s", "n_gpus": group.world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": "stub (launcher test, no GPU work)"},
                          "spawned_by_bench": bool(os.environ.get("LIG_BENCH_SPAWNED")), **extra}), flush=True)
    group.close()


TRANSPORT_NOTES = {
    "rccl-stream": "librccl, stream-ordered: grouped ncclSend/ncclRecv per round + ncclAllGather (leaves, partial sums, opened columns) on the context's HIP streams",
    "rccl-sync": "librccl, host-synchronous forms of the same communicator (collective, then hipStreamSynchronize)",
    "ipc-stream": "comm_ipc: peers map each other's buffers (hipIpcMemHandle), GPU-ordered on flags in shared memory",
    "ipc-sync": "comm_ipc, host-synchronous forms",
    "torch": "torch.distributed's own RCCL communicator (all_to_all_single / all_gather_into_tensor on the library's device buffers), host-synchronous",
    "host": "host-synchronous callbacks staged through host memory (gloo) / plain copies (one rank)",
}


def sharded_leg(ctx, group, pkg, log2c, steps, warmup, fence, comm=None, transport=None):
    """configs[3]: ONE trace sharded over all ranks, timed like the main region (barrier + synchronize on both sides, MAX over
    ranks).  Collective: every rank calls it."""
    swl = ShardedWorkload(ctx, 1 << log2c, group, pkg, comm)
    transport = transport or getattr(group, "transport", None)
    try:
        for _ in range(warmup):
            swl.step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            swl.step()
        fence()
        dt = group.max_over_ranks(time.perf_counter() - t0)
        d = swl.describe()
        out = {"workload": d["workload"], "log2_constraints": log2c, "rows": d["rows"], "ranks": group.world, "steps": steps, "warmup": warmup,
               "ms_per_proof": 1e3 * dt / steps, "constraints_per_s": (1 << log2c) * steps / dt, "scaling": "strong",
               "stage_ms": d.get("stage_ms"), "proof_bytes": d.get("proof_bytes"), "proof_sha256": d.get("proof_sha256"),
               "exchange_rounds": swl.rounds, "transport": transport, "collectives": TRANSPORT_NOTES.get(transport)}
        out["rccl_ranks"] = group.rccl_ranks() if hasattr(group, "rccl_ranks") else None       # ncclCommCount of the communicator the proofs ran on
        if out["rccl_ranks"] is not None:
            out["rccl_library"] = pkg.rccl_available()[1]
        pin_path = os.path.join(ROOT, "tests", "golden", "full_pin_2p%d.json" % log2c)
        if os.path.exists(pin_path):
            with open(pin_path) as f:
                pin = json.load(f)
            out["proof_equals_oracle_pin"] = bool(out["proof_sha256"] == pin["proof_sha256"] and bytes(swl.last_info.root).hex() == pin["root"])
        digs = group.gather_digests(hashlib.sha256(C.string_at(swl.last[0], swl.last[1])).digest())
        out["all_ranks_same_envelope"] = len(set(digs)) == 1
        return out
    finally:
        swl.close()


RESULT_TAG = "LIG_SHARDED_RESULT "


def ladder_transports(world):
    """the rungs of the sharded leg, first = the product path.  Every rung runs in FRESH processes (one child per rank, its own
    rendezvous), so a rung that fails or hangs -- on one rank or on all -- is killed and cannot poison the next one."""
    if os.environ.get("LIG_COMM") == "ipc":           # tests: W ranks share the one GPU of the box
        return ["ipc-stream", "ipc-sync", "host"]
    if world == 1:
        return ["rccl-stream", "rccl-sync", "host"]
    return ["rccl-stream", "rccl-sync", "torch", "ipc-stream"]


def sharded_child(a, lig_dist):
    """one rank of one rung of the sharded leg (spawned by sharded_ladder): builds the communicator of --transport, proves the
    sharded trace at every size of --sharded-log2, rank 0 prints one tagged JSON line per size as soon as it is measured"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sizes = [int(x) for x in str(a.sharded_log2).split(",") if x]
    if a.transport.startswith("stub"):                # launcher tests on CPU: a rendezvous, a barrier, canned outcomes
        group = lig_dist.Group("gloo")
        if a.transport == "stub-fail" and rank == world - 1:
            raise SystemExit("stub-fail: rank %d gives up" % rank)
        if a.transport == "stub-hang" and rank == world - 1:
            time.sleep(3600)
        group.barrier()
        for lg in sizes:
            if rank == 0:
                print(RESULT_TAG + json.dumps({"log2_constraints": lg, "ranks": world, "transport": a.transport, "ms_per_proof": 1.0}), flush=True)
        group.close()
        return
    import torch
    if os.environ.get("LIG_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    pkg = load_pkg()
    group = lig_dist.Group("nccl" if a.transport == "torch" and world > 1 else "gloo")      # rendezvous only, except for the torch rung
    ctx = pkg.Context(L_, K_, N_, device=local_rank)

    def fence():
        group.barrier()
        torch.cuda.synchronize()

    comm = group.make_comm(pkg, ctx, a.transport)
    for lg in sizes:
        out = sharded_leg(ctx, group, pkg, lg, a.sharded_steps, 2, fence, comm, a.transport)
        if rank == 0:
            print(RESULT_TAG + json.dumps(out), flush=True)
    group.close()
    ctx.close()


def sharded_ladder(a, group, rank, world, local_rank, workload="sharded"):
    """The configs[3] leg of `bench.py --gpus N`: rank by rank one CHILD process per rung (a fresh rendezvous on a port rank 0
    picks), the rungs of ladder_transports() in order until one delivers every size.  All ranks agree on the outcome of a rung
    through the parent's process group (MIN of the ranks' success flags), a child that does not finish within
    --sharded-timeout is killed.  Returns (results by log2 size, attempts)."""
    import subprocess
    names = [x for x in (a.sharded_transports or "").split(",") if x] or ladder_transports(world)
    sizes = [int(x) for x in str(a.sharded_log2).split(",") if x]
    t_begin = time.time()
    attempts, results = [], {}
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}     # (an agent store would swallow the child's own rendezvous)
    for name in names:
        # rank 0 decides whether there is time for another rung and picks the rendezvous port; everybody learns both
        go = 0
        if rank == 0 and time.time() - t_begin + 15 < a.sharded_budget:
            go = free_port()
        go = group.sum_over_ranks(go)
        if not go:
            attempts.append({"transport": name, "ok": False, "skipped": "the leg's time budget (--sharded-budget %d s) is used up" % a.sharded_budget})
            continue
        env = dict(env0, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(go),
                   LIG_BENCH_CHILD="1")
        cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", "--transport", name, "--gpus", str(world), "--workload", workload,
               "--sharded-log2", ",".join(map(str, sizes)), "--sharded-steps", str(a.sharded_steps)]
        t0 = time.time()
        timed_out = False
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        try:
            o, e = p.communicate(timeout=a.sharded_timeout)
        except subprocess.TimeoutExpired:
            timed_out = True
            p.kill()
            o, e = p.communicate()
        ok = (not timed_out) and p.returncode == 0
        got = {}
        for ln in o.decode(errors="replace").splitlines():
            if ln.startswith(RESULT_TAG):
                r = json.loads(ln[len(RESULT_TAG):])
                got[str(r["log2_constraints"])] = r
        if rank == 0:
            ok = ok and all(str(lg) in got for lg in sizes)
        n_ok = group.sum_over_ranks(1 if ok else 0)
        att = {"transport": name, "ok": n_ok == world, "ranks_ok": n_ok, "seconds": round(time.time() - t0, 1)}
        if n_ok != world:
            att["this_rank"] = "timeout after %d s (killed)" % a.sharded_timeout if timed_out else "rc %s" % p.returncode
            tail = [ln for ln in e.decode(errors="replace").splitlines() if ln.strip()][-6:]
            att["stderr_tail"] = " | ".join(tail)[-700:]
            if got:
                att["sizes_done_before_failure"] = sorted(got)
        attempts.append(att)
        for k_, v in got.items():                       # a size measured before a later one failed still counts
            results.setdefault(k_, v)
        if n_ok == world:
            break
    return results, attempts


def rocm_smi_topo():
    """link types between the GPUs as rocm-smi reports them (rank 0, best effort)"""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        txt = subprocess.run([exe, "--showtopotype"], capture_output=True, timeout=30).stdout.decode(errors="replace")
    except (OSError, subprocess.SubprocessError) as e:
        return {"error": repr(e)[:200]}
    kinds = re.findall(r"\b(XGMI|PCIE)\b", txt)
    return {"link_types": {k_: kinds.count(k_) for k_ in sorted(set(kinds))}, "raw_tail": " / ".join(ln.strip() for ln in txt.splitlines() if ln.strip().startswith("GPU"))[:1200]}


def preflight(group, pkg, torch, local_rank, world):
    """what the N > 1 run stands on, recorded in the line: which physical device every rank has, peer access, the RCCL the
    library resolves, the IPC mode the ranks run with, the node's link types"""
    bus = pkg.device_pci_bus_id(local_rank) or "?"
    ids = [b.rstrip(b"\0").decode(errors="replace") for b in group.gather_digests(bus.encode()[:32].ljust(32, b"\0"))]
    ndev = torch.cuda.device_count()
    peers = [pkg.device_peer_access(local_rank, j) for j in range(ndev) if j != local_rank]
    all_peers = group.sum_over_ranks(1 if all(peers) else 0) == world
    ok, lib, ver = pkg.rccl_available()
    out = {"pci_bus_ids": ids, "distinct_devices": len(set(ids)) == world, "visible_devices_rank0": ndev,
           "peer_access_between_all_visible_devices": all_peers,
           "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
           "rccl": {"available": ok, "library": lib, "version": ver}}
    if group.rank == 0:
        out["topology"] = rocm_smi_topo()
    return out


def usable_cores():
    """host threads this process can really run at once: the affinity mask, capped by the cgroup CPU quota (a GPU box may show
    256 logical CPUs to a container that is allowed a handful: 256 OpenMP threads then time-slice and the all-core figure
    collapses -- round 2's baseline had exactly that)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / float(g.read().strip()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(workload_name, budget_s=20.0):
    """the oracle (CPU restatement of the reference algorithm: radix-2 stages + bit reversal as in
    src/webgpu/engine.cpp:844-968, every row re-encoded in each of the three stages as in
    include/zkp/nonbatch_context.hpp) timed on this box's host cores on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    cores = usable_cores()
    o = ol.Ctx(L_, K_, N_)
    msgs = ol.rng_fill(synth_key(), 0, K_).reshape(1, K_, 8)
    t0 = time.perf_counter()
    o.encode_rows(msgs, threads=1)
    t_row = time.perf_counter() - t0
    if workload_name == "encode":
        rows = max(cores, min(4096, int(budget_s * cores / max(t_row, 1e-6))))
        rows -= rows % cores
        msgs = np.ascontiguousarray(np.broadcast_to(msgs, (rows, K_, 8)))
        t0 = time.perf_counter()
        o.encode_rows(msgs, threads=cores)
        dt = time.perf_counter() - t0
        return {"value": rows * L_ / dt, "unit": "constraints/s", "cores": cores, "kind": "port",
                "sample": "%d rows of k=8192 (INTT_k + NTT_4k, radix-2 stages as the reference), OpenMP over rows, %.1f s; "
                          "1-thread row time %.1f ms" % (rows, dt, 1e3 * t_row)}
    # full proof, reference structure (every row re-encoded in each of the three stages + one randomness-row encode + the
    # per-row hash / accumulator passes: ~10 row-encode equivalents per row).  Two runs: ONE thread on a small sample (the
    # per-core rate), then ALL host threads on a sample with >= 8 rows per thread (capped at the full 2^24 trace), so that every
    # thread has work in every parallel region.  Timed by the oracle's own stage timers: synthetic-input generation and table
    # set-up are outside, as for the GPU figure.
    def run(rows, threads):
        job = ol.make_job(L_, K_, N_, T_, rows * L_, 0, threads=threads)
        pr = ol.Proof()
        t0 = time.perf_counter()
        rc = ol.lib().lo_prove(C.byref(job), C.byref(pr))
        wall = time.perf_counter() - t0
        ok = rc == 0 and pr.valid_code and pr.valid_linear and pr.valid_quad
        stages = (pr.t_stage1, pr.t_stage2, pr.t_stage3)
        ol.lib().lo_proof_free(C.byref(pr))
        if not ok:
            raise SystemExit("CPU baseline prover failed its self-check")
        return rows * L_ / sum(stages), stages, wall
    import resource
    # per-proof constants (3 mask encodes per stage, 3 decodes, the Merkle tree, the seed hashes): an empty statement is exactly that
    _, st0_1, _ = run(0, 1)
    _, st0_a, _ = run(0, cores)
    rows1 = 64                            # >= 64 rows on ONE thread: the constants are < 5 % of the sample (8 rows made the per-core rate look 15 % low)
    v1_raw, st1, wall1 = run(rows1, 1)
    v1 = rows1 * L_ / max(sum(st1) - sum(st0_1), 1e-9)
    est_row_s = (sum(st1) - sum(st0_1)) / rows1
    rows = int(min(2098, max(64, 8 * cores, 0.5 * budget_s * cores / max(est_row_s, 1e-6))))      # >= 8 rows per thread, ~budget_s / 2 of wall time
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    va_raw, sta, walla = run(rows, cores)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(walla, 1e-9)
    va = rows * L_ / max(sum(sta) - sum(st0_a), 1e-9)          # marginal rate: rows per second beyond the per-proof constants
    # what a full 2^24-constraint proof would take at these rates: constants + 2098 rows at the marginal rate
    full = (1 << 24) / (sum(st0_a) + 2098 * (sum(sta) - sum(st0_a)) / rows)
    return {"value": full, "unit": "constraints/s", "cores": cores, "kind": "port", "logical_cpus": os.cpu_count(),
            "cpu_seconds_per_wall_second": busy,
            "value_allcores": full, "value_allcores_sample_raw": va_raw, "marginal_allcores": va, "marginal_1thread": v1, "value_1thread": v1,
            "value_1thread_sample_raw": v1_raw, "value_1thread_x_cores": v1 * cores, "parallel_efficiency": va / (v1 * cores),
            "per_proof_constants_s": {"1thread": sum(st0_1), "allcores": sum(st0_a)},
            "sample": "full 3-stage proof of %d rows (%d constraints) of k=8192 on %d threads: %.2f s in the stages (%.2f/%.2f/%.2f; %.2f s "
                      "wall incl. synthetic row forming), reference structure (every row re-encoded per stage, per-row hash / accumulator "
                      "passes), OpenMP over rows / column blocks; 1-thread run: %d rows in %.2f s; per-proof constants (an empty statement: masks, "
                      "decodes, Merkle) %.2f s on one thread / %.2f s on all, subtracted for the marginal rates and the parallel efficiency; "
                      "`value` = 2^24 constraints / (constants + 2098 rows at the all-thread marginal rate); 1-thread encode %.1f ms/row"
                      % (rows, rows * L_, cores, sum(sta), sta[0], sta[1], sta[2], walla, rows1, sum(st1), sum(st0_1), sum(st0_a), 1e3 * t_row)}


def gpus_sweep(a, argv):
    """`python bench.py --gpus-sweep 1,2,4,8`: the bench once per N (each run is its own `bench.py --gpus N`, ranks launched by the run itself),
    one compact JSON line per N as it finishes -- the four figures north_star asks for: `value` (independent traces: weak scaling),
    `sharded` (ONE 2^26 trace over the N ranks: strong scaling), the number of ranks the RCCL communicator had, and the dominant kernel's
    HBM fraction -- and a last line with all of them.  Scaling efficiency is for the reader (the driver) to compute.  On a box with fewer
    GPUs than N the sweep only runs when LIG_BENCH_SHARE_GPU=1 (all ranks on GPU 0, collectives through comm_ipc over a gloo rendezvous):
    every line then says `shared_device: true` -- a functional run of the N-rank code paths, NOT a scaling measurement."""
    import subprocess
    ns = [int(x) for x in a.gpus_sweep.split(",") if x]
    if not ns or min(ns) < 1:
        raise SystemExit("bench.py: --gpus-sweep takes a comma-separated list of GPU counts, e.g. 1,2,4,8")
    rest, skip = [], False
    for tok in argv:                      # this run's other flags go to every N
        if skip:
            skip = False
            continue
        if tok in ("--gpus-sweep", "--gpus"):
            skip = True
            continue
        if tok.startswith("--gpus-sweep=") or tok.startswith("--gpus="):
            continue
        rest.append(tok)
    try:
        import torch
        n_dev = torch.cuda.device_count()
    except Exception:                     # (the stub workload runs without torch devices)
        n_dev = 0
    share = os.environ.get("LIG_BENCH_SHARE_GPU") == "1"
    lines = []
    for n in ns:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
        args = ["--gpus", str(n)] + rest
        shared = a.workload != "stub" and n > 1 and n_dev < n
        if shared and not share:
            line = {"n_gpus": n, "skipped": "%d GPUs visible (set LIG_BENCH_SHARE_GPU=1 for a functional shared-device run)" % n_dev}
            lines.append(line)
            print(json.dumps(line), flush=True)
            continue
        if shared:
            env.update(LIG_COMM="ipc", LIG_COMM_TAG="sweep%d_%d" % (os.getpid(), n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            if "--backend" not in rest:
                args += ["--backend", "gloo"]
        elif n == 1:
            env.pop("LIG_BENCH_SHARE_GPU", None)
        t0 = time.perf_counter()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + args, env=env, capture_output=True, timeout=a.sweep_timeout)
            rc, so, se = p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
        except subprocess.TimeoutExpired as e:
            rc, so, se = -9, (e.stdout or b"").decode(errors="replace"), "timeout after %d s" % a.sweep_timeout
        full = None
        for ln in reversed(so.splitlines()):
            if ln.startswith("{"):
                try:
                    full = json.loads(ln)
                    break
                except ValueError:
                    pass
        line = {"n_gpus": n, "shared_device": bool(shared), "wall_s": round(time.perf_counter() - t0, 1)}
        if full is None:
            line["error"] = "rc %d: %s" % (rc, se[-400:])
        else:
            sh = full.get("sharded") or {}
            line.update({
                "metric": full.get("metric"), "unit": full.get("unit"), "value": full.get("value"), "scaling": full.get("scaling"), "ms_per_step": full.get("ms_per_step"),
                "steps": full.get("steps"), "workload": (full.get("config") or {}).get("workload"),
                "hbm_frac": (full.get("roofline") or {}).get("frac"),
                "sharded": None if not sh else {k: sh.get(k) for k in ("log2_constraints", "ranks", "ms_per_proof", "constraints_per_s", "scaling", "transport", "proof_equals_oracle_pin",
                                                                       "all_ranks_same_envelope", "rccl_ranks", "stage_ms", "error") if k in sh},
                "rccl_ranks": sh.get("rccl_ranks"), "distinct_devices": (sh.get("devices") or {}).get("distinct"),
            })
            for key, v in full.items():           # the other sizes of the sharded leg (sharded_2p24, ...)
                if key.startswith("sharded_2p") and isinstance(v, dict):
                    line[key] = {k: v.get(k) for k in ("ms_per_proof", "constraints_per_s", "transport", "proof_equals_oracle_pin", "all_ranks_same_envelope", "error") if k in v}
            if a.sweep_full:
                line["line"] = full
        lines.append(line)
        print(json.dumps(line), flush=True)
    print(json.dumps({"gpus_sweep": lines, "note": "per N: value = independent traces (weak), sharded = ONE trace over the ranks (strong); shared_device lines are functional "
                                                   "runs on one GPU and carry no scaling information"}), flush=True)
    return 0 if all("error" not in ln for ln in lines) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--gpus-sweep", default=None, help="comma-separated GPU counts (e.g. 1,2,4,8): run the bench once per N and print one compact line per N "
                    "(value = weak, sharded = strong, RCCL rank count, HBM fraction) plus a summary line; see gpus_sweep()")
    ap.add_argument("--sweep-timeout", type=int, default=1500, help="--gpus-sweep: seconds one N may take")
    ap.add_argument("--sweep-full", action="store_true", help="--gpus-sweep: embed every N's complete JSON line")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="full", choices=["full", "encode", "sharded", "stub"])
    ap.add_argument("--backend", default=None, help="process-group backend (default nccl = RCCL; the stub workload uses gloo)")
    ap.add_argument("--sharded-log2", default="24,26", help="N > 1: sizes of the ONE trace sharded over the ranks (configs[3] is 2^26; 2^24 is the "
                    "N = 1 bench size, so strong scaling is visible there too)")
    ap.add_argument("--sharded-steps", type=int, default=5)
    ap.add_argument("--sharded-timeout", type=int, default=180, help="seconds after which the child processes of one rung of the sharded leg are killed")
    ap.add_argument("--sharded-budget", type=int, default=480, help="seconds the whole sharded leg may take: no further rung is started beyond it")
    ap.add_argument("--sharded-transports", default=None, help="comma-separated rungs instead of the default ladder (ligero-prover_amd/dist.py: TRANSPORTS)")
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)       # one rank of one rung (spawned by the ladder)
    ap.add_argument("--transport", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--sharded-leg", action="store_true", help="run the configs[3] leg with one rank as well")
    ap.add_argument("--no-sharded-leg", action="store_true")
    ap.add_argument("--log2-constraints", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-probe", action="store_true", help="skip the one-proof-at-a-time latency probe before the timed region (profiler runs: "
                    "with --warmup 0 the process then launches the dominant kernel in the timed region only)")
    ap.add_argument("--no-verify", action="store_true", help="skip the (untimed, informational) HIP verifier run on the last proof")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive second measurement (value_incl_h2d)")
    ap.add_argument("--quad-percent", type=int, default=0, help="main measurement: this share of the constraints quadratic (default 0: configs[2] is all linear)")
    ap.add_argument("--quad-mix", type=int, default=50, help="N = 1: a second, shorter measurement with this share of quadratic constraints, "
                    "reported as `quad_mix` beside `value` (0: skip)")
    ap.add_argument("--no-h2d-rands", action="store_true", help="skip the leg that also ships the caller's randomness rows from host memory")
    ap.add_argument("--h2d-inflight", type=int, default=2, help="contexts alternating in the PCIe-inclusive measurement: each pipelines "
                    "upload i+1 under proof i, the uploads of all contexts go through one uploader thread (one at a time), so two "
                    "contexts keep the link busy: commit / prove of one under the upload of the other")
    ap.add_argument("--h2d-narrow", type=int, default=8, choices=[0, 4, 8], help="also time a small-witness trace shipped in the narrow row "
                    "format with this many bytes per witness slot (0: skip)")
    ap.add_argument("--inflight", type=int, default=2, help="full workload: proofs (traces) proved concurrently per GPU in one step")
    a = ap.parse_args()
    global NO_VERIFY
    NO_VERIFY = a.no_verify
    log2c = a.log2_constraints if a.log2_constraints is not None else (20 if a.workload == "encode" else 24)
    if a.gpus_sweep:
        sys.exit(gpus_sweep(a, sys.argv[1:]))
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))         # no launcher around us: be the launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- the line would report the wrong number of GPUs" % (a.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    spec = importlib.util.spec_from_file_location("lig_dist", os.path.join(ROOT, "ligero-prover_amd", "dist.py"))
    lig_dist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lig_dist)
    if a.sharded_child:
        return sharded_child(a, lig_dist)
    if a.workload == "stub":
        return run_stub(a, lig_dist)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("LIG_BENCH_SHARE_GPU") == "1":
        # test mode (tests/test_gpu_sharded.py): all ranks on GPU 0 of a one-GPU box, collectives through the process-to-process
        # communicator (LIG_COMM=ipc) over a gloo rendezvous -- RCCL cannot put two ranks on one device.  Not a benchmark.
        local_rank = 0
    if world > 1 and torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d of %d has no GPU of its own (%d visible)" % (local_rank, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    pkg = load_pkg()
    group = lig_dist.Group(a.backend or "nccl")          # RCCL over xGMI; a no-op object when WORLD_SIZE == 1
    dist = group.dist

    ctx = pkg.Context(L_, K_, N_, device=local_rank)
    if a.workload == "sharded":
        wl = ShardedWorkload(ctx, 1 << log2c, group, pkg)
    else:
        wl = (FullWorkload(ctx, 1 << log2c, pkg, max(1, a.inflight), local_rank, quad_percent=a.quad_percent) if a.workload == "full"
              else EncodeWorkload(ctx, 1 << log2c))

    def fence():
        group.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.step()
    single_ms, single_prof, single_prof512 = None, None, None
    if hasattr(wl, "single_proof_ms") and not a.no_latency_probe:      # latency probe: one proof at a time, dominant kernel bracketed as well
        ctx.profile_enable(True)
        single_ms = wl.single_proof_ms()
        single_prof = ctx.profile_read()
        single_prof512 = ctx.profile_read_launches(512)
        ctx.profile_enable(False)
    fence()
    # the dominant kernel is bracketed on EVERY context of the workload (two proofs in flight = two contexts): the figures below cover exactly the
    # launches of the timed region, which is what a rocprofv3 kernel table of `--warmup 0 --no-latency-probe` lists (tools/profile_round.sh)
    prof_ctxs = list(getattr(wl, "ctxs", [ctx]))
    for pc in prof_ctxs:
        pc.profile_enable(True)
    t0 = time.perf_counter()
    if hasattr(wl, "run"):
        wl.run(a.steps)
    else:
        for _ in range(a.steps):
            wl.step()
    fence()
    dt = time.perf_counter() - t0
    launches = prows = launches512 = 0
    kms = kms512 = 0.0
    for pc in prof_ctxs:
        a_, b_, c_ = pc.profile_read()
        d_, e_ = pc.profile_read_launches(512)
        launches += a_; prows += b_; kms += c_; launches512 += d_; kms512 += e_
        pc.profile_enable(False)
    dt = group.max_over_ranks(dt)

    # what the workload says about itself (incl. the untimed verifier run on the last proof), then its contexts go: every
    # context holds three HIP streams, and idle streams still occupy slots of the 4 hardware queues the next leg's streams map to
    wl_desc = wl.describe() if rank == 0 else None
    wl.close()

    # the quadratic mix (optional): the same number of constraints, half of them quadratic -- three committed rows
    # per 8000 quadratic slots instead of one, so constraints/s differs from the all-linear figure by up to 3x
    quad_mix = None
    if a.workload == "full" and a.quad_mix and not a.quad_percent and world == 1:
        try:
            qw = FullWorkload(ctx, 1 << log2c, pkg, max(1, a.inflight), local_rank, quad_percent=a.quad_mix)
            qw.run(2)
            fence()
            t0 = time.perf_counter()
            qw.run(a.steps)
            fence()
            dtq = time.perf_counter() - t0
            qd = qw.describe()
            quad_mix = {"value": qw.constraints * a.steps / dtq, "unit": "constraints/s", "ms_per_step": 1e3 * dtq / a.steps, "quad_percent": a.quad_mix,
                        "n_linear": qw.n_lin, "n_quad": qw.n_quad, "rows": qd["rows"], "proofs_in_flight": qw.inflight,
                        "rows_per_s": (qw.rows + 3) * qw.inflight * a.steps / dtq,
                        "proof_sha256": qd.get("proof_sha256"), "proof_equals_oracle_pin": qd.get("proof_equals_oracle_pin"),
                        "verifier_accepts": qd.get("verifier_accepts"), "stage_ms": qd.get("stage_ms")}
            qw.close()
        except (RuntimeError, MemoryError, pkg.LigError) as e:
            sys.stderr.write("quadratic-mix leg skipped: %r\n" % (e,))

    # PCIe-inclusive figure: the same proofs with the witness matrix starting in pinned host memory (caller-rows entry)
    incl = None
    if a.workload == "full" and not a.no_h2d and world == 1:      # informational, N = 1 only (like cpu_baseline): keeps the scaling runs lean
        try:
            hw = RowsFromHostWorkload(ctx, 1 << log2c, pkg, max(1, a.h2d_inflight), local_rank)
            hw.run(3)                                   # warm-up: the second message matrix is allocated by the first prefetch, pages are touched
            fence()
            # the link itself: the same pinned matrix copied to the device by a plain stream copy (best of 3)
            link_ms = None
            try:
                dev_buf = torch.empty_like(hw.host, device="cuda")
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); dev_buf.copy_(hw.host, non_blocking=True); e1.record(); e1.synchronize()
                    link_ms = e0.elapsed_time(e1) if link_ms is None else min(link_ms, e0.elapsed_time(e1))
                del dev_buf
            except RuntimeError:
                pass
            fence()
            t0 = time.perf_counter()
            hw.run(a.steps)
            fence()
            dth = group.max_over_ranks(time.perf_counter() - t0)
            incl = {"value": hw.constraints * a.steps * world / dth, "ms_per_step": 1e3 * dth / a.steps,
                    "proof_sha256": hw.proof_sha256(),
                    "witness_bytes_per_trace": int(hw.host.numel() * 4),
                    "link_ms_per_trace": link_ms, "link_GBps": None if not link_ms else hw.host.numel() * 4 / link_ms / 1e6,
                    "frac_of_link_bound": None if not link_ms else (link_ms * 1e-3) / (dth / (a.steps * hw.inflight)),
                    "how": "lig_rows_restart/commit/prove: witness rows uploaded from pinned host memory inside the timed region "
                           "(chunk by chunk by the library's uploader thread, each chunk encoded as it arrives; upload i+1 under proof i), "
                           "%d contexts alternating" % hw.inflight}
            if not a.no_h2d_rands:
                # the real driver's figure: witness rows AND the caller's dense randomness rows from pinned host memory
                try:
                    hw.close()
                    hw = RowsFromHostWorkload(ctx, 1 << log2c, pkg, max(1, a.h2d_inflight), local_rank, caller_rands=True)
                    hw.run(3)
                    fence()
                    t0 = time.perf_counter()
                    hw.run(a.steps)
                    fence()
                    dtr = group.max_over_ranks(time.perf_counter() - t0)
                    both = int(hw.host.numel() * 4 + hw.rands.numel() * 4)
                    bound_ms = None if not incl["link_GBps"] else both / incl["link_GBps"] / 1e6
                    incl["caller_rands"] = {
                        "value": hw.constraints * a.steps * world / dtr, "ms_per_step": 1e3 * dtr / a.steps, "ms_per_proof": 1e3 * dtr / (a.steps * hw.inflight),
                        "h2d_bytes_per_trace": both, "link_bound_ms_per_trace": bound_ms,
                        "frac_of_link_bound": None if not bound_ms else bound_ms / (1e3 * dtr / (a.steps * hw.inflight)),
                        "proof_sha256": hw.proof_sha256(),
                        "how": "as above, and lig_rows_prove takes the caller's dense randomness rows (one 256-bit coefficient per witness, what "
                               "nonbatch_context.hpp:654-780 ships per linear row) from pinned host memory: the uploader thread fills the double "
                               "buffer chunk by chunk, the proof's stream waits on words in pinned memory (hipStreamWaitValue32)"}
                except (RuntimeError, MemoryError, pkg.LigError) as e:
                    sys.stderr.write("caller-rands H2D leg skipped: %r\n" % (e,))
            if a.h2d_narrow:
                try:
                    hw.close()
                    hw = RowsFromHostWorkload(ctx, 1 << log2c, pkg, max(1, a.h2d_inflight), local_rank, narrow_bytes=a.h2d_narrow)
                    hw.run(3)
                    fence()
                    t0 = time.perf_counter()
                    hw.run(a.steps)
                    fence()
                    dtn = group.max_over_ranks(time.perf_counter() - t0)
                    incl["narrow_format"] = {
                        "value": hw.constraints * a.steps * world / dtn, "ms_per_step": 1e3 * dtn / a.steps, "elem_bytes": a.h2d_narrow,
                        "witness_bytes_per_trace": int(hw.host.numel() * 4),
                        "note": "a small-witness trace (every witness < 2^%d) shipped in the narrow row format (lig_rows_job.elem_bytes): 4x less on "
                                "the link; another trace than the dense one, hence another proof" % (8 * a.h2d_narrow)}
                except (RuntimeError, MemoryError, pkg.LigError) as e:
                    sys.stderr.write("narrow H2D leg skipped: %r\n" % (e,))
            hw.close()
        except (RuntimeError, MemoryError, pkg.LigError) as e:      # e.g. no pinned memory left: the resident figure below stands on its own
            incl = None
            sys.stderr.write("value_incl_h2d skipped: %r\n" % (e,))
            if world > 1:
                raise                                    # ranks must not diverge around the collectives above

    if rank == 0:
        sharded = a.workload == "sharded"
        total_constraints = wl.constraints * a.steps * (1 if sharded else world)
        # dominant kernel = k_encode_tiles (K2: all tile transforms of a row, 90 % of the encode's multiplies): per encoded row
        # it reads k elements and writes the three computed cosets: (k + 3k) * 32 B = 1,048,576 B.  (SURVEY.md 8(d) quotes
        # 1,310,720 B for a whole row encode, read k*32 + write n*32; the fourth coset is the message row itself, so the
        # conservative figure for THIS kernel is its own compulsory traffic.)
        alg_bytes_per_row = (K_ + 3 * K_) * 32
        avg_launch_s = (kms / max(launches, 1)) * 1e-3
        rows_per_launch = prows / max(launches, 1)
        achieved = rows_per_launch * alg_bytes_per_row / max(avg_launch_s, 1e-12) / 1e9
        # HBM traffic of that kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs of this same command, FETCH_SIZE doubled per the gfx950 note of the microarch guide), per launch
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)["k_encode_tiles<10, true>"]
            traffic = pmc["hbm_bytes_per_row"] * rows_per_launch
            traffic_src = "profiles/pmc_traffic.json (%.0f B/row measured on %d-row launches)" % (pmc["hbm_bytes_per_row"], pmc["rows_in_launch"])
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "prover constraints/sec", "value": total_constraints / dt, "unit": "constraints/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u32 limbs (BN254 Fr: 256-bit modular integers, 9x29-bit limbs in registers; SHA-256 words)",
            "data": "synthetic",
            "config": dict(wl_desc, parallelism=("1 trace sharded over %d GPUs: all-to-all of codeword column slices + all-gathers" % world)
                           if sharded else "1 trace per GPU (independent traces, no collective)"),
            "proof_wall_ms": (single_ms if single_ms is not None else 1e3 * dt / a.steps) if a.workload != "encode" else None,
            "roofline": {"bound": "hbm", "kernel": "k_encode_tiles", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": rows_per_launch * alg_bytes_per_row,
                         "avg_launch_ms": 1e3 * avg_launch_s, "rows_per_launch": rows_per_launch, "launches": launches,
                         "algorithmic_bytes_per_row": alg_bytes_per_row,
                         # the full chunks alone (512 rows per launch): the figure a rocprofv3 kernel table lists for that launch size, so that
                         # profiles/r0N_inflight*_kernel_stats.md can be set against this line without averaging over the six launch sizes of a proof
                         "launches_of_512_rows": None if not launches512 else {
                             "launches": launches512, "avg_launch_ms": kms512 / launches512,
                             "achieved": 512 * alg_bytes_per_row / (kms512 / launches512 * 1e-3) / 1e9,
                             "frac": 512 * alg_bytes_per_row / (kms512 / launches512 * 1e-3) / 1e9 / 8000.0},
                         "encode_row_bytes_survey_8d": (K_ + N_) * 32,
                         # the same launches priced with SURVEY 8(d)'s whole-row encode figure (read k*32 + write n*32): this kernel
                         # carries ~90 % of the encode's arithmetic, the two radix-8 passes around it move the remaining bytes
                         "frac_with_survey_8d_row_bytes": rows_per_launch * (K_ + N_) * 32 / max(avg_launch_s, 1e-12) / 1e9 / 8000.0,
                         "one_proof_in_flight": None if not single_prof or not single_prof[0] else {
                             "avg_launch_ms": single_prof[2] / single_prof[0], "rows_per_launch": single_prof[1] / single_prof[0],
                             "achieved": (single_prof[1] * alg_bytes_per_row) / (single_prof[2] * 1e-3) / 1e9,
                             "frac": (single_prof[1] * alg_bytes_per_row) / (single_prof[2] * 1e-3) / 1e9 / 8000.0,
                             "avg_launch_ms_512_rows": None if not single_prof512 or not single_prof512[0] else single_prof512[1] / single_prof512[0]},
                         "note": "integer-VALU-bound kernel (~185k 256-bit Montgomery products per row in this kernel: since round 2 it "
                                 "also carries the inverse tile transforms that used to be a kernel of their own, same bytes, more "
                                 "arithmetic; v_mad_u64_u32 issues at half the simple-ALU rate); the HBM fraction is small by "
                                 "construction, see DESIGN.md"},
        }
        # the roof this kernel actually runs against: issue slots of v_mad_u64_u32.  Work per encoded row: 8 tiles x 1024 elements, an
        # inverse + three forward tile transforms with their twists and seams = ~185k windowed products of 117 v_mad_u64_u32 + 4
        # v_mul_lo_u32 (DESIGN.md sections 2 and 4; the kernel's ISA: 10,796 multiplies per wave x 32 waves per row); peak: the MEASURED
        # chip-wide rate of independent v_mad_u64_u32 with 4 waves per SIMD (profiles/r01_ubench_valu_issue_rates.txt: 32.4e12 lane-multiplies/s
        # = one wave64 multiply per SIMD per ~4.6 cycles at the 2.3 GHz the chip holds).  Informational: the contract's roofline object above is the HBM one.
        mads_per_row = 185000.0 * 121
        peak_mads = 32.4e12
        if launches:
            out["roofline"]["valu_multiplier"] = {"achieved_mads_per_s": rows_per_launch * mads_per_row / max(avg_launch_s, 1e-12), "peak_mads_per_s": peak_mads,
                                                  "frac": rows_per_launch * mads_per_row / max(avg_launch_s, 1e-12) / peak_mads,
                                                  "how": "185k products/row x 121 multiplies / in-run average launch time of k_encode_tiles, against the measured chip-wide "
                                                         "v_mad_u64_u32 rate (32.4e12/s, profiles/r01_ubench_valu_issue_rates.txt); the other 46 % of the kernel's instructions "
                                                         "(limb adds, masks, shifts, LDS exchanges: a third of its issue time) share the same issue slots"}
        if a.workload != "encode":
            # whole-proof algorithmic bytes (4*k*32 + 192*32 = 1,054,720 B per committed row) over the wall time
            e2e = 1054720.0 * (wl.rows + 3) * (total_constraints / wl.constraints_per_trace if hasattr(wl, "constraints_per_trace") else a.steps) / dt / 1e9
            out["roofline"]["hbm_frac_end_to_end"] = e2e / 8000.0
            out["roofline"]["end_to_end_GBps"] = e2e
        def newest(pattern):
            """the newest committed profile of that kind (profiles/r0N_<pattern>), None if there is none"""
            import glob
            hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
            return hits[-1] if hits else None
        budget_path = newest("valu_budget.json")
        try:
            with open(budget_path) as f:
                out["roofline"]["valu_busy_pct"] = {kname: v.get("valu_busy_pct") for kname, v in json.load(f).get("kernels", {}).items()
                                                    if v.get("valu_busy_pct", 0) >= 20}
                out["roofline"]["valu_busy_source"] = ("profiles/%s (rocprofv3 --pmc VALUBusy --kernel-trace over full proofs, one in flight, dispatches serialised; "
                                                       "a committed measurement, not taken in this run)" % os.path.basename(budget_path))
        except (OSError, ValueError, TypeError):
            pass
        if a.workload == "full":
            # chip-wide: the VALU issue time ONE proof needs (sum over all its launches of stand-alone duration x VALUBusy, a committed PMC
            # pass of this build: tools/valu_budget.sh) against the time the bench takes per proof -- the fraction of issue slots in use
            try:
                with open(budget_path) as f:
                    busy_ms = json.load(f)["valu_busy_ms_per_proof"]
                ms_per_proof = 1e3 * dt / a.steps / max(1, getattr(wl, "inflight", 1))
                out["roofline"]["valu_issue"] = {"busy_ms_per_proof": busy_ms, "measured_ms_per_proof": ms_per_proof, "frac": busy_ms / ms_per_proof,
                                                 "source": "profiles/%s (rocprofv3 --pmc VALUBusy --kernel-trace over full proofs; a committed measurement, not taken in "
                                                           "this run), see profiles/r03_valu_budget.md (method), profiles/r05_issue_timeline.md (where the idle slots are) "
                                                           "and DESIGN.md section 4" % os.path.basename(budget_path)}
            except (OSError, ValueError, KeyError, TypeError):
                pass
        out["value_definition"] = ("witness matrix resident in HBM when the clock starts (the measurement contract this build is judged by reserves `value` for "
                                   "inputs resident in HBM; a PCIe-inclusive rate is reported beside it, never as `value`); SURVEY 8(d)'s H2D-inclusive figure is "
                                   "value_incl_h2d, and with the caller's randomness rows from host memory as well incl_h2d.caller_rands, measured in the same run")
        if quad_mix is not None:
            out["quad_mix"] = quad_mix
        if incl is not None:
            out["value_incl_h2d"] = incl["value"]
            out["incl_h2d"] = incl
            out["incl_h2d"]["same_proof_bytes"] = incl["proof_sha256"] == out["config"].get("proof_sha256")
            if "caller_rands" in incl:
                incl["caller_rands"]["same_proof_bytes"] = incl["caller_rands"]["proof_sha256"] == out["config"].get("proof_sha256")
    else:
        out = None
    if a.workload == "full" and not a.no_sharded_leg and (world > 1 or a.sharded_leg):
        # The sharded leg is the one part of this script that needs real peers (RCCL over xGMI).  It runs in child processes, rung
        # by rung (sharded_ladder): whatever happens there -- an error on one rank, a hung collective, a crash -- the weak-scaling
        # figure measured above is already in `out` and is printed.
        ctx.close()                                         # the children use the GPU now (close() again below is a no-op)
        import threading
        done = threading.Event()

        def watchdog():           # last resort: the PARENTS' own collectives hang (a peer parent died): print what there is and leave
            if done.wait(a.sharded_budget + a.sharded_timeout + 60):
                return
            if out is not None:
                out.setdefault("sharded", {"error": "the sharded leg's launcher did not return (a rank of the bench itself is gone?); the weak-scaling figures above are unaffected"})
                sys.stdout.flush()
                print(json.dumps(out), flush=True)
            os._exit(0 if out is not None else 1)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            pre = preflight(group, pkg, torch, local_rank, world)
        except (RuntimeError, OSError, pkg.LigError) as e:
            pre = {"error": repr(e)[:300]}
        results, attempts = sharded_ladder(a, group, rank, world, local_rank)
        done.set()
        if out is not None:
            sizes = [int(x) for x in str(a.sharded_log2).split(",") if x]
            main_size = 26 if 26 in sizes else sizes[-1]
            sh = dict(results.get(str(main_size)) or {"error": "no rung of the ladder delivered the 2^%d trace" % main_size})
            sh["attempts"] = attempts
            sh["answers"] = ("strong scaling of ONE trace (north_star's '>= 6x further at 8 GPUs' read as latency of one proof); the per-column hash chain does "
                             "not shorten with W, see DESIGN.md section 7 for the per-W prediction; `value` above (independent traces) is the throughput answer")
            # so that a run on a real node cannot be mistaken for this box's stand-ins: how many DEVICES the ranks had
            sh["devices"] = {"distinct": bool(pre.get("distinct_devices")), "pci_bus_ids": pre.get("pci_bus_ids"),
                             "note": "ranks on distinct GPUs" if pre.get("distinct_devices") else "ALL ranks shared one GPU (test stand-in: no xGMI link carried anything)"}
            out["answers"] = {"value": "throughput: independent traces, one per GPU, no collective (weak scaling)", "sharded": sh["answers"]}
            out["sharded"] = sh
            for lg in sizes:
                if lg != main_size:
                    out["sharded_2p%d" % lg] = results.get(str(lg)) or {"error": "not delivered"}
            out["preflight"] = pre
    if out is not None:
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl.name)
        result = json.dumps(out)
    else:
        result = None
    group.close()               # the RCCL communicator lives on the context's streams: it goes first
    ctx.close()
    try:                            # RCCL prints its banner through C stdio, which is block-buffered when stdout is a pipe:
        C.CDLL(None).fflush(None)   # push it out now so that the JSON line below really is the last line
    except OSError:
        pass
    if result is not None:          # printed last and flushed: RCCL / the runtime may print their own lines earlier
        sys.stdout.flush()
        print(result, flush=True)


if __name__ == "__main__":
    main()
