#!/usr/bin/env python3
"""bench.py -- prover constraints/s of the MI355X-native Ligero hot path on synthetic BN254 traces.

  python bench.py --gpus N --steps K --warmup W [--workload encode|full] [--log2-constraints C]

One "step" = one pass of the hot path over one synthetic trace whose witness rows are already resident in
HBM (generated on the GPU with the reference's AES-256-CTR field sampler, key SHA256("lig-synth"||le64(1))):
  encode : configs[1] of BASELINE.json -- 2^20 constraints = 132 rows of l=8000, RS-encode only (INTT_k + NTT_4k)
  full   : configs[2] -- 2^24 constraints = 2098 rows (+3 masks): encode + column SHA-256 + Merkle root +
           stage-2 RLC accumulators (with dense randomness rows sampled and encoded on the GPU) + column gather
N > 1: one process per GPU (torch.distributed over RCCL, launched by torch.distributed.run); traces are
independent objects, so every rank proves its own trace (weak scaling, no data-path collective); the timed
region is bracketed by barrier + torch.cuda.synchronize() and the MAX over ranks is reported.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
L_, K_, N_, T_ = 8000, 8192, 32768, 192


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
        # rows = [l witnesses | k-l pad randoms]; for throughput purposes every slot is a stream sample
        ctx.rng_fill(synth_key(), 0, self.msgs, self.rows * K_)
        ctx.sync()

    def step(self):
        self.ctx.encode_rows(self.msgs, self.cws, self.rows)

    def describe(self):
        return {"workload": "configs[1]: 2^%d-constraint synthetic BN254 witness, RS-encode only (INTT_k + NTT_4k)"
                            % (self.constraints.bit_length() - 1),
                "rows": self.rows, "l": L_, "k": K_, "n": N_}


def cpu_baseline(workload_name, budget_s=15.0):
    """the oracle (CPU restatement of the reference algorithm, radix-2 stages + bit reversal as in
    src/webgpu/engine.cpp:844-968) timed on this box's host cores on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    cores = os.cpu_count() or 1
    o = ol.Ctx(L_, K_, N_)
    msgs = ol.rng_fill(synth_key(), 0, K_).reshape(1, K_, 8)
    t0 = time.perf_counter()
    o.encode_rows(msgs, threads=1)
    t_row = time.perf_counter() - t0
    rows = max(cores, min(4096, int(budget_s * cores / max(t_row, 1e-6))))
    rows -= rows % cores
    msgs = np.ascontiguousarray(np.broadcast_to(msgs, (rows, K_, 8)))
    t0 = time.perf_counter()
    o.encode_rows(msgs, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": rows * L_ / dt, "unit": "constraints/s", "cores": cores, "kind": "port",
            "sample": "%d rows of k=8192 (INTT_k + NTT_4k, radix-2 stages as the reference), OpenMP over rows, %.1f s; "
                      "1-thread row time %.1f ms" % (rows, dt, 1e3 * t_row)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="encode", choices=["encode"])
    ap.add_argument("--log2-constraints", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = load_pkg()
    ctx = pkg.Context(L_, K_, N_, device=local_rank)
    wl = EncodeWorkload(ctx, 1 << a.log2_constraints)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.step()
    fence()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.step()
    fence()
    dt = time.perf_counter() - t0
    launches, prows, kms = ctx.profile_read()
    ctx.profile_enable(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_constraints = wl.constraints * a.steps * world
        # dominant kernel = encode_mid (K2): per row it must read the k seam-twiddled inputs and write the n coset
        # values: (k + n) * 32 B = 1,310,720 B (SURVEY.md 8d encode-only figure)
        alg_bytes_per_row = (K_ + N_) * 32
        avg_launch_s = (kms / max(launches, 1)) * 1e-3
        rows_per_launch = prows / max(launches, 1)
        achieved = rows_per_launch * alg_bytes_per_row / max(avg_launch_s, 1e-12) / 1e9
        out = {
            "metric": "prover constraints/sec", "value": total_constraints / dt, "unit": "constraints/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (BN254 Fr, 256-bit modular integer)",
            "data": "synthetic", "config": dict(wl.describe(), parallelism="1 trace per GPU (independent, no collective)"),
            "roofline": {"bound": "hbm", "kernel": "k_encode_mid", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "avg_launch_ms": 1e3 * avg_launch_s, "rows_per_launch": rows_per_launch,
                         "note": "integer-VALU-bound kernel (~270k 256-bit Montgomery products per row); see DESIGN.md"},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl.name)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
