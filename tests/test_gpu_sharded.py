"""Single trace sharded over W ranks (BASELINE.json configs[3]) -- run as W processes that share the one GPU of the
test box, with gloo carrying the collectives through the host-synchronous callbacks of lig_comm; the product path (the
library's own RCCL communicator, csrc/comm_rccl.hip, stream-ordered) is exercised on a 1-rank communicator with the
exchange forced on.  Every rank's envelope must be byte-identical to the unsharded prover's (which the parity tests tie
to the oracle)."""
import hashlib
import json
import os
import subprocess
import sys
import textwrap

import pytest

import multirank as mr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import hashlib, importlib.util, json, os, sys
    root, l, k, n, n_lin, n_quad = sys.argv[1], *map(int, sys.argv[2:7])
    batch = len(sys.argv) > 7 and sys.argv[7] == "1"
    sys.path.insert(0, os.path.join(root, "tests"))
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "ligero-prover_amd", rel))
        m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
    pkg = load("ligero_prover_amd", "__init__.py")
    dist = load("lig_dist", "dist.py")
    g = dist.Group("gloo")
    ctx = pkg.Context(l, k, n, device=0)
    job = pkg.Context.make_job(n_lin, n_quad, generated_at=77)
    if batch:                            # a batch program ahead of the synthetic stream (tests/test_batch_rows.py)
        import test_batch_rows
        test_batch_rows.demo_program().attach(job)
    comm = g.make_comm(pkg, ctx)
    sh = ctx.shard_prepare(job, g.rank, g.world, comm)
    proof, info = ctx.shard_prove(sh)
    proof2, _ = ctx.shard_prove(sh)
    ctx.shard_destroy(sh)
    ref = None
    if g.rank == 0:                      # the unsharded prover on the same job
        tr = ctx.synth_prepare_job(job)
        ref, rinfo = ctx.synth_prove(tr)
        ctx.trace_destroy(tr)
    digs = g.gather_digests(hashlib.sha256(proof).digest())
    print(json.dumps({"rank": g.rank, "len": len(proof), "sha": hashlib.sha256(proof).hexdigest(), "again": proof == proof2,
                      "valid": [info.valid_code, info.valid_linear, info.valid_quad], "rows": info.rows,
                      "ref_sha": hashlib.sha256(ref).hexdigest() if ref is not None else None,
                      "all_equal": len(set(digs)) == 1}))
    ctx.close()
    g.close()
''')


def run_world(tmp_path, world, l, k, n, n_lin, n_quad, batch=False, comm=None, timeout=300):
    """comm=None: host-synchronous gloo callbacks; comm="ipc": the stream-ordered process-to-process communicator
    (csrc/comm_ipc.hip) -- the double-buffered exchange pipeline of lig_shard_prove with real peers on the one GPU.
    Every invocation has its own rendezvous port and communicator tag; all ranks are watched (tests/multirank.py)."""
    script = tmp_path / "shard_worker.py"
    script.write_text(WORKER)
    outs = mr.run_ranks(mr.python_argv(script, ROOT, l, k, n, n_lin, n_quad, "1" if batch else "0"), world, mr.rendezvous_env(world, comm), timeout=timeout)
    return sorted((mr.last_json(o) for o, _ in outs), key=lambda d: d["rank"])


@pytest.mark.parametrize("world,l,k,n,n_lin,n_quad", [
    (2, 320, 512, 2048, 2000, 900),       # 7 + 9 rows: linear block, quadratic triples, partial rows
    (4, 320, 512, 2048, 700, 0),          # 3 rows on 4 ranks: one rank owns no row
    (2, 8000, 8192, 32768, 5 * 8000 + 17, 8000),
    (2, 320, 512, 2048, 320 * 1500 + 7, 330),     # 1501 + 6 rows on 2 ranks: two exchange rounds (double-buffered send / receive)
    (4, 320, 512, 2048, 320 * 4300 + 1, 0),       # three rounds on 4 ranks, the last chunks shorter
])
def test_sharded_proof_equals_single_gpu_proof(tmp_path, world, l, k, n, n_lin, n_quad):
    outs = run_world(tmp_path, world, l, k, n, n_lin, n_quad)
    assert all(o["valid"] == [1, 1, 1] and o["again"] and o["all_equal"] for o in outs)
    assert len({o["sha"] for o in outs}) == 1
    assert outs[0]["ref_sha"] == outs[0]["sha"], "sharded envelope differs from the single-GPU envelope"


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_proof_with_batch_rows_equals_single_gpu_proof(tmp_path, world):
    """the batch program's rows are dealt to the ranks like any other rows (every rank runs the small program and keeps
    its slice; equality pairs and product triples are never split across ranks)"""
    outs = run_world(tmp_path, world, 320, 512, 2048, 900, 330, batch=True)
    assert all(o["valid"] == [1, 1, 1] and o["again"] and o["all_equal"] for o in outs)
    assert outs[0]["ref_sha"] == outs[0]["sha"], "sharded envelope differs from the single-GPU envelope"


@pytest.mark.parametrize("world,l,k,n,n_lin,n_quad", [
    (2, 320, 512, 2048, 2000, 900),               # one round
    (4, 320, 512, 2048, 700, 0),                  # one round, a rank without rows
    (2, 320, 512, 2048, 320 * 1500 + 7, 330),     # two rounds: both halves of the send / receive buffers
    (4, 320, 512, 2048, 320 * 4300 + 1, 0),       # three rounds on 4 ranks: buffer REUSE gated by ev_comm / ev_hash with real peers
    (2, 320, 512, 2048, 320 * 5200 + 3, 960),     # six rounds on 2 ranks
    (2, 8000, 8192, 32768, 5 * 8000 + 17, 8000),
    (8, 320, 512, 2048, 320 * 9000 + 11, 330),   # the node's shape: 8 ranks (8 processes on the one GPU), three rounds, 256 columns per rank
])
def test_sharded_over_stream_ordered_ipc_comm_equals_single_gpu_proof(tmp_path, world, l, k, n, n_lin, n_quad):
    """W processes on the one GPU, collectives = comm_ipc.hip: peers pull from each other's send buffers, ordering on the
    GPU (stream memory operations), so lig_shard_prove takes its `ordered` branch (exchange of round c on the copy stream
    under the encode of round c+1 and the hash of round c-1) with W > 1"""
    outs = run_world(tmp_path, world, l, k, n, n_lin, n_quad, comm="ipc")
    assert all(o["valid"] == [1, 1, 1] and o["again"] and o["all_equal"] for o in outs)
    assert len({o["sha"] for o in outs}) == 1
    assert outs[0]["ref_sha"] == outs[0]["sha"], "sharded envelope differs from the single-GPU envelope"


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_batch_rows_over_ipc_comm(tmp_path, world):
    outs = run_world(tmp_path, world, 320, 512, 2048, 900, 330, batch=True, comm="ipc")
    assert all(o["valid"] == [1, 1, 1] and o["again"] and o["all_equal"] for o in outs)
    assert outs[0]["ref_sha"] == outs[0]["sha"], "sharded envelope differs from the single-GPU envelope"


NCCL_WORKER = textwrap.dedent('''
    import hashlib, importlib.util, json, os, sys
    root = sys.argv[1]
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "ligero-prover_amd", rel))
        m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
    import torch
    torch.cuda.set_device(0)
    pkg = load("ligero_prover_amd", "__init__.py")
    dist = load("lig_dist", "dist.py")
    g = dist.Group("nccl", force_init=True)          # torch carries the unique id; the collectives are the library's own RCCL calls
    ctx = pkg.Context(320, 512, 2048, device=0)
    nl, nq = 320 * 1300 + 5, 900                     # 1301 + 9 rows: three exchange rounds
    job = pkg.Context.make_job(nl, nq, generated_at=77)
    comm = g.make_comm(pkg, ctx)
    assert bool(comm.all_to_all_on) and bool(comm.all_gather_on)
    sh = ctx.shard_prepare(job, g.rank, g.world, comm)
    proof, info = ctx.shard_prove(sh)
    proof2, _ = ctx.shard_prove(sh)
    ctx.shard_destroy(sh)
    tr = ctx.synth_prepare(nl, nq, generated_at=77)
    ref, _ = ctx.synth_prove(tr)
    ctx.trace_destroy(tr)
    print(json.dumps({"equal": proof == ref and proof2 == ref, "valid": [info.valid_code, info.valid_linear, info.valid_quad]}))
    g.close(); ctx.close()
''')


def test_rccl_stream_ordered_collectives_one_rank(tmp_path):
    """the product's RCCL communicator (grouped ncclSend/ncclRecv + ncclAllGather on the context's streams, exchange of
    round c under the encode of round c+1) on a 1-rank communicator with the exchange forced on -- all that a 1-GPU box
    can exercise of the RCCL path"""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(mr.free_port()), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               LIG_SHARD_FORCE_EXCHANGE="1")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])     # RCCL prints its own lines
    assert out == {"equal": True, "valid": [1, 1, 1]}


BIG_WORKER = textwrap.dedent('''
    import hashlib, importlib.util, json, os, sys
    root, lg = sys.argv[1], int(sys.argv[2])
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "ligero-prover_amd", rel))
        m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
    import torch
    torch.cuda.set_device(0)
    pkg = load("ligero_prover_amd", "__init__.py")
    dist = load("lig_dist", "dist.py")
    g = dist.Group(sys.argv[3] if len(sys.argv) > 3 else "nccl", force_init=True)
    ctx = pkg.Context(8000, 8192, 32768, device=0)
    job = pkg.Context.make_job(1 << lg, 0, synth_seed=1, generated_at=0)
    sh = ctx.shard_prepare(job, g.rank, g.world, g.make_comm(pkg, ctx))
    proof, info = ctx.shard_prove(sh)
    ctx.shard_destroy(sh)
    print(json.dumps({"sha": hashlib.sha256(proof).hexdigest(), "root": bytes(info.root).hex(), "rows": info.rows,
                      "valid": [info.valid_code, info.valid_linear, info.valid_quad]}))
    g.close(); ctx.close()
''')


def test_sharded_configs3_trace_full_size_equals_oracle_pin(tmp_path):
    """configs[3]'s 2^26-constraint trace through the sharded prover with the exchange pipeline ON (1-rank RCCL
    communicator, LIG_SHARD_FORCE_EXCHANGE: 17 rounds of pack -> grouped send/recv -> column hash, double-buffered) at
    full size: envelope SHA-256 and root equal the oracle's pin (tests/golden/full_pin_2p26.json)"""
    with open(os.path.join(ROOT, "tests", "golden", "full_pin_2p26.json")) as f:
        pin = json.load(f)
    script = tmp_path / "big_worker.py"
    script.write_text(BIG_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(mr.free_port()), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               LIG_SHARD_FORCE_EXCHANGE="1")
    p = subprocess.run([sys.executable, str(script), ROOT, "26"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["valid"] == [1, 1, 1] and out["rows"] == pin["rows"]
    assert out["root"] == pin["root"] and out["sha"] == pin["proof_sha256"]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_configs3_trace_full_size_over_ipc_comm_equals_oracle_pin(tmp_path, world):
    """configs[3]'s 2^26-constraint trace sharded over W REAL ranks (processes on the one GPU, comm_ipc.hip): 9 / 5 / 3 exchange
    rounds with the stream-ordered double-buffered pipeline; every rank's envelope equals the oracle's pin.  W = 8 is the node's
    shape (round 6): 4096 columns per rank, seven peers per exchange, 8-way all-gathers."""
    with open(os.path.join(ROOT, "tests", "golden", "full_pin_2p26.json")) as f:
        pin = json.load(f)
    script = tmp_path / "big_worker.py"
    script.write_text(BIG_WORKER)
    outs = mr.run_ranks(mr.python_argv(script, ROOT, "26", "gloo"), world, mr.rendezvous_env(world, "ipc"), timeout=420 if world < 8 else 900)
    for o, _ in outs:
        out = mr.last_json(o)
        assert out["valid"] == [1, 1, 1] and out["rows"] == pin["rows"]
        assert out["root"] == pin["root"] and out["sha"] == pin["proof_sha256"]


def _bench(args, timeout=600, **envkw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **envkw)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


TWO_ON_ONE = ["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--log2-constraints", "22", "--sharded-steps", "2",
              "--no-cpu-baseline", "--no-verify"]


def test_bench_py_gpus_2_on_one_gpu_runs_the_sharded_leg_with_real_peers(tmp_path):
    """`python bench.py --gpus 2` (its own launcher) with both ranks on the one GPU (LIG_BENCH_SHARE_GPU, collectives =
    comm_ipc over a gloo rendezvous): the N > 1 code path of the bench -- weak leg on every rank, preflight, then ONE trace
    sharded over the ranks in child processes, at two sizes -- prints one JSON line with n_gpus = 2 and a `sharded` object whose
    envelope equals the oracle pin, delivered by the first rung of the ladder"""
    out = _bench(TWO_ON_ONE + ["--sharded-log2", "22,24"], LIG_BENCH_SHARE_GPU="1", LIG_COMM="ipc", LIG_COMM_TAG=mr.fresh_tag())
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    sh = out["sharded"]
    assert sh["ranks"] == 2 and sh["log2_constraints"] == 24 and sh["scaling"] == "strong"
    assert sh["proof_equals_oracle_pin"] is True and sh["all_ranks_same_envelope"] is True
    assert sh["transport"] == "ipc-stream" and [a["ok"] for a in sh["attempts"]] == [True]
    assert out["sharded_2p22"]["all_ranks_same_envelope"] is True and out["sharded_2p22"]["ranks"] == 2
    pre = out["preflight"]
    assert len(pre["pci_bus_ids"]) == 2 and pre["distinct_devices"] is False         # (both ranks share the one device here, and the line says so)
    assert sh["devices"]["distinct"] is False and "shared one GPU" in sh["devices"]["note"] and sh.get("rccl_ranks") in (None, 0)
    assert set(out["answers"]) == {"value", "sharded"} and "weak" in out["answers"]["value"] and "ONE trace" in out["answers"]["sharded"]
    assert pre["hsa_enable_ipc_mode_legacy"] == "0" and pre["rccl"]["available"] is True


@pytest.mark.parametrize("fault,delivered_by,failed", [
    ("1", "ipc-sync", ["ipc-stream"]),                  # the stream-ordered all-to-all errors out on every rank
    ("2", "host", ["ipc-stream", "ipc-sync"]),          # ... and so does the host-synchronous form of the same communicator
    ("3", "ipc-sync", ["ipc-stream"]),                  # the first rung HANGS (host side): its children are killed at the timeout
])
def test_bench_ladder_falls_through_an_injected_transport_failure_with_two_real_ranks(tmp_path, fault, delivered_by, failed):
    """a failure of the first rung (LIG_FAULT_COMM: injected in the library's communicators) must not cost the
    measurement -- the ladder falls through to the next rung, which still produces the oracle pin's envelope on both ranks"""
    out = _bench(TWO_ON_ONE + ["--sharded-log2", "24", "--sharded-timeout", "30" if fault == "3" else "180"],
                 LIG_BENCH_SHARE_GPU="1", LIG_COMM="ipc", LIG_COMM_TAG=mr.fresh_tag(), LIG_FAULT_COMM=fault)
    sh = out["sharded"]
    assert out["value"] > 0 and sh["transport"] == delivered_by
    assert [a["transport"] for a in sh["attempts"] if not a["ok"]] == failed
    assert sh["proof_equals_oracle_pin"] is True and sh["all_ranks_same_envelope"] is True
    if fault == "3":
        assert "timeout" in sh["attempts"][0]["this_rank"]
    else:
        assert "injected fault" in sh["attempts"][0]["stderr_tail"]


def test_bench_ladder_on_the_real_rccl_library_with_one_rank(tmp_path):
    """the product's first two rungs on librccl itself (all a one-GPU box can run: one rank, exchange forced on): with the
    stream-ordered all-to-all failing, the host-synchronous forms of the same RCCL communicator deliver the pin"""
    out = _bench(["--steps", "2", "--warmup", "1", "--log2-constraints", "22", "--sharded-leg", "--sharded-log2", "24", "--sharded-steps", "2",
                  "--no-cpu-baseline", "--no-verify", "--no-h2d"], LIG_SHARD_FORCE_EXCHANGE="1", LIG_FAULT_COMM="1")
    sh = out["sharded"]
    assert [(a["transport"], a["ok"]) for a in sh["attempts"]] == [("rccl-stream", False), ("rccl-sync", True)]
    assert sh["transport"] == "rccl-sync" and sh["rccl_ranks"] == 1 and sh["proof_equals_oracle_pin"] is True
    assert out["preflight"]["distinct_devices"] is True


def test_torch_transport_zero_copy_on_device_buffers_one_rank(tmp_path):
    """the third rung (`torch`: torch.distributed's own RCCL communicator on the library's device buffers, zero-copy through
    __cuda_array_interface__) with one rank: all_to_all_single / all_gather_into_tensor on raw device pointers"""
    script = tmp_path / "torch_rung.py"
    script.write_text(textwrap.dedent('''
        import hashlib, importlib.util, json, os, sys
        root = sys.argv[1]
        def load(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(root, "ligero-prover_amd", rel))
            m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
        pkg = load("ligero_prover_amd", "__init__.py")
        dist = load("lig_dist", "dist.py")
        import torch
        torch.cuda.set_device(0)
        g = dist.Group("nccl", force_init=True)
        ctx = pkg.Context(320, 512, 2048, device=0)
        job = pkg.Context.make_job(320 * 1500 + 7, 330, generated_at=77)
        comm = g.make_comm(pkg, ctx, "torch")
        sh = ctx.shard_prepare(job, 0, 1, comm)
        proof, info = ctx.shard_prove(sh)
        ctx.shard_destroy(sh)
        tr = ctx.synth_prepare_job(job)
        ref, _ = ctx.synth_prove(tr)
        ctx.trace_destroy(tr)
        print(json.dumps({"same": proof == ref, "valid": [info.valid_code, info.valid_linear, info.valid_quad]}))
        ctx.close(); g.close()
    '''))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(mr.free_port()), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               LIG_SHARD_FORCE_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out == {"same": True, "valid": [1, 1, 1]}


ROWS_WORKER = textwrap.dedent('''
    import ctypes as C, hashlib, importlib.util, json, os, sys
    import numpy as np
    root, l, k, n, n_lin, n_quad, batch, mode = sys.argv[1], *map(int, sys.argv[2:7]), sys.argv[7] == "1", sys.argv[8]
    sys.path.insert(0, os.path.join(root, "tests"))
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "ligero-prover_amd", rel))
        m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
    pkg = load("ligero_prover_amd", "__init__.py")
    dist = load("lig_dist", "dist.py")
    import oracle_lib as ol
    g = dist.Group("gloo")
    ctx = pkg.Context(l, k, n, device=0)
    # the oracle plays the guest + witness_manager: ALL rows, their kinds, later the randomness rows (every rank runs the
    # same deterministic "guest" and keeps its slice, as N copies of a constraint generator would)
    job = ol.make_job(l, k, n, 192, n_lin, n_quad, generated_at=55, threads=4)
    if batch:
        import test_batch_rows
        test_batch_rows.demo_program().attach(job)
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job).copy()
    rounds, b = pkg.shard_rows_plan(kinds, g.world)
    mine = pkg.local_rows_of(b, g.rank, g.world)
    local = rows[mine] if len(mine) else np.zeros((0, k, 8), dtype=np.uint32)
    kk = kinds.copy()
    if mode == "library_pads":                       # the library draws the pads of every row that draws upstream
        draws = (kinds <= 3) | (kinds == pkg.ROW_KINDS["INIT"])
        kk[draws] |= pkg.ROW_DRAW_PAD
        local = local.copy()
        if len(mine):
            local[draws[mine], l:] = 0xDEADBEEF
    comm = g.make_comm(pkg, ctx)
    dense = None
    if mode == "dense_rands":                        # the synthetic stream's dense coefficient rows are generated by the library
        per = [l] * (n_lin // l) + [l] * (3 * (n_quad // l)) + ([n_lin % l] if n_lin % l else []) + ([n_quad % l] * 3 if n_quad % l else [])
        dense = np.array([0] * (len(kinds) - len(per)) + per, dtype=np.uint32)       # batch rows (in front) carry no randomness
    if mode == "device_rows":
        d_local = ctx.upload(local) if len(mine) else ctx.malloc(32)
        sh = ctx.shard_rows_begin(kk, d_local, g.rank, g.world, comm, on_device=True, generated_at=55)
    else:
        sh = ctx.shard_rows_begin(kk, local, g.rank, g.world, comm, generated_at=55, dense_rands_per_row=dense)
    out = []
    for rep in range(2):                             # the second pass: lig_shard_rows_restart with the same rows
        if rep:
            ctx.check(ctx.L.lig_shard_rows_restart(sh, d_local if mode == "device_rows" else C.c_void_p(local.ctypes.data if local.size else None), int(mode == "device_rows")))
        root_, seed1 = ctx.shard_rows_commit(sh)
        rands, const_sum = ol.rand_rows(job, seed1)
        lr = rands[mine] if len(mine) else np.zeros((0, k, 8), dtype=np.uint32)
        if mode == "device_rows":
            d_r = ctx.upload(lr) if len(mine) else ctx.malloc(32)
            proof, info = ctx.shard_rows_prove(sh, d_r, const_sum if rep == 0 else None, on_device=True)
        elif mode == "dense_rands":
            proof, info = ctx.shard_rows_prove(sh, None, const_sum if rep == 0 else None)
        else:
            proof, info = ctx.shard_rows_prove(sh, lr, const_sum if rep == 0 else None)
        out.append((proof, bytes(info.const_sum) == const_sum, [info.valid_code, info.valid_linear, info.valid_quad]))
    ctx.shard_destroy(sh)
    ref = oref = None
    if g.rank == 0:                                  # the unsharded rows entry on the same rows, and the oracle's prover
        tr, keep = ctx.rows_begin(kinds, rows, generated_at=55)
        ctx.rows_commit(tr)
        ref, _ = ctx.rows_prove(tr, rands, const_sum)
        ctx.trace_destroy(tr)
        pr = ol.Proof()
        assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
        oref = bytes(pr.proof[:pr.proof_len])
        ol.lib().lo_proof_free(C.byref(pr))
    digs = g.gather_digests(hashlib.sha256(out[0][0]).digest())
    print(json.dumps({"rank": g.rank, "local_rows": len(mine), "rounds": rounds, "again": out[0][0] == out[1][0], "const": out[0][1] and out[1][1],
                      "valid": out[0][2], "all_equal": len(set(digs)) == 1,
                      "equals_rows_prove": None if ref is None else ref == out[0][0], "equals_oracle": None if oref is None else oref == out[0][0]}))
    g.close(); ctx.close()
''')


def run_rows_world(tmp_path, world, l, k, n, n_lin, n_quad, batch, mode, comm, timeout=300, **env):
    script = tmp_path / "shard_rows_worker.py"
    script.write_text(ROWS_WORKER)
    outs = mr.run_ranks(mr.python_argv(script, ROOT, l, k, n, n_lin, n_quad, "1" if batch else "0", mode), world, mr.rendezvous_env(world, comm, **env), timeout=timeout)
    return sorted((mr.last_json(o) for o, _ in outs), key=lambda d: d["rank"])


@pytest.mark.parametrize("world,n_lin,n_quad,batch,mode,comm", [
    (2, 2000, 900, False, "library_pads", None),          # gloo callbacks, library draws the pads
    (2, 2000, 900, False, "own_pads", "ipc"),             # stream-ordered peers, rows carry their pads
    (4, 700, 0, False, "library_pads", "ipc"),            # 3 rows on 4 ranks: a rank without rows
    (2, 900, 330, True, "own_pads", "ipc"),               # batch rows (init / equal / product / bit) dealt like any other rows
    (4, 900, 330, True, "library_pads", None),
    (2, 320 * 1500 + 7, 330, False, "device_rows", "ipc"),      # two exchange rounds, local rows and randomness rows on the device
    (4, 320 * 4300 + 1, 0, False, "library_pads", "ipc"),       # three rounds on 4 ranks
    (2, 320 * 700 + 9, 330 + 5, False, "dense_rands", "ipc"),  # the dense coefficient rows generated (and accumulated) by the library on every rank
    (4, 900, 330, True, "dense_rands", None),
    (8, 320 * 9000 + 11, 330, False, "library_pads", "ipc"),   # the node's shape (round 6): 8 ranks, three rounds, caller rows
    (8, 900, 330, True, "own_pads", "ipc"),                    # 8 ranks, fewer chunks than ranks: most ranks own no row
])
def test_sharded_rows_entry_equals_rows_prove_and_oracle(tmp_path, world, n_lin, n_quad, batch, mode, comm):
    """lig_shard_rows_*: one trace whose rows come from the caller, sharded over W ranks (each rank passes all kinds + its own
    rows and randomness rows): every rank's envelope == lig_rows_prove on the whole trace == the oracle's prover"""
    outs = run_rows_world(tmp_path, world, 320, 512, 2048, n_lin, n_quad, batch, mode, comm)
    assert all(o["valid"] == [1, 1, 1] and o["again"] and o["const"] and o["all_equal"] for o in outs), outs
    assert outs[0]["equals_rows_prove"] is True and outs[0]["equals_oracle"] is True, outs
    if world == 4 and n_lin == 700:
        assert min(o["local_rows"] for o in outs) == 0


FAILING_PEER_WORKER = textwrap.dedent('''
    import importlib.util, json, os, signal, sys, time
    import numpy as np
    root, how = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.join(root, "tests"))
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "ligero-prover_amd", rel))
        m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
    pkg = load("ligero_prover_amd", "__init__.py")
    dist = load("lig_dist", "dist.py")
    import oracle_lib as ol
    l, k, n = 320, 512, 2048
    g = dist.Group("gloo")
    ctx = pkg.Context(l, k, n, device=0)
    job = ol.make_job(l, k, n, 192, 320 * 1500 + 7, 330, generated_at=55, threads=4)      # two exchange rounds
    rows, _, _, _ = ol.form_rows(job)
    kinds = ol.row_kinds(job).copy()
    rounds, b = pkg.shard_rows_plan(kinds, g.world)
    mine = pkg.local_rows_of(b, g.rank, g.world)
    local = rows[mine]
    comm = g.make_comm(pkg, ctx)
    sh = ctx.shard_rows_begin(kinds, local, g.rank, g.world, comm, generated_at=55)
    g.barrier()
    if g.rank == 1:
        if how == "killed":                          # the process disappears (no destructor runs)
            os.kill(os.getpid(), signal.SIGKILL)
        if how == "leaves":                          # an orderly exit of a rank that never joins the collective
            ctx.shard_destroy(sh); ctx.ipc_comm_destroy(comm); ctx.close(); os._exit(0)
        if how == "never_arrives":                   # alive, but never reaches the collective (host skew without end)
            time.sleep(25); os._exit(0)
        # (how == "gpu_stall": this rank runs the call like rank 0; LIG_FAULT_COMM=4 makes it never raise its `ready` flag)
    t0 = time.time()
    try:
        ctx.shard_rows_commit(sh)
        out = {"error": None}
    except pkg.LigError as e:
        out = {"error": str(e)}
    out["seconds"] = time.time() - t0
    t1 = time.time()
    try:                                             # the communicator is dead: the next call fails at once
        ctx.shard_rows_commit(sh)
        out["again"] = None
    except pkg.LigError as e:
        out["again"] = str(e)
    out["again_seconds"] = time.time() - t1
    ctx.shard_destroy(sh)
    ctx.close()
    out["closed_after"] = time.time() - t0
    print(json.dumps(out), flush=True)
    os._exit(0)                                      # (no process-group barrier with a dead peer)
''')


@pytest.mark.parametrize("how,needle,limit,knobs", [
    ("killed", "is gone", 30, {}),                                # the peer's process is killed: noticed through its pid
    ("leaves", "left the communicator", 30, {}),                  # the peer destroys its communicator without ever joining the collective
    # the peer is alive but never reaches the collective: that is host skew, bounded by the HOST wait of the ranks that wait for its
    # publication (LIG_IPC_HOST_S = 5 here, 120 by default) -- not by the stall timer, which must not fire on a slow constraint generator
    ("never_arrives", "never reached collective", 30, dict(LIG_IPC_HOST_S=5, LIG_IPC_STALL_S=2)),
    # every rank has published the collective on the host, but a peer's flag never comes (LIG_FAULT_COMM=4): the queued waits would never
    # complete -- the watchdog's stall timer (LIG_IPC_STALL_S = 5 here) declares the communicator dead and releases them
    # (both ranks run the same timer: rank 0 reports its own finding or the abort word rank 1's watchdog raised a moment earlier)
    ("gpu_stall", ("no flag of any rank changed", "rank 1 declared the communicator dead"), 40, dict(LIG_IPC_STALL_S=5, LIG_FAULT_COMM=4)),
])
def test_a_failing_peer_makes_the_sharded_call_return_an_error_instead_of_hanging(tmp_path, how, needle, limit, knobs):
    """stream-ordered collectives wait inside GPU queues, where nothing times out.  comm_ipc's watchdog thread
    declares the communicator dead (dead pid / departed peer / stall), releases the queued waits, and lig_shard_rows_commit returns
    LIG_E_STATE with the reason -- in seconds, with every stream drained so that the shard and the context can be destroyed."""
    script = tmp_path / "failing_peer_worker.py"
    script.write_text(FAILING_PEER_WORKER)
    env = mr.rendezvous_env(2, "ipc", **knobs)
    tmpd = tmp_path / "ranks"
    tmpd.mkdir()
    procs, files = [], []
    for r in range(2):
        fo, fe = open(tmpd / ("out%d" % r), "w+b"), open(tmpd / ("err%d" % r), "w+b")
        files.append((fo, fe))
        procs.append(subprocess.Popen(mr.python_argv(script, ROOT, how), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=fo, stderr=fe, start_new_session=True))
    try:
        rc0 = procs[0].wait(timeout=300)
    finally:
        for p in procs:
            mr._kill_group(p)
    files[0][0].seek(0); files[0][1].seek(0)
    o, e = files[0][0].read().decode(), files[0][1].read().decode()
    assert rc0 == 0, e[-3000:]
    out = mr.last_json(o)
    assert out["error"] and any(nd in out["error"] for nd in ((needle,) if isinstance(needle, str) else needle)), out
    assert out["seconds"] < limit and out["closed_after"] < limit + 10, out
    assert out["again"] and out["again_seconds"] < 5, out


def test_gpus_sweep_1_2_8_on_one_gpu_is_a_labelled_functional_run(tmp_path):
    """round 6: the hardware-day command `python bench.py --gpus-sweep 1,2,4,8` end to end on this one-GPU box
    (LIG_BENCH_SHARE_GPU=1: all ranks of an N on GPU 0, comm_ipc over gloo): per N one line with the weak figure, the sharded (strong) figure
    whose envelope equals the oracle pin on every rank, the HBM fraction -- and `shared_device: true` for N > 1, so that the numbers cannot
    be mistaken for a scaling curve.  N = 8 is the node's shape: the 8-rank deal, the 7-peer exchange, the 8-way all-gathers."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", LIG_BENCH_SHARE_GPU="1")
    args = ["--gpus-sweep", "1,2,8", "--steps", "2", "--warmup", "1", "--log2-constraints", "22", "--sharded-log2", "22,24", "--sharded-steps", "2",
            "--sharded-leg", "--no-cpu-baseline", "--no-verify", "--no-h2d", "--quad-mix", "0", "--sharded-timeout", "300"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, timeout=1500)
    assert p.returncode == 0, (p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
    lines = [json.loads(ln) for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 4 and [ln["n_gpus"] for ln in lines[:3]] == [1, 2, 8]
    for ln in lines[:3]:
        assert "error" not in ln and ln["value"] > 0 and ln["scaling"] == "weak" and 0 < ln["hbm_frac"] < 1, ln
        assert ln["shared_device"] is (ln["n_gpus"] > 1)
        sh = ln["sharded"]
        assert sh["ranks"] == ln["n_gpus"] and sh["log2_constraints"] == 24 and sh["scaling"] == "strong", ln
        assert sh["proof_equals_oracle_pin"] is True and sh["all_ranks_same_envelope"] is True, ln
        assert ln["sharded_2p22"]["all_ranks_same_envelope"] is True, ln
    assert lines[2]["sharded"]["transport"] == "ipc-stream" and lines[2]["distinct_devices"] is False
