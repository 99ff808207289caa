"""world_size-2 gloo test (CPU) of the N>1 plumbing used by bench.py: job dealing, barrier, MAX/SUM reductions,
digest exchange.  The per-rank compute is HIP-only and is covered by the -m gpu tests."""
import hashlib
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import hashlib, importlib.util, json, os, sys, time
    root = sys.argv[1]
    spec = importlib.util.spec_from_file_location("lig_dist", os.path.join(root, "ligero-prover_amd", "dist.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    g = m.Group("gloo")
    jobs = g.my_jobs(7)
    g.barrier()
    t = 0.010 * (g.rank + 1)                       # rank 1 is the slow one
    worst = g.max_over_ranks(t)
    total = g.sum_over_ranks(len(jobs) * 1000)
    digs = g.gather_digests(hashlib.sha256(bytes([g.rank])).digest())
    print(json.dumps({"rank": g.rank, "world": g.world, "jobs": jobs, "worst": worst, "total": total,
                      "digests": [d.hex() for d in digs]}))
    g.close()
''')


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import multirank as mr
    res = mr.run_ranks(mr.python_argv(script, ROOT), 2, mr.rendezvous_env(2), timeout=300)
    outs = [__import__("json").loads(o.strip().splitlines()[-1]) for o, _ in res]
    outs.sort(key=lambda d: d["rank"])
    assert [o["world"] for o in outs] == [2, 2]
    assert outs[0]["jobs"] == [0, 2, 4, 6] and outs[1]["jobs"] == [1, 3, 5]            # every job exactly once
    assert all(abs(o["worst"] - 0.020) < 1e-12 for o in outs)                          # MAX over ranks
    assert all(o["total"] == 7000 for o in outs)                                       # whole-job units
    want = [hashlib.sha256(bytes([r])).hexdigest() for r in range(2)]
    assert all(o["digests"] == want for o in outs)


def test_single_process_group_is_a_noop():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lig_dist", os.path.join(ROOT, "ligero-prover_amd", "dist.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        g = m.Group()
        assert (g.rank, g.world) == (0, 1) and g.my_jobs(3) == [0, 1, 2]
        assert g.max_over_ranks(1.5) == 1.5 and g.sum_over_ranks(4) == 4
        g.barrier(); g.close()
    finally:
        os.environ.update(env)


def test_shard_plan_block_cyclic_deal():
    """host-only view of the sharded prover's row deal (lig_shard_plan): boundaries cover the rows in order, chunks hold at
    most 512 (+2) rows, never split an x,y,z triple or an equality pair, are balanced, and chunk g goes to rank g mod W"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hip_lib
    import oracle_lib as ol
    import test_batch_rows as tb
    amd = hip_lib.load()
    l, k, n = 320, 512, 2048
    for (nl, nq, prog) in ((700, 0, False), (320 * 1500 + 7, 330, False), (0, 0, False), (100, 320 * 40 + 3, True), (320 * 4300 + 1, 0, False)):
        oj = ol.make_job(l, k, n, 192, nl, nq)
        hj = amd.Context.make_job(nl, nq)
        if prog:
            tb.demo_program().attach(oj); tb.demo_program().attach(hj)
        kinds = list(ol.row_kinds(oj))
        R = len(kinds)
        for W in (1, 2, 4, 8):
            rounds, b = amd.shard_plan(hj, l, W)
            assert rounds == max(1, -(-R // (W * 512))) and len(b) == rounds * W + 1
            assert b[0] == 0 and b[-1] == R and all(x <= y for x, y in zip(b, b[1:]))
            sizes = [y - x for x, y in zip(b, b[1:])]
            assert max(sizes, default=0) <= 512 + 2
            for x in b[1:-1]:
                if x < R:
                    assert kinds[x] not in (2, 3, 7, 9, 10), "a chunk boundary inside a row group"
            per_rank = [sum(sizes[g] for g in range(r, len(sizes), W)) for r in range(W)]
            assert sum(per_rank) == R and max(per_rank) - min(per_rank) <= max(3 * rounds + (R % (W * rounds) != 0) * (-(-R // (W * rounds))), 3)


def test_transport_names_and_default_ladders(monkeypatch):
    """dist.Group's transports (the rungs of bench.py's ladder) and which ladder bench.py climbs where"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lig_dist", os.path.join(ROOT, "ligero-prover_amd", "dist.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    spec = importlib.util.spec_from_file_location("lig_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LIG_COMM"):
        monkeypatch.delenv(k, raising=False)
    g = m.Group("nccl")                                   # world 1: no process group is made
    assert g.default_transport() == "rccl-stream" and m.Group("gloo").default_transport() == "host"
    monkeypatch.setenv("LIG_COMM", "ipc")
    assert g.default_transport() == "ipc-stream" and b.ladder_transports(2) == ["ipc-stream", "ipc-sync", "host"]
    monkeypatch.delenv("LIG_COMM")
    assert b.ladder_transports(8) == ["rccl-stream", "rccl-sync", "torch", "ipc-stream"]        # product path first
    assert b.ladder_transports(1) == ["rccl-stream", "rccl-sync", "host"]
    assert set(b.ladder_transports(8)) | set(b.ladder_transports(1)) <= set(m.Group.TRANSPORTS) and set(b.TRANSPORT_NOTES) == set(m.Group.TRANSPORTS)
    import pytest
    with pytest.raises(ValueError):
        g.make_comm(None, None, "carrier-pigeon")
