"""bench.py's own launcher (CPU, gloo, stub workload): `python bench.py --gpus N` with no launcher around it must start N
ranks itself and print ONE JSON line with n_gpus = N; a world size that disagrees with --gpus is an error, not a silent
N = 1 run (`--gpus` was parsed and never read)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_ranks_and_reports_them():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["spawned_by_bench"] is True
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 1000 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]      # whole-job units / MAX time


def test_under_an_external_launcher_the_ranks_are_used_as_given():
    import multirank as mr
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "2", "--warmup", "0"]
    res = mr.run_ranks(argv, 2, mr.rendezvous_env(2, base=_env()), timeout=180)
    outs = [(o.encode(), e.encode()) for o, e in res]
    line0 = [ln for ln in outs[0][0].decode().splitlines() if ln.startswith("{")]
    assert len(line0) == 1 and not [ln for ln in outs[1][0].decode().splitlines() if ln.startswith("{")]
    out = json.loads(line0[0])
    assert out["n_gpus"] == 2 and out["spawned_by_bench"] is False


def test_world_size_that_disagrees_with_gpus_fails_loudly():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub"],
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"), capture_output=True, timeout=120)
    assert p.returncode != 0 and b"--gpus 2 but WORLD_SIZE=1" in p.stderr
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub"],
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(__import__("multirank").free_port())), capture_output=True, timeout=120)
    assert p.returncode != 0 and b"--gpus 1 but WORLD_SIZE=2" in p.stderr


def test_a_dying_rank_ends_the_launcher():
    """rank 1 has no GPU here (no GPU at all in the CPU container): the launcher must return non-zero instead of hanging"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_env(HIP_VISIBLE_DEVICES=""), capture_output=True, timeout=300)
    assert p.returncode != 0


def test_under_torch_distributed_run_as_the_driver_launches_it():
    """the driver's N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ... (stub workload, gloo)"""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29961", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["spawned_by_bench"] is False and out["steps"] == 2


def _ladder(extra, launcher=None, timeout=300):
    cmd = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "2", "--warmup", "0", "--sharded-leg",
           "--sharded-timeout", "20", "--sharded-log2", "24,26"] + extra      # (20 s: a rung's children import torch + rendezvous; 10 s was seen to be too short once on a loaded box)
    cmd = (launcher or [sys.executable]) + cmd
    p = subprocess.run(cmd, env=_env(), capture_output=True, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_transport_ladder_falls_through_a_dead_rank_and_a_hung_rank():
    """the sharded leg of `bench.py --gpus N` runs every rung in fresh child processes with their own rendezvous:
    a rung whose LAST rank exits (the others then fail in their barrier) and a rung whose last rank hangs (every child is killed
    after --sharded-timeout) are recorded as failed attempts, the third rung delivers, rc stays 0 and the weak figure is intact"""
    out = _ladder(["--sharded-transports", "stub-fail,stub-hang,stub-ok"])
    sh = out["sharded"]
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert [a["transport"] for a in sh["attempts"]] == ["stub-fail", "stub-hang", "stub-ok"]
    assert [a["ok"] for a in sh["attempts"]] == [False, False, True]
    assert "timeout" in sh["attempts"][1]["this_rank"] and sh["attempts"][0]["stderr_tail"]
    assert sh["transport"] == "stub-ok" and sh["log2_constraints"] == 26 and sh["ranks"] == 2


def test_transport_ladder_with_every_rung_failing_still_prints_the_weak_line():
    out = _ladder(["--sharded-transports", "stub-fail,stub-fail"])
    assert out["value"] > 0 and "error" in out["sharded"] and len(out["sharded"]["attempts"]) == 2


def test_transport_ladder_time_budget_stops_further_rungs():
    out = _ladder(["--sharded-transports", "stub-hang,stub-ok", "--sharded-budget", "30"])      # (a rung starts while elapsed + 15 s < budget; a hung rung takes --sharded-timeout = 20 s)
    att = out["sharded"]["attempts"]
    assert att[0]["ok"] is False and "skipped" not in att[0] and "skipped" in att[1] and "error" in out["sharded"]


def test_transport_ladder_under_torch_distributed_run():
    """the children must not inherit the elastic agent's store (TORCHELASTIC_USE_AGENT_STORE): they rendezvous on their own port"""
    out = _ladder(["--sharded-transports", "stub-fail,stub-ok"],
                  launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29963"])
    assert out["spawned_by_bench"] is False and out["sharded"]["transport"] == "stub-ok"
    assert [a["ok"] for a in out["sharded"]["attempts"]] == [False, True]


def test_gpus_sweep_runs_every_n_and_prints_one_line_per_n_plus_a_summary():
    """round 6: `python bench.py --gpus-sweep 1,2,4` = the bench once per N, each its own `bench.py --gpus N`; one compact
    line per N (value, the sharded figure, RCCL ranks, HBM fraction) and a summary line -- here with the stub workload (no GPU)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus-sweep", "1,2,4", "--workload", "stub", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 4
    assert [ln["n_gpus"] for ln in lines[:3]] == [1, 2, 4] and all(ln["value"] > 0 and ln["scaling"] == "weak" and ln["steps"] == 2 for ln in lines[:3])
    assert lines[3]["gpus_sweep"] == lines[:3] and "shared_device" in lines[3]["note"]
    # the keys a reader of the sweep needs are there even when the workload has nothing to say about them
    assert all({"value", "sharded", "rccl_ranks", "hbm_frac", "shared_device"} <= set(ln) for ln in lines[:3])
