"""Launcher for tests that run W ranks as W processes (one GPU shared by all of them, or none).

What round 4 taught this file (one hang of a world-2 test cost 900 s, seven tests and the evidence of why):
  * every invocation gets its OWN rendezvous port (asked from the kernel) and its OWN communicator tag (pid + counter), so that a
    shared-memory segment or a listening socket of an earlier test can never be mistaken for this one's;
  * all ranks are watched together: the first rank that exits non-zero ends the run (the others get a few seconds to fail by
    themselves -- their error text is the interesting one -- and are then killed, whole process group);
  * a timeout kills ALL ranks (no orphan keeps the GPU) and the assertion carries EVERY rank's stderr tail;
  * timeouts are sized to the box, not to the test (default 300 s: the slowest multi-rank test takes 12 s on the target box, but a box
    whose page cache has just been dropped imports torch in a minute or two per process; a real hang is ended by the watchdogs of the
    library in seconds and by this timeout at the latest, and `pytest.ini` caps every test at 420 s).
"""
import itertools
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

_counter = itertools.count(1)


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def fresh_tag():
    return "%d_%d" % (os.getpid(), next(_counter))


def rendezvous_env(world, comm=None, base=None, **extra):
    """environment common to all ranks of one invocation (RANK / LOCAL_RANK are added per rank by run_ranks)"""
    env = dict(os.environ if base is None else base)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if comm:
        env.update(LIG_COMM=comm, LIG_COMM_TAG=fresh_tag())
    env.update({k: str(v) for k, v in extra.items()})
    return env


class RankFailure(AssertionError):
    pass


def _kill_group(p):
    if p.poll() is not None:
        return
    try:
        os.killpg(p.pid, signal.SIGKILL)         # start_new_session: pid == pgid; takes the rank's own children with it
    except (ProcessLookupError, PermissionError):
        try:
            p.kill()
        except ProcessLookupError:
            pass


def run_ranks(argv, world, env, timeout=300.0, grace=8.0, tail=3000):
    """argv (a list: the same for every rank; or a function rank -> list) as `world` processes with RANK / LOCAL_RANK = 0 .. world-1.
    Returns [(stdout, stderr)] in rank order when every rank exits 0; raises RankFailure with every rank's state otherwise."""
    tmp = tempfile.mkdtemp(prefix="lig_ranks_")
    files, procs = [], []
    for r in range(world):
        fo, fe = open(os.path.join(tmp, "out%d" % r), "w+b"), open(os.path.join(tmp, "err%d" % r), "w+b")
        files.append((fo, fe))
        procs.append(subprocess.Popen(argv(r) if callable(argv) else argv, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=fo, stderr=fe, start_new_session=True))
    t0 = time.monotonic()
    verdict, first_bad = None, None
    try:
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs):
                break
            now = time.monotonic()
            if first_bad is None and any(rc not in (None, 0) for rc in rcs):
                first_bad = now                      # a rank has failed: the others either fail too (soon) or hang on it
            if first_bad is not None and now - first_bad > grace:
                verdict = "a rank failed and the others did not end within %.0f s of it" % grace
                break
            if now - t0 > timeout:
                verdict = "timeout after %.0f s" % timeout
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            _kill_group(p)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
    outs = []
    for fo, fe in files:
        fo.seek(0); fe.seek(0)
        outs.append((fo.read().decode(errors="replace"), fe.read().decode(errors="replace")))
        fo.close(); fe.close()
    for name in os.listdir(tmp):
        os.unlink(os.path.join(tmp, name))
    os.rmdir(tmp)
    rcs = [p.returncode for p in procs]
    if verdict is None and all(rc == 0 for rc in rcs):
        return outs
    lines = ["%s; exit codes by rank: %r (negative = killed by that signal)" % (verdict or "a rank exited non-zero", rcs)]
    for r, (o, e) in enumerate(outs):
        lines.append("---- rank %d (rc %r) stderr tail:\n%s" % (r, rcs[r], e[-tail:]))
        if o.strip():
            lines.append("---- rank %d stdout tail:\n%s" % (r, o[-tail // 2:]))
    raise RankFailure("\n".join(lines))


def last_json(stdout):
    import json
    return json.loads([ln for ln in stdout.splitlines() if ln.startswith("{")][-1])


def python_argv(script, *args):
    return [sys.executable, str(script)] + [str(a) for a in args]
