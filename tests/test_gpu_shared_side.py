"""One side stream per device shared by every context of the process (LIG_SHARED_SIDE=1, the default since round 6; csrc/lig_capi.hip:
lig_ctx_create) -- and the per-context side stream of rounds 1-5 (LIG_SHARED_SIDE=0).  Three contexts proving from three host threads at the
same time (stage 1 hash, stage-2 samplers and the mask-row transforms of all of them on the one side stream; a context created and destroyed
while the others keep proving: the shared stream outlives it) must give the envelopes each job gives alone = the oracle's.  No reference
counterpart (src/webgpu/device_context.cpp:344-354: one in-order queue); the knob is read once per process: child processes."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import ctypes as C, json, os, sys, threading
    root = sys.argv[1]
    sys.path.insert(0, os.path.join(root, "tests"))
    import hip_lib, oracle_lib as ol
    amd = hip_lib.load()
    jobs = [(320, 512, 2048, 320 * 700 + 3, 330, 11), (320, 512, 2048, 320 * 1300, 0, 12), (8000, 8192, 32768, 8000 * 150 + 1, 8000, 13)]
    want = []
    for (l, k, n, nl, nq, ts) in jobs:
        job = ol.make_job(l, k, n, 192, nl, nq, generated_at=ts, threads=8)
        pr = ol.Proof()
        assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
        want.append(bytes(pr.proof[:pr.proof_len]))
        ol.lib().lo_proof_free(C.byref(pr))
    bad = []
    def prove(i, reps):
        l, k, n, nl, nq, ts = jobs[i]
        for rep in range(reps):                     # a fresh context per repetition: contexts come and go while the others prove
            c = amd.Context(l, k, n)
            try:
                tr = c.synth_prepare(nl, nq, generated_at=ts)
                for _ in range(3):
                    proof, info = c.synth_prove(tr)
                    if proof != want[i] or [info.valid_code, info.valid_linear, info.valid_quad] != [1, 1, 1]:
                        bad.append((i, rep))
                c.trace_destroy(tr)
            finally:
                c.close()
    th = [threading.Thread(target=prove, args=(i, 4)) for i in range(3)]
    [t.start() for t in th]; [t.join() for t in th]
    print(json.dumps(dict(proofs=3 * 4 * 3, bad=bad)))
''')


@pytest.mark.parametrize("shared", ["1", "0"])
def test_contexts_sharing_one_side_stream_prove_independently(tmp_path, shared):
    script = tmp_path / "shared_side_child.py"
    script.write_text(CHILD)
    p = subprocess.run([sys.executable, str(script), ROOT], env=dict(os.environ, LIG_SHARED_SIDE=shared), capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["proofs"] == 36 and out["bad"] == [], out
