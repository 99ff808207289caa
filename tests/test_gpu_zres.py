"""LIG_ZRES=1 (round 6): stage 1 keeps the encoder's Z tiles instead of codeword planes -- the last radix-8 pass of the row
encoder (src/webgpu/engine.cpp:844-882's last stages) runs inside the column hash's producer waves (shader/sha256.wgsl:148-177),
stage 2 / 3 take single radix-8 outputs from the tiles.  The knob is read once per process, so every case runs in a child process
with the variable set; the proofs must equal the oracle's byte for byte (same bar as the default build)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import ctypes as C, hashlib, json, os, sys
    root = sys.argv[1]; cases = json.loads(sys.argv[2])
    sys.path.insert(0, os.path.join(root, "tests"))
    import hip_lib, oracle_lib as ol
    amd = hip_lib.load()
    out = []
    for case in cases:
        if "pin" in case:
            with open(os.path.join(root, "tests", "golden", case["pin"])) as f:
                pin = json.load(f)
            c = amd.Context(pin["l"], pin["k"], pin["n"])
            tr = c.synth_prepare(pin["n_linear"], pin["n_quad"], synth_seed=pin["synth_seed"], generated_at=pin["generated_at"])
            proof, info = c.synth_prove(tr)
            proof2, _ = c.synth_prove(tr)
            job = amd.Context.make_job(pin["n_linear"], pin["n_quad"], synth_seed=pin["synth_seed"], generated_at=pin["generated_at"])
            acc = c.synth_verify(job, None, proof).accept
            c.trace_destroy(tr); c.close()
            out.append(dict(case=case, ok=hashlib.sha256(proof).hexdigest() == pin["proof_sha256"] and bytes(info.root).hex() == pin["root"] and proof == proof2,
                            accept=acc))
            continue
        l, k, n, nl, nq = case["shape"]
        c = amd.Context(l, k, n)
        tr = c.synth_prepare(nl, nq, generated_at=5)
        proof, info = c.synth_prove(tr)
        proof2, _ = c.synth_prove(tr)
        c.trace_destroy(tr); c.close()
        job = ol.make_job(l, k, n, 192, nl, nq, generated_at=5, threads=8)
        pr = ol.Proof()
        assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
        want = bytes(pr.proof[:pr.proof_len])
        out.append(dict(case=case, ok=proof == want and proof == proof2 and bytes(info.root) == bytes(pr.root),
                        valid=[info.valid_code, info.valid_linear, info.valid_quad]))
        ol.lib().lo_proof_free(C.byref(pr))
    print(json.dumps(out))
''')


def run_cases(tmp_path, cases, timeout=600, env_extra=None):
    script = tmp_path / "zres_child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LIG_ZRES="1")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(script), ROOT, json.dumps(cases)], env=env, capture_output=True, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("[")][-1])


@pytest.mark.parametrize("env", [dict(LIG_ZRES="1"), dict(LIG_ZRES="1", LIG_SHA_GATE="2"), dict(LIG_ZRES="0", LIG_SHA_GATE="2")],
                         ids=["zres", "zres-k1ahead", "planar-k1ahead"])
def test_zres_small_shapes_equal_the_oracle(tmp_path, env):
    """k = 512 / 1024 (tile lengths 64 and 128): empty statement, one row, odd and even row counts, more rows than one batch of
    the in-hash butterflies (8), chunk boundaries of the 512-row launches (pending half blocks between launches), and a trace with
    quadratic triples (which keeps the planar matrix: the knob must not change its proof either).  Also with LIG_SHA_GATE=2 (K1 of the
    next chunk pipelined ahead of the hash placement), with and without the resident tiles."""
    cases = [dict(shape=s) for s in [
        (320, 512, 2048, 0, 0), (320, 512, 2048, 1, 0), (320, 512, 2048, 320 * 2, 0), (320, 512, 2048, 320 * 7 + 5, 0),
        (320, 512, 2048, 320 * 8, 0), (320, 512, 2048, 320 * 9, 0), (320, 512, 2048, 320 * 23 + 1, 0),
        (320, 512, 2048, 320 * 511 + 3, 0), (320, 512, 2048, 320 * 641, 0), (320, 512, 2048, 320 * 1290, 0),
        (832, 1024, 4096, 832 * 30 + 7, 0), (320, 512, 2048, 640, 330)]]
    for r in run_cases(tmp_path, cases, env_extra=env):
        assert r["ok"] is True and r["valid"] == [1, 1, 1], r


@pytest.mark.parametrize("ws", ["1", "2"])
def test_zres_production_geometry_and_pin(tmp_path, ws):
    """k = 8192 (tile length 1024): a few rows against the oracle, then configs[2]'s 2^24-constraint job against its pin
    (tests/golden/full_pin_2p24.json); with one and with two column groups per hash workgroup"""
    cases = [dict(shape=(8000, 8192, 32768, 8000 * 19 + 123, 0)), dict(pin="full_pin_2p24.json")]
    res = run_cases(tmp_path, cases, env_extra=dict(LIG_SHA_WS=ws))
    assert res[0]["ok"] is True and res[0]["valid"] == [1, 1, 1], res[0]
    assert res[1]["ok"] is True and res[1]["accept"] == 1, res[1]


def test_zres_other_tile_lengths(tmp_path):
    """tile lengths 2048 and 4096 (k = 16384 / 32768)"""
    cases = [dict(shape=(16192, 16384, 65536, 16192 * 11 + 9, 0)), dict(shape=(32576, 32768, 131072, 32576 * 10 + 1, 0))]
    for r in run_cases(tmp_path, cases):
        assert r["ok"] is True and r["valid"] == [1, 1, 1], r
