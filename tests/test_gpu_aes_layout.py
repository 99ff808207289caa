"""LIG_AES_LAYOUT (round 6; 1 = default): the sampler's AES tables entry-major in LDS so that a lookup address is one v_perm_b32
(csrc/aes.hip).  Same keystream, same field elements (include/util/csprng.hpp:54-107, include/zkp/finite_field_gmp.hpp:66-78): the big
launches (>= 2^20 elements: persistent workgroups with replicated tables) against the oracle's sampler, a 300-row proof at k = 8192
against the oracle's prover, configs[2]'s 2^24 pin.  The knob is read once per process: child process."""
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
    import numpy as np
    root = sys.argv[1]
    sys.path.insert(0, os.path.join(root, "tests"))
    import hip_lib, oracle_lib as ol
    amd = hip_lib.load()
    out = {}
    c = amd.Context(8000, 8192, 32768)
    key = hashlib.sha256(b"layout").digest()
    for count, first in (((1 << 20) + 37, 0), ((1 << 21) + 5, (1 << 33) + 12345)):       # big launches; a stream position beyond 2^32 blocks
        d = c.malloc(32 * count)
        c.rng_fill(key, first, d, count)
        got = c.download(d, (count, 8))
        want = ol.rng_fill(key, first, count).reshape(count, 8)
        out["fill_%d" % count] = bool(np.array_equal(got, want))
        c.free(d)
    # dense randomness rows + the fused sampler / accumulator of stage 2 (k_rand_rlc), pads and masks (k_rng_fill_rows): a proof
    nl = 8000 * 300 + 11
    tr = c.synth_prepare(nl, 0, generated_at=5)
    proof, info = c.synth_prove(tr)
    c.trace_destroy(tr)
    job = ol.make_job(8000, 8192, 32768, 192, nl, 0, generated_at=5, threads=8)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(job), C.byref(pr)) == 0
    out["proof_300_rows"] = proof == bytes(pr.proof[:pr.proof_len]) and [info.valid_code, info.valid_linear, info.valid_quad] == [1, 1, 1]
    with open(os.path.join(root, "tests", "golden", "full_pin_2p24.json")) as f:
        pin = json.load(f)
    tr = c.synth_prepare(pin["n_linear"], pin["n_quad"], synth_seed=pin["synth_seed"], generated_at=pin["generated_at"])
    proof, info = c.synth_prove(tr)
    out["pin_2p24"] = hashlib.sha256(proof).hexdigest() == pin["proof_sha256"] and bytes(info.root).hex() == pin["root"]
    c.trace_destroy(tr); c.close()
    print(json.dumps(out))
''')


@pytest.mark.parametrize("layout", ["1", "0"])          # 1 = the default since round 6; 0 = the table-major layout of rounds 2-5 (kept as a knob)
def test_both_aes_table_layouts_give_the_same_stream_and_proofs(tmp_path, layout):
    script = tmp_path / "aes_layout_child.py"
    script.write_text(CHILD)
    p = subprocess.run([sys.executable, str(script), ROOT], env=dict(os.environ, LIG_AES_LAYOUT=layout), capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out and all(v is True for v in out.values()), out
