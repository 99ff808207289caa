"""Every shape of the column hash (shader/sha256.wgsl:148-228; engine.cpp:1514-1686) against the CPU oracle, for every value of
LIG_SHA_WS: 0 = one wave per 64 columns (rounds 1-4, the documented fallback), 1 / 2 / 4 = the wave-specialised kernel (producer +
consumer wave per 64 columns, 1 / 2 / 4 groups per workgroup; 2 is the default).  The knob is read once per process, so every value
runs in a child process.  Covered per value: instance counts that are not a multiple of 64 (ragged last group, a lone
lane, groups missing from the last workgroup), odd and even row counts, and every parity of the rows absorbed before a call
(the pending half block: one SHA-256 block = two 32-byte rows) including one-row calls that only fill or only flush it."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import ctypes as C, json, os, sys
    import numpy as np
    root = sys.argv[1]
    sys.path.insert(0, os.path.join(root, "tests"))
    import hip_lib, oracle_lib as ol
    amd = hip_lib.load()
    c = amd.Context(320, 512, 2048)
    rng = np.random.default_rng(11)
    bad = []
    n_cases = 0
    # (instances, splits of the row sequence into lig_sha_update_rows calls)
    shapes = [1, 37, 63, 64, 65, 127, 128, 129, 192, 200, 257, 2048, 2049]
    splits = [[1], [2], [3], [1, 1], [1, 2], [2, 1], [1, 1, 1], [5, 4], [4, 5], [1, 7, 1, 2], [2, 2, 2, 1], [9], [16], [3, 1, 1, 1, 3]]
    for ninst in shapes:
        for split in splits:
            rows = sum(split)
            data = np.stack([ol.rand_field(rng, ninst) for _ in range(rows)])
            dd, st, dl = c.upload(data), c.sha_state(ninst), c.malloc(32 * ninst)
            off = 0
            for cnt in split:
                c.sha_update_rows(st, C.c_void_p(dd.value + off * 32 * ninst), cnt)
                off += cnt
            c.sha_final(st, dl)
            got = c.download(dl, (ninst, 32), dtype=np.uint8)
            n_cases += 1
            if not np.array_equal(got, ol.colsha(data)):
                bad.append([ninst, split])
            for p in (dd, st, dl):
                c.free(p)
    c.close()
    print(json.dumps(dict(cases=n_cases, bad=bad)))
''')


@pytest.mark.parametrize("ws", ["0", "1", "2", "4"])
def test_column_hash_shapes_for_every_kernel_variant(tmp_path, ws):
    script = tmp_path / "sha_ws_child.py"
    script.write_text(CHILD)
    p = subprocess.run([sys.executable, str(script), ROOT], env=dict(os.environ, LIG_SHA_WS=ws), capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["cases"] == 13 * 14 and out["bad"] == [], out
