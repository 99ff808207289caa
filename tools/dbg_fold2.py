import ctypes as C, sys, os, hashlib
sys.path.insert(0, "tests")
import hip_lib, oracle_lib as ol
import numpy as np
amd = hip_lib.load()
rng = np.random.default_rng(3)
for (l,k,n,rows) in [(320,512,2048,600),(320,512,2048,130),(320,512,2048,300),(8000,8192,32768,40)]:
    c = amd.Context(l,k,n)
    msgs = np.stack([ol.rand_field(rng, k) for _ in range(rows)])
    dm, dc = c.upload(msgs), c.malloc(32*n*rows)
    c.encode_rows(dm, dc, rows)
    got = c.download(dc, (rows, n, 8))
    o = ol.Ctx(l,k,n)
    want = o.encode_rows(msgs, threads=8)
    bad = [r for r in range(rows) if not np.array_equal(got[r], want[r])]
    print((k, rows), "bad rows", len(bad), bad[:12])
    if bad:
        r = bad[0]
        d = np.nonzero((got[r] != want[r]).any(axis=1))[0]
        print("  row", r, "bad elements", len(d), d[:16], "cosets", sorted(set(int(x) & 3 for x in d))[:4])
    c.close()
