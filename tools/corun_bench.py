"""Co-scheduling experiment: encode (VALU-bound, barrier-coupled workgroups) next to the column hash (serial chain per
column, 512 waves).  Two contexts = two independent sets of streams on the same GPU.
  python tools/corun_bench.py            (LIG_SHA_BLOCK=64|256|512 selects the hash workgroup size)"""
import sys, time
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
l, k, n = 8000, 8192, 32768
rows = 512
A = amd.Context(l, k, n)
B = amd.Context(l, k, n)
msgs = A.malloc(rows * k * 32)
A.rng_fill(bytes(32), 0, msgs, rows * k)
cwA = A.malloc(rows * n * 32)
cwB = B.malloc(rows * n * 32)
B.rng_fill(bytes(32), 7, cwB, rows * n)
st = B.sha_state(n)
A.sync(); B.sync()

def t(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); A.sync(); B.sync()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best

enc = lambda: A.encode_rows(msgs, cwA, rows)
sha = lambda: B.sha_update_rows(st, cwB, rows)
both = lambda: (sha(), enc())
for _ in range(2): both(); A.sync(); B.sync()
print("encode %d rows alone : %.3f ms" % (rows, t(enc)))
print("hash   %d rows alone : %.3f ms" % (rows, t(sha)))
print("both concurrently    : %.3f ms" % t(both))
def staged():
    best = (1e9, 1e9)
    for _ in range(5):
        t0 = time.perf_counter(); sha(); enc(); A.sync(); t1 = time.perf_counter(); B.sync(); t2 = time.perf_counter()
        if t2 - t0 < best[1]: best = (t1 - t0, t2 - t0)
    return 1e3 * best[0], 1e3 * best[1]
print("hash first, then encode: encode done at %.3f ms, all done at %.3f ms" % staged())
enc2 = lambda: (A.encode_rows(msgs, cwA, rows), A.encode_rows(msgs, cwA, rows))
print("encode x2 alone      : %.3f ms" % t(enc2))
both2 = lambda: (sha(), enc2())
print("hash + encode x2     : %.3f ms" % t(both2))
A.close(); B.close()
