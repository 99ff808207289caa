import sys, time, hashlib
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
c = amd.Context(8000, 8192, 32768)
rows, n = 2101, 32768
dc = c.malloc(rows * n * 32)
c.rng_fill(bytes(32), 0, dc, rows * n)
st = c.sha_state(n)
c.sync()
for rep in range(3):
    t = time.perf_counter()
    c.sha_update_rows(st, dc, rows)
    c.sync()
    print("sha_update_rows %d rows: %.3f ms" % (rows, 1e3 * (time.perf_counter() - t)))
c.close()
