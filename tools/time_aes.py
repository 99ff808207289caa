"""stand-alone timing of the dense AES-CTR row sampler (2098 rows x 8000 elements, the stage-2 randomness rows of a 2^24 proof)"""
import sys, time
import numpy as np
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
c = amd.Context(8000, 8192, 32768)
R = 2098
per_row = np.full(R, 8000, dtype=np.uint32)
d = c.malloc(R * 8192 * 32)
key = bytes(range(32))
for _ in range(3):
    c.rng_fill_rows(key, 0, per_row, d); c.sync()
t0 = time.perf_counter()
for _ in range(10):
    c.rng_fill_rows(key, 0, per_row, d)
c.sync()
print("dense fill of %d rows: %.3f ms" % (R, 1e3 * (time.perf_counter() - t0) / 10))
