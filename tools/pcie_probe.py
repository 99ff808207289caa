"""Host -> device rate of the ABI's row upload path (lig_write from ordinary host memory), to state the PCIe-inclusive
figure DESIGN.md section 6 quotes: a 2^24-constraint trace is 2098 rows x 256 KiB."""
import sys, time
import numpy as np
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
c = amd.Context(8000, 8192, 32768)
rows, k = 2098, 8192
host = np.ones((rows, k, 8), dtype=np.uint32)
d = c.malloc(host.nbytes)
c.write(d, host); c.sync()
best = 1e9
for _ in range(3):
    t = time.perf_counter(); c.write(d, host); c.sync(); best = min(best, time.perf_counter() - t)
print("upload of %d rows (%.0f MB): %.2f ms = %.1f GB/s" % (rows, host.nbytes / 1e6, 1e3 * best, host.nbytes / best / 1e9))
c.close()
