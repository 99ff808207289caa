"""PCIe host->device rate of a 550 MB pinned witness matrix (one 2^24-constraint trace) on this box: idle, and while one / two
contexts prove resident traces back to back (the GPU saturated with the prover's kernels).  Says whether the H2D-inclusive
figure is bounded by the link itself or by how the upload is scheduled.  (tools/, not product)"""
import sys
import threading
import time

import torch

sys.path.insert(0, "tests")
import hip_lib

amd = hip_lib.load()
L_, K_, N_ = 8000, 8192, 32768
R = 2098
host = torch.empty((R, K_, 8), dtype=torch.int32, pin_memory=True)
host.fill_(7)
dev = torch.empty((R, K_, 8), dtype=torch.int32, device="cuda")
st = torch.cuda.Stream()
nbytes = host.numel() * 4


def upload_rate(reps=8):
    out = []
    with torch.cuda.stream(st):
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            dev.copy_(host, non_blocking=True)
            e1.record(st)
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
    return out


def fmt(ms):
    return "min %.2f / median %.2f / max %.2f ms = %.1f / %.1f / %.1f GB/s" % (
        min(ms), sorted(ms)[len(ms) // 2], max(ms), nbytes / min(ms) / 1e6, nbytes / sorted(ms)[len(ms) // 2] / 1e6, nbytes / max(ms) / 1e6)


print("idle link:            ", fmt(upload_rate()))
for n_ctx in (1, 2):
    ctxs = [amd.Context(L_, K_, N_) for _ in range(n_ctx)]
    traces = [c.synth_prepare(1 << 24, 0, synth_seed=1, generated_at=0) for c in ctxs]
    for c in ctxs:
        c.sync()
    stop = threading.Event()
    count = [0] * n_ctx

    def prove_loop(i):
        while not stop.is_set():
            ctxs[i].synth_prove(traces[i], copy=False)
            count[i] += 1

    th = [threading.Thread(target=prove_loop, args=(i,)) for i in range(n_ctx)]
    for t in th:
        t.start()
    time.sleep(0.3)
    c0, t0 = sum(count), time.perf_counter()
    ms = upload_rate(12)
    dt, proofs = time.perf_counter() - t0, sum(count) - c0
    stop.set()
    for t in th:
        t.join()
    print("%d context(s) proving: " % n_ctx, fmt(ms), "| meanwhile %.2f ms per resident proof" % (1e3 * dt / max(proofs, 1)))
    for c, t in zip(ctxs, traces):
        c.trace_destroy(t)
        c.close()
