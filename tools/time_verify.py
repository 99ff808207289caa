import sys, time
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
c = amd.Context(8000, 8192, 32768)
job = amd.Context.make_job(1 << 24, 0)
tr = c.synth_prepare_job(job)
proof, info = c.synth_prove(tr)
for i in range(3):
    v = c.synth_verify(job, None, proof); print("derive", v.accept, v.ms_total)
for i in range(3):
    v = c.synth_verify(job, bytes(info.const_sum), proof); print("given", v.accept, v.ms_total)
