import ctypes as C, sys, os
sys.path.insert(0, "tests")
import hip_lib, oracle_lib as ol
import numpy as np
amd = hip_lib.load()
for (l,k,n,nl,nq) in [(8000,8192,32768,24123,8005),(8000,8192,32768,24123,0),(8000,8192,32768,0,8005),(8000,8192,32768,8000*3,0),(320,512,2048,162880,320),(320,512,2048,162880,0),(320,512,2048,320*400,0),(320,512,2048,320*509,0)]:
    c = amd.Context(l,k,n)
    tr = c.synth_prepare(nl,nq,generated_at=5)
    proof, info = c.synth_prove(tr)
    job = ol.make_job(l,k,n,192,nl,nq,generated_at=5,threads=4)
    pr = ol.Proof(); assert ol.lib().lo_prove(C.byref(job), C.byref(pr))==0
    want = bytes(pr.proof[:pr.proof_len])
    print((l,k,nl,nq), "proof equal", proof==want, "root equal", bytes(info.root)==bytes(pr.root), "valid", info.valid_code, info.valid_linear, info.valid_quad, "seed2 equal", bytes(info.stage2_seed)==bytes(pr.stage2_seed))
    c.trace_destroy(tr); c.close()
