"""soak: two contexts proving the 2^24 bench trace concurrently, EVERY proof compared with the oracle pin (races between the
streams / events of a context would show up as a different envelope).  python tools/soak.py [proofs per thread]"""
import hashlib, json, sys, threading, ctypes as C
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pin = json.load(open("tests/golden/full_pin_2p24.json"))["proof_sha256"]
bad = []
def run(i):
    c = amd.Context(8000, 8192, 32768)
    tr = c.synth_prepare(1 << 24, 0, synth_seed=1, generated_at=0)
    for it in range(N):
        (addr, ln), info = c.synth_prove(tr, copy=False)
        h = hashlib.sha256(C.string_at(addr, ln)).hexdigest()
        if h != pin or not (info.valid_code and info.valid_linear and info.valid_quad):
            bad.append((i, it, h))
    c.trace_destroy(tr); c.close()
th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print("soak: %d proofs, %d mismatches" % (2 * N, len(bad)), bad[:3])
sys.exit(1 if bad else 0)
