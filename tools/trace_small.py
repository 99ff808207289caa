"""LIG_TRACE timeline of one proof (synchronised phase marks, stderr) for a given trace size: python tools/trace_small.py [log2]"""
import os, sys
os.environ["LIG_TRACE"] = "1"
sys.path.insert(0, "tests")
import hip_lib
amd = hip_lib.load()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = amd.Context(8000, 8192, 32768)
tr = c.synth_prepare(1 << lg, 0)
for i in range(3):
    print("--- proof", i, file=sys.stderr)
    proof, info = c.synth_prove(tr)
    print("stage ms: %.3f %.3f %.3f total %.3f" % (info.ms_stage1, info.ms_stage2, info.ms_stage3, info.ms_total), file=sys.stderr)
