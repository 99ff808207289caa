"""prove + verify one trace with a share of quadratic constraints, timed:  python tools/quad_probe.py LOG2 QUAD_PERCENT"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ligero_prover_amd", os.path.join(ROOT, "ligero-prover_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); sys.modules["ligero_prover_amd"] = pkg; spec.loader.exec_module(pkg)
lg, qp = int(sys.argv[1]), int(sys.argv[2])
nq = ((1 << lg) * qp) // 100
nl = (1 << lg) - nq
c = pkg.Context(8000, 8192, 32768)
t0 = time.time(); tr = c.synth_prepare(nl, nq, synth_seed=1, generated_at=0); c.sync(); print("prepare %.3f s" % (time.time() - t0), flush=True)
for i in range(2):
    t0 = time.time(); proof, info = c.synth_prove(tr); print("prove %.3f s  stages %.2f %.2f %.2f ms rows %d" % (time.time() - t0, info.ms_stage1, info.ms_stage2, info.ms_stage3, info.rows), flush=True)
c.trace_destroy(tr)
job = pkg.Context.make_job(nl, nq, synth_seed=1, generated_at=0)
t0 = time.time(); v = c.synth_verify(job, None, proof); print("verify %.3f s accept %d" % (time.time() - t0, v.accept), flush=True)
c.close()
