#!/usr/bin/env python3
"""Pins the EXACT bench.py job (configs[2]: 2^24 linear constraints, k=8192, synthetic seed 1, encoding seed 0..31,
generated_at 0) with the oracle's reference-structured prover (one executor call per row, every row re-encoded in each
of the three stages): envelope SHA-256, length, root, both seeds and the linear constant go to
tests/golden/full_pin_2p24.json, which `-m gpu` tests and bench.py compare the HIP prover's proof against.

About 20 minutes on 8 cores (2101 rows x 3 stages of 1 MiB radix-2 transforms).  Like proof_pins.json this is the
build's OWN oracle, not a reference vector (the reference cannot run here).

  python tests/golden/make_full_pin.py [log2_constraints=24] [threads=all] [quad_percent=0]

quad_percent = Q > 0: the quadratic mix of SURVEY.md 8(d) -- Q % of the 2^log2 constraints are quadratic (x*y = z slots: three
committed rows per 8000 of them and no randomness row, nonbatch_context.hpp:771-780), the rest linear; the pin goes to
full_pin_2p<log2>_q<Q>.json (bench.py --quad-percent Q).
"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol          # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    qp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    n_quad = ((1 << lg) * qp) // 100
    n_lin = (1 << lg) - n_quad
    j = ol.make_job(8000, 8192, 32768, 192, n_lin, n_quad, synth_seed=1, generated_at=0, threads=threads)
    pr = ol.Proof()
    t0 = time.time()
    assert ol.lib().lo_prove(C.byref(j), C.byref(pr)) == 0
    proof = bytes(pr.proof[:pr.proof_len])
    out = dict(l=8000, k=8192, n=32768, n_linear=n_lin, n_quad=n_quad, synth_seed=1, generated_at=0, rows=pr.rows,
               proof_len=pr.proof_len, proof_sha256=hashlib.sha256(proof).hexdigest(), root=bytes(pr.root).hex(),
               stage1_seed=bytes(pr.stage1_seed).hex(), stage2_seed=bytes(pr.stage2_seed).hex(),
               const_sum=bytes(pr.const_sum).hex(), valid=[pr.valid_code, pr.valid_linear, pr.valid_quad],
               sample_idx=[int(pr.sample_idx[i]) for i in range(192)],
               oracle_seconds=round(time.time() - t0, 1), oracle_threads=threads,
               note="oracle/liblig_oracle.so lo_prove on the bench.py job; the build's own oracle, not a reference vector")
    ol.lib().lo_proof_free(C.byref(pr))
    name = "full_pin_2p%d%s.json" % (lg, "_q%d" % qp if qp else "")
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", name, out["proof_sha256"], "in", out["oracle_seconds"], "s")


if __name__ == "__main__":
    main()
