#!/usr/bin/env python3
"""Regression pins of whole proofs: SHA-256 of the envelope, root, seeds and linear constant for a few small jobs,
produced by the build's OWN oracle (oracle/liblig_oracle.so).  They pin today's agreed behaviour of oracle and HIP
prover against accidental drift in later rounds; they are NOT reference vectors (the reference holds none and cannot
run here -- parity with it stays "unpinned" for proof bytes).

  python tests/golden/make_proof_pins.py        (writes tests/golden/proof_pins.json)
"""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol          # noqa: E402
import test_batch_rows as tb     # noqa: E402   (demo_program)

JOBS = [
    dict(l=320, k=512, n=2048, n_linear=700, n_quad=0, generated_at=1, batch=None),
    dict(l=320, k=512, n=2048, n_linear=640, n_quad=330, generated_at=2, batch=None),
    dict(l=320, k=512, n=2048, n_linear=0, n_quad=0, generated_at=0, batch=None),
    dict(l=832, k=1024, n=4096, n_linear=2000, n_quad=900, generated_at=3, batch=None),
    dict(l=320, k=512, n=2048, n_linear=100, n_quad=0, generated_at=5, batch="demo"),
    dict(l=320, k=512, n=2048, n_linear=100, n_quad=0, generated_at=5, batch="demo_no_bits"),
    # the two slicing semantics of buffer_view (include/lig_hip.h, LIG_BOP_UPSTREAM_COMPAT): declared / as upstream defines it
    dict(l=320, k=512, n=2048, n_linear=100, n_quad=0, generated_at=5, batch="slicing_declared"),
    dict(l=320, k=512, n=2048, n_linear=100, n_quad=0, generated_at=5, batch="slicing_upstream"),
]


def run(job):
    j = ol.make_job(job["l"], job["k"], job["n"], 192, job["n_linear"], job["n_quad"], generated_at=job["generated_at"], threads=4)
    if job["batch"]:
        tb.pin_program(job["batch"]).attach(j)
    pr = ol.Proof()
    assert ol.lib().lo_prove(C.byref(j), C.byref(pr)) == 0
    out = dict(job, rows=pr.rows, proof_len=pr.proof_len, proof_sha256=hashlib.sha256(bytes(pr.proof[:pr.proof_len])).hexdigest(),
               root=bytes(pr.root).hex(), stage1_seed=bytes(pr.stage1_seed).hex(), stage2_seed=bytes(pr.stage2_seed).hex(),
               const_sum=bytes(pr.const_sum).hex(), valid=[pr.valid_code, pr.valid_linear, pr.valid_quad])
    ol.lib().lo_proof_free(C.byref(pr))
    return out


if __name__ == "__main__":
    pins = {"note": "regression pins from the build's own oracle, not reference vectors", "jobs": [run(j) for j in JOBS]}
    with open(os.path.join(HERE, "proof_pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
    print("wrote", len(pins["jobs"]), "pins")
