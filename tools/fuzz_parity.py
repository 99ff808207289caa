"""One-off differential fuzz: random trace shapes (and random batch programs) through the HIP prover and the oracle at
k = 512 (or LIG_FUZZ_K = 1024 / 2048 / 4096 / 8192); envelopes must be identical.  python tools/fuzz_parity.py [cases] [seed]"""
import os
import ctypes as C
import random
import sys
sys.path.insert(0, "tests")
import batch_prog
import hip_lib
import oracle_lib as ol

amd = hip_lib.load()
K_ = int(os.environ.get("LIG_FUZZ_K", "512"))
L_, N_ = K_ - 192, 4 * K_
SCALE = K_ // 512
P = ol.P


def random_program(rng):
    p = batch_prog.Program()
    live = []
    nslots = rng.randint(2, 12)
    for s in range(nslots):                      # initialise every slot so that later ops have defined operands
        if rng.random() < 0.5:
            p.set(s, [rng.randrange(1, P) for _ in range(rng.randint(0, 20))])
        else:
            p.set_scalar(s, rng.randrange(1, P))
        live.append(s)
    for _ in range(rng.randint(0, 25)):
        op = rng.choice(["add", "sub", "mul", "div", "copy", "const", "assert", "free_set"])
        a, b, o = rng.choice(live), rng.choice(live), rng.choice(live)
        if op == "add": p.add(o, a, b)
        elif op == "sub": p.sub(o, a, b)
        elif op == "mul": p.mul(o, a, b)
        elif op == "div": p.div(o, a, b)
        elif op == "copy": p.copy(o, a)
        elif op == "const": p.const(rng.choice(["ADD_CONST", "SUB_CONST", "CONST_SUB", "MUL_CONST", "MONTMUL_CONST"]), o, a, rng.randrange(P))
        elif op == "assert": p.assert_equal(a, b)          # usually false: the proof is then invalid in BOTH provers
        else:
            p.free(o); p.set_scalar(o, rng.randrange(P))
    if rng.random() < 0.3:
        base = nslots
        p.bit_decompose(list(range(base, base + 254)), rng.choice(live))
    return p


def main(cases, seed):
    rng = random.Random(seed)
    c = amd.Context(L_, K_, N_)
    for i in range(cases):
        n_lin = rng.choice([0, 1, rng.randint(0, 2000 * SCALE), rng.randint(0, 400000 * SCALE // (SCALE * SCALE) * SCALE)])
        n_quad = rng.choice([0, rng.randint(0, 1000 * SCALE), rng.randint(0, 120000 * SCALE // (SCALE * SCALE) * SCALE)])
        ts = rng.randint(0, 1 << 40)
        prog = random_program(rng) if rng.random() < 0.5 else None
        pub = [bytes(rng.randrange(256) for _ in range(rng.randint(0, 12))) for _ in range(rng.randint(0, 3))] if rng.random() < 0.5 else None
        oj = ol.make_job(L_, K_, N_, 192, n_lin, n_quad, generated_at=ts, threads=8, public_args=pub)
        hj = amd.Context.make_job(n_lin, n_quad, generated_at=ts, public_args=pub)
        if prog is not None:
            prog.attach(oj); prog.attach(hj)
        pr = ol.Proof()
        assert ol.lib().lo_prove(C.byref(oj), C.byref(pr)) == 0
        tr = c.synth_prepare_job(hj)
        proof, info = c.synth_prove(tr)
        c.trace_destroy(tr)
        want = bytes(pr.proof[:pr.proof_len])
        ok = proof == want and [info.valid_code, info.valid_linear, info.valid_quad] == [pr.valid_code, pr.valid_linear, pr.valid_quad]
        v = c.synth_verify(hj, bytes(info.const_sum), proof)
        valid = bool(info.valid_code and info.valid_linear and info.valid_quad)
        ok = ok and bool(v.accept) == valid
        ok = ok and bool(c.synth_verify(hj, None, proof).accept) == valid       # constant derived from the public statement
        if ok and info.rows <= 1500:                                             # the caller-rows entry on the same rows
            import numpy as np
            rows = ol.form_rows(oj)[0]
            kinds = ol.row_kinds(oj).copy()
            lib_pads = rng.random() < 0.5
            if lib_pads:
                draws = (kinds <= 3) | (kinds == 4)
                rows = rows.copy(); rows[draws, L_:] = 0
                kinds[draws] |= amd.ROW_DRAW_PAD
            tr, keep = c.rows_begin(kinds, rows, generated_at=ts, public_args=pub)
            _, seed1 = c.rows_commit(tr)
            rands, cs = ol.rand_rows(oj, seed1)
            proof_r, info_r = c.rows_prove(tr, rands, cs)
            c.trace_destroy(tr)
            ok = ok and proof_r == want
            vt, vseed, vi = c.rows_verify_begin(ol.row_kinds(oj), proof_r, public_args=pub)
            ok = ok and vt is not None and bool(c.rows_verify_finish(vt, rands, cs).accept) == valid
        print("case %3d lin %7d quad %7d batch %-5s rows %5d valid %s -> %s" % (i, n_lin, n_quad, prog is not None, info.rows, valid, "ok" if ok else "MISMATCH"), flush=True)
        ol.lib().lo_proof_free(C.byref(pr))
        if not ok:
            sys.exit(1)
    c.close()
    print("all", cases, "cases identical")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
