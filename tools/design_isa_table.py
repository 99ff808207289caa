#!/usr/bin/env python3
"""The ISA-facts table of DESIGN.md section 4.1, generated from the built library (tools/kernel_facts.py).
  python tools/design_isa_table.py            prints the table
  python tools/design_isa_table.py --write     replaces the block between the isa-table markers in DESIGN.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_facts import ROOT, kernel_facts           # noqa: E402

KERNELS = ["k_encode_in<10>", "k_encode_tiles<10, true>", "k_encode_tiles<10, false>", "k_encode_out<10, 2>", "k_encode_out<10, 0>",
           "k_encode_out_dot<10>",
           "k_encode_out_dot_z<10>", "k_gather_rows_z<10>", "k_sha_update_rows", "k_sha_update_rows_ws<2>", "k_sha_update_rows_z<10, 2>", "k_rand_rlc<4, 0>", "k_rand_rlc<4, 1>", "k_rng_fill_rows_dense<4, 0>", "k_rlc_partial",
           "k_quad_rows<EvenOfView>", "k_merkle_level", "k_tiled_pass1<7, false>", "k_tiled_pass2<8>", "k_div_batched<4>"]
BEGIN, END = "<!-- isa-table:begin -->", "<!-- isa-table:end -->"


def table(lib=None):
    f = kernel_facts(lib)
    rows = ["| kernel | VGPRs | SGPRs | LDS B | scratch B | instructions | `v_mad_u64_u32` | `s_barrier` | `v_mfma` |", "|---|---|---|---|---|---|---|---|---|"]
    for k in KERNELS:
        v = f[k]
        rows.append("| `%s` | %d | %d | %d | %d | %d | %d | %d | %d |" % (k, v["vgprs"], v["sgprs"], v["lds_bytes"], v["scratch_bytes"], v["instructions"],
                                                                 v["v_mad_u64_u32"], v["s_barrier"], v["v_mfma"]))
    total_mfma = sum(v["v_mfma"] for v in f.values())
    rows.append("")
    rows.append("`v_mfma` in the whole library (%d kernels): %d." % (len(f), total_mfma))
    return "\n".join(rows)


if __name__ == "__main__":
    t = table()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
        open(p, "w").write(s[:a] + "\n" + t + "\n" + s[b:])
    else:
        print(t)
