#!/usr/bin/env python3
"""Whole-proof HBM traffic from two rocprofv3 counter passes of the same bench command (FETCH_SIZE and WRITE_SIZE, separate
passes with --kernel-trace only): every launch of every kernel is summed and divided by the number of proofs in the run
(= launches of k_sha_final) and by the committed rows of a proof.  FETCH_SIZE is doubled for gfx950 (microarch guide).
    python tools/pmc_whole_proof.py FETCH.csv WRITE.csv [rows_per_proof=2101]"""
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(?:lig::)?(k_[A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def load(path):
    per, n = {}, {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    return per, n


def main(fetch_csv, write_csv, rows=2101):
    fetch, nf = load(fetch_csv)
    write, nw = load(write_csv)
    proofs = nf.get("k_sha_final", 0)
    assert proofs and proofs == nw.get("k_sha_final", 0), "the two passes must run the same command"
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of one bench command; KiB counters; "
                   "FETCH_SIZE doubled for gfx950; summed over ALL launches of the run and divided by its %d proofs" % proofs,
           "proofs": proofs, "rows_per_proof": rows, "kernels": {}}
    tot = 0.0
    for k in sorted(set(fetch) | set(write)):
        b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / proofs
        tot += b
        out["kernels"][k] = {"launches_per_proof": nf.get(k, 0) / proofs, "MB_per_proof": b / 1e6, "bytes_per_row": b / rows}
    out["total_MB_per_proof"] = tot / 1e6
    out["total_bytes_per_committed_row"] = tot / rows
    out["algorithmic_bytes_per_row_survey_8d"] = 1054720
    out["ratio_to_algorithmic"] = tot / rows / 1054720
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2101)
