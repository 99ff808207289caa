#!/usr/bin/env python3
"""How much VALU issue time does one proof need?  From one rocprofv3 --pmc VALUBusy --kernel-trace pass (dispatches are
serialised under counter collection, so every duration is the kernel's stand-alone time): per kernel the summed duration, the
duration-weighted VALUBusy and their product = the SIMD-busy time the kernel needs; the total per proof is the time a proof
would take on a chip that issued a VALU instruction on every SIMD in every cycle.
    python tools/valu_budget.py <counter_collection.csv> <kernel_trace.csv> [proofs]      (default: the launches of k_merkle_top, one per proof)"""
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(?:lig::)?(k_[A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def main(counters, trace, proofs):
    dur = {}
    with open(trace) as f:
        for row in csv.DictReader(f):
            did = row.get("Dispatch_Id") or row.get("Correlation_Id")
            dur[did] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3        # us
    per = {}
    with open(counters) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != "VALUBusy":
                continue
            d = dur.get(row.get("Dispatch_Id") or row.get("Correlation_Id"))
            if d is None:
                continue
            k = per.setdefault(short(row["Kernel_Name"]), [0, 0.0, 0.0])
            k[0] += 1; k[1] += d; k[2] += d * float(row["Counter_Value"]) / 100.0
    if not proofs:
        proofs = per.get("k_merkle_top", [1])[0]
    out = {"note": "rocprofv3 --pmc VALUBusy --kernel-trace over full proofs (dispatches serialised: stand-alone durations); "
                   "busy_ms = sum over launches of duration x VALUBusy, per proof", "proofs": proofs, "kernels": {}}
    tot_d = tot_b = 0.0
    for k, (n, d, b) in sorted(per.items(), key=lambda kv: -kv[1][2]):
        out["kernels"][k] = {"launches_per_proof": round(n / proofs, 2), "standalone_ms_per_proof": round(d / proofs / 1e3, 4),
                             "valu_busy_pct": round(100.0 * b / d, 2) if d else 0.0, "busy_ms_per_proof": round(b / proofs / 1e3, 4)}
        tot_d += d; tot_b += b
    out["standalone_ms_per_proof"] = round(tot_d / proofs / 1e3, 3)
    out["valu_busy_ms_per_proof"] = round(tot_b / proofs / 1e3, 3)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
