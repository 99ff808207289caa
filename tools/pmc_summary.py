#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 --pmc pass (CSV of one counter): for every kernel the average counter value over
its launches, and separately over its largest launches (the steady-state chunk), as JSON.
    python tools/pmc_summary.py pmc_counter_collection.csv"""
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(?:lig::)?(k_[A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def main(path):
    per = {}
    counter = None
    with open(path) as f:
        for row in csv.DictReader(f):
            counter = row.get("Counter_Name", counter)
            per.setdefault(short(row["Kernel_Name"]), []).append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    out = {"counter": counter, "kernels": {}}
    for k, v in sorted(per.items()):
        gmax = max(g for g, _ in v)
        big = [x for g, x in v if g == gmax]
        out["kernels"][k] = {"launches": len(v), "avg": sum(x for _, x in v) / len(v), "largest_grid": gmax,
                             "launches_at_largest_grid": len(big), "avg_at_largest_grid": sum(big) / len(big)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
