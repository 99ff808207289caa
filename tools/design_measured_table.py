#!/usr/bin/env python3
"""The measured-figures table of DESIGN.md section 6, generated from the JSON files committed under profiles/ (the
prose once quoted 638.5 us from a file that said 0.594 ms).  Every figure DESIGN.md states about speed or traffic lives in this table;
each row names the committed file and the path inside it, and tests/test_design_facts.py regenerates the block and compares.
  python tools/design_measured_table.py            prints the table
  python tools/design_measured_table.py --write     replaces the block between the measured-table markers in DESIGN.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- measured-table:begin -->", "<!-- measured-table:end -->"

# (what, file under profiles/, path into the JSON, unit / format)
BENCH = "r06_bench_default.json"
ROWS = [
    ("headline: constraints/s, two proofs in flight, witness resident", BENCH, "value", "e9"),
    ("time per step (two proofs)", BENCH, "ms_per_step", "ms"),
    ("one proof alone, wall", BENCH, "proof_wall_ms", "ms"),
    ("stage 1 / 2 / 3 of the last proof of the timed region (two in flight: each stage shares the chip)", BENCH, "config.stage_ms", "list_ms"),
    ("witness over PCIe (`value_incl_h2d`)", BENCH, "value_incl_h2d", "e9"),
    ("witness and caller randomness rows over PCIe", BENCH, "incl_h2d.caller_rands.value", "e9"),
    ("half of the constraints quadratic", BENCH, "quad_mix.value", "e9"),
    ("CPU restatement (`cpu_baseline`, kind port)", BENCH, "cpu_baseline.value", "e6"),
    ("cores it used", BENCH, "cpu_baseline.cores", "int"),
    ("dominant kernel: average launch (all launch sizes, HIP events in the run)", BENCH, "roofline.avg_launch_ms", "us"),
    ("dominant kernel: rows per launch (average)", BENCH, "roofline.rows_per_launch", "f1"),
    ("dominant kernel: 512-row launches alone", BENCH, "roofline.launches_of_512_rows.avg_launch_ms", "us"),
    ("dominant kernel: HBM fraction (`roofline.frac`)", BENCH, "roofline.frac", "f4"),
    ("... of the 512-row launches", BENCH, "roofline.launches_of_512_rows.frac", "f4"),
    ("... with one proof in flight", BENCH, "roofline.one_proof_in_flight.frac", "f4"),
    ("dominant kernel: fraction of the measured multiplier rate", BENCH, "roofline.valu_multiplier.frac", "f3"),
    ("chip-wide VALU issue slots in use", BENCH, "roofline.valu_issue.frac", "f3"),
    ("dominant kernel: HBM traffic per row (PMC)", "pmc_traffic.json", "k_encode_tiles<10, true>.hbm_bytes_per_row", "bytes"),
    ("whole proof: HBM traffic per committed row, planar (default)", "r06_zres0_whole_proof_traffic.json", "total_bytes_per_committed_row", "bytes"),
    ("whole proof: the same with `LIG_ZRES=1`", "r06_zres1_whole_proof_traffic.json", "total_bytes_per_committed_row", "bytes"),
]


def lookup(obj, path):
    for part in path.split("."):
        if isinstance(obj, dict) and part in obj:
            obj = obj[part]
            continue
        # keys that contain dots or commas ("k_encode_tiles<10, true>"): longest-prefix match
        hit = None
        for k in (obj if isinstance(obj, dict) else {}):
            if path.startswith(k):
                hit = k
        if hit is None:
            raise KeyError(path)
        rest = path[len(hit):].lstrip(".")
        return lookup(obj[hit], rest) if rest else obj[hit]
    return obj


def fmt(v, how):
    if how == "e9":
        return "%.3f × 10⁹" % (v / 1e9)
    if how == "e6":
        return "%.2f × 10⁶" % (v / 1e6)
    if how == "ms":
        return "%.2f ms" % v
    if how == "us":
        return "%.1f µs" % (v * 1e3)
    if how == "list_ms":
        return " / ".join("%.2f" % x for x in (v.values() if isinstance(v, dict) else v)) + " ms"
    if how == "int":
        return "%d" % v
    if how == "f1":
        return "%.1f" % v
    if how == "f3":
        return "%.3f" % v
    if how == "f4":
        return "%.4f" % v
    if how == "bytes":
        return "{:,} B".format(int(round(v)))
    raise ValueError(how)


def table():
    rows = ["| figure | value | `profiles/` file | JSON path |", "|---|---|---|---|"]
    cache = {}
    for what, fn, path, how in ROWS:
        if fn not in cache:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                cache[fn] = json.load(f)
        rows.append("| %s | %s | `%s` | `%s` |" % (what, fmt(lookup(cache[fn], path), how), fn, path))
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
