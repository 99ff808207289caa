#!/usr/bin/env python3
"""Where are the idle VALU issue slots of the benched configuration?  (a timeline, not another knob.)

Inputs:
  trace.csv    rocprofv3 --kernel-trace --output-format csv of the default bench command (two proofs in flight, dispatches NOT serialised:
               start / end of every launch as it ran next to the others);
  budget.json  tools/valu_budget.py of a --pmc VALUBusy pass of the same build: per kernel class the stand-alone VALUBusy (the share of
               SIMD issue slots the kernel fills when it has the chip to itself).

Method.  The steady-state window of the trace (between the first and the last launch of the timed proofs) is cut at every launch start /
end into slices; in a slice a fixed set of launches is running.  If the kernels of a slice filled slots independently, the slice would
use  u = min(1, sum over running launches of the class's stand-alone VALUBusy)  of the chip's slots -- an OPTIMISTIC bound for what that
mix of kernels can use (contention for LDS, the scheduler and HBM only lowers it).  So
  idle_attributable = sum over slices of (1 - u) * duration        idle slots that exist because of WHAT is running (or that nothing is),
                                                                   split by the running mix;
  idle_total        = window - (busy time of the PMC budget x proofs in the window)      what the counters say was really idle;
  idle_contention   = idle_total - idle_attributable               slots lost although the mix could have used them.
The table names, for every mix with more than 1 % of the window, its time, its bound u, and its share of the idle slots.

    python tools/issue_timeline.py trace.csv budget.json [first_fraction last_fraction]     (default window: the middle 60 % of the trace)
"""
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(?:lig::)?(k_[A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


GROUPS = [("K2 tiles (full)", ("k_encode_tiles<10, true>",)), ("K2 tiles (half)", ("k_encode_tiles<10, false>",)), ("column hash", ("k_sha_update_rows",)),
          ("sampler + RLC", ("k_rand_rlc", "k_rlc_partial", "k_rng_fill")), ("K1 / K3 / dot", ("k_encode_in", "k_encode_out")),
          ("small (single-row transforms, merkle, copies, combines)", ("",))]


def group_of(k):
    for g, pats in GROUPS:
        if any(k.startswith(p) for p in pats):
            return g
    return GROUPS[-1][0]


def main(trace, budget, f0=0.2, f1=0.8):
    bud = json.load(open(budget))
    busy = {k: v["valu_busy_pct"] / 100.0 for k, v in bud["kernels"].items()}
    ev = []
    with open(trace) as f:
        for row in csv.DictReader(f):
            ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"])))
    ev.sort()
    t_first, t_last = ev[0][0], max(e[1] for e in ev)
    w0, w1 = t_first + (t_last - t_first) * f0, t_first + (t_last - t_first) * f1
    points = []
    for s, e, k in ev:
        if e <= w0 or s >= w1:
            continue
        points.append((max(s, w0), 1, k)); points.append((min(e, w1), -1, k))
    points.sort(key=lambda p: (p[0], p[1]))
    running = {}
    mixes = {}
    t_prev = w0
    idle_attr = used_bound = 0.0
    for t, d, k in points:
        dt = t - t_prev
        if dt > 0:
            groups = sorted({group_of(x) for x, c in running.items() if c > 0})
            u = min(1.0, sum(busy.get(x, 0.0) * c for x, c in running.items() if c > 0))
            key = " + ".join(groups) if groups else "(no kernel running)"
            m = mixes.setdefault(key, [0.0, 0.0])
            m[0] += dt; m[1] += dt * (1.0 - u)
            idle_attr += dt * (1.0 - u); used_bound += dt * u
        running[k] = running.get(k, 0) + d
        t_prev = t
    window = w1 - w0
    merkle_tops = sum(1 for s, e, k in ev if k == "k_merkle_top" and w0 <= s < w1)
    proofs = max(merkle_tops, 1)
    busy_ms = bud["valu_busy_ms_per_proof"] * proofs
    idle_total = window / 1e6 - busy_ms
    out = {"window_ms": round(window / 1e6, 3), "proofs_in_window": proofs, "ms_per_proof": round(window / 1e6 / proofs, 3),
           "valu_busy_ms_per_proof_pmc": bud["valu_busy_ms_per_proof"], "issue_slots_used": round(busy_ms / (window / 1e6), 4),
           "idle_ms": round(idle_total, 3), "idle_attributable_to_the_running_mix_ms": round(idle_attr / 1e6, 3),
           "idle_lost_to_contention_ms": round(idle_total - idle_attr / 1e6, 3), "mixes": []}
    # how much longer every kernel class runs next to the others than alone (the PMC pass serialises dispatches: stand-alone durations):
    # the contention term, by class -- a class whose launches stretch is one whose waves wait for issue slots, LDS or memory that
    # another kernel is using
    conc = {}
    for s_, e_, k in ev:
        if e_ <= w0 or s_ >= w1:
            continue
        conc[group_of(k)] = conc.get(group_of(k), 0.0) + (min(e_, w1) - max(s_, w0))
    alone = {}
    for k, v in bud["kernels"].items():
        alone[group_of(k)] = alone.get(group_of(k), 0.0) + v["standalone_ms_per_proof"]
    out["stretch_by_class"] = [{"class": g, "standalone_ms_per_proof": round(alone.get(g, 0.0), 3), "concurrent_ms_per_proof": round(conc.get(g, 0.0) / 1e6 / proofs, 3),
                                "stretch": round(conc.get(g, 0.0) / 1e6 / proofs / alone[g], 2) if alone.get(g) else None} for g, _ in GROUPS]
    for key, (tm, idle) in sorted(mixes.items(), key=lambda kv: -kv[1][1]):
        if tm / window < 0.01:
            continue
        out["mixes"].append({"running": key, "time_pct_of_window": round(100 * tm / window, 2), "bound_u": round(1 - idle / tm, 3),
                             "idle_pct_of_window": round(100 * idle / window, 2), "share_of_idle_slots_pct": round(100 * (idle / 1e6) / idle_total, 1) if idle_total > 0 else None})
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(float(x) for x in sys.argv[3:5]))
