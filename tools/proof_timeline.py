#!/usr/bin/env python3
"""Timeline of ONE proof out of a rocprofv3 rocpd database of `bench.py --inflight 1`: every launch between two consecutive
k_sha_init launches (= one proof), start offset / duration / queue, consecutive launches of the same kernel on the same queue merged,
plus the gaps in which NO kernel of the proof runs (host phases).   python tools/proof_timeline.py x.db [proof-index]"""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("lig::", "")


def main(path, which=10):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(db.execute("select name, start, end, %s from kernels order by start" % qcol))
    inits = [i for i, r in enumerate(rows) if "k_sha_init" in r[0]]
    a, b = inits[which], inits[which + 1]
    # a proof starts a little before its k_sha_init (pads, masks): take everything from the previous proof's last launch on
    prev_end = max(r[2] for r in rows[:a]) if a else rows[0][1]
    seg = [r for r in rows if r[1] >= prev_end - 1 and r[1] < rows[b][1]]
    seg = [r for r in seg if r[1] >= rows[a][1] - 400000]
    t0 = seg[0][1]
    merged = []
    for name, s, e, q in seg:
        n = short(name)
        if merged and merged[-1][0] == n and merged[-1][4] == q and s - merged[-1][2] < 30000:
            merged[-1][2] = e; merged[-1][3] += 1; merged[-1][5] += e - s
        else:
            merged.append([n, s, e, 1, q, e - s])
    print("| start ms | end ms | busy ms | launches | queue | kernel |")
    print("|---|---|---|---|---|---|")
    for n, s, e, cnt, q, busy in merged:
        if e - s > 20000:
            print("| %.3f | %.3f | %.3f | %d | %s | `%s` |" % ((s - t0) / 1e6, (e - t0) / 1e6, busy / 1e6, cnt, q, n))
    # gaps with nothing running
    ev = sorted((s, e) for _, s, e, _ in seg)
    cur = ev[0][1]
    gaps = []
    for s, e in ev[1:]:
        if s > cur + 20000:
            gaps.append((cur, s))
        cur = max(cur, e)
    print()
    print("nothing of the proof on the GPU (> 20 us):")
    for s, e in gaps:
        print("  %.3f .. %.3f ms  (%.0f us)" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3))
    print("proof span %.3f ms, idle total %.3f ms" % ((cur - t0) / 1e6, sum(e - s for s, e in gaps) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
