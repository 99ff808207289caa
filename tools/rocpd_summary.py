#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`) as markdown:
per-kernel calls, average / min / max / total duration, and -- for one kernel -- the same split by grid size, which
is what bench.py's in-run HIP-event average of the dominant kernel has to be compared with (a proof launches it on
chunks of 512, 512, 512, 466, 96 rows, plus single-row encodes that bench.py does not bracket).
    python tools/rocpd_summary.py x.db [kernel-substring-for-the-grid-table]"""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main(path, by_grid=None):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                           "from kernels group by name order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, avg, mn, mx, tot in rows:
        print("| `%s` | %d | %.2f | %.2f | %.2f | %.3f | %.1f |" % (short(name), n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e6, 100.0 * tot / total))
    for by_grid in (by_grid or "").split(";"):
        if not by_grid:
            continue
        print()
        print("`%s` by launch size (workgroups):" % by_grid)
        print()
        print("| workgroups | calls | avg us | min us | max us |")
        print("|---|---|---|---|---|")
        q = ("select grid_x / workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
             "where name like ? group by 1 order by 1 desc")
        for g, n, avg, mn, mx in db.execute(q, ("%" + by_grid + "%",)):
            print("| %d | %d | %.2f | %.2f | %.2f |" % (g, n, avg / 1e3, mn / 1e3, mx / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
