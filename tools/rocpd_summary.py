#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`) as a
markdown table: per-kernel calls, average / min / max / total duration.   python tools/rocpd_summary.py x.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                           "from kernels group by name order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, avg, mn, mx, tot in rows:
        short = name.split("(")[0].replace("void ", "")
        print("| `%s` | %d | %.2f | %.2f | %.2f | %.3f | %.1f |" % (short, n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e6, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
