#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 `--kernel-trace` run from its rocpd database(s):
    python tools/rocpd_stats.py <dir-or-db> [top N]  ->  name, launches, average us, total ms (CSV on stdout)"""
import glob
import os
import sqlite3
import sys


def main():
    path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dbs = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    print("kernel,launches,avg_us,total_ms")
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for name, n, avg, tot in cur.execute("select name, count(*), avg(end - start) / 1e3, sum(end - start) / 1e6 from kernels "
                                             "group by name order by 4 desc limit ?", (top,)):
            print(f"\"{name[:110]}\",{n},{avg:.1f},{tot:.2f}")


if __name__ == "__main__":
    main()
