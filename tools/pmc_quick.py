#!/usr/bin/env python
"""Medians of the PMC counters of one kernel over the rocpd databases below a directory (one rocprofv3 --pmc pass
each), for quick A/B runs on the GPU box:   python tools/pmc_quick.py <dir> <kernel substring>"""
import glob
import json
import os
import sqlite3
import statistics
import sys


def main():
    src, like = sys.argv[1], "%" + sys.argv[2] + "%"
    out = {}
    for p in sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)):
        d = sqlite3.connect(p)
        try:
            durs = [x for (x,) in d.execute("select duration from kernels where name like ?", (like,))]
        except sqlite3.Error:
            continue
        if not durs:
            continue
        med = statistics.median(durs)
        out.setdefault("median_ns", []).append(med)
        vals = {}
        try:
            for cname, value, dur in d.execute(
                    "select counter_name, value, duration from counters_collection where kernel_name like ?", (like,)):
                if dur > 0.6 * med:
                    vals.setdefault(cname, []).append(value)
        except sqlite3.Error:
            pass
        for k, v in vals.items():
            out[k] = statistics.median(v)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
