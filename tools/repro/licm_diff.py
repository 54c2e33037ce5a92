#!/usr/bin/env python
"""Which machine instructions did the pre-RA MachineLICM pass move?  Input: the stderr of
    hipcc ... -mllvm -print-before=early-machinelicm -mllvm -print-after=early-machinelicm -mllvm -filter-print-funcs=<kernel>
Output: per run of the pass, the moved instructions grouped by opcode (instruction text is matched by its defined virtual
register, which the pass keeps) with source block -> destination block, and every moved instruction that is convergent,
reads or writes EXEC / lanes of another thread, or touches memory."""
import collections
import re
import sys


def parse(lines):
    where, text = {}, {}
    bb = None
    for ln in lines:
        m = re.match(r"^(bb\.\d+)", ln)
        if m:
            bb = m.group(1)
            continue
        m = re.match(r"^\s+(%\d+)(?::\w+)?(?:\.\w+)? = (.*)$", ln) or re.match(r"^\s+(?:early-clobber )?(%\d+):\S+ = (.*)$", ln)
        if m and bb:
            where[m.group(1)] = bb
            text[m.group(1)] = m.group(2)
    return where, text


def main():
    lines = open(sys.argv[1]).read().splitlines()
    marks = [i for i, ln in enumerate(lines) if ln.startswith("# *** IR Dump")]
    marks.append(len(lines))
    runs = [(marks[i], marks[i + 1], marks[i + 2]) for i in range(0, len(marks) - 2, 2)]
    for n, (a, b, c) in enumerate(runs):
        wb, tb = parse(lines[a:b])
        wa, ta = parse(lines[b:c])
        moved = [(v, wb[v], wa[v], ta[v]) for v in wa if v in wb and wa[v] != wb[v]]
        ops = collections.Counter(re.split(r"[ (]", t[3].replace("nofpexcept ", "").replace("contract ", ""))[0] for t in moved)
        print(f"== run {n}: {len(moved)} instructions moved")
        for op, k in ops.most_common():
            print(f"   {k:5d}  {op}")
        dst = collections.Counter((t[1], t[2]) for t in moved)
        for (s, d), k in dst.most_common(8):
            print(f"   {k:5d}  {s} -> {d}")
        sus = [t for t in moved if re.search(r"dpp|DPP|READLANE|READFIRSTLANE|WRITELANE|PERMUTE|exec|EXEC|LOAD|load|DS_|BUFFER|GLOBAL|SCRATCH|convergent|INLINEASM", t[3])]
        print(f"   suspicious: {len(sus)}")
        for t in sus[:60]:
            print(f"      {t[0]} {t[1]} -> {t[2]}: {t[3][:150]}")


if __name__ == "__main__":
    main()
