#!/usr/bin/env python
"""Basic blocks of a kernel in `hipcc -S` output with instruction / scratch / store / VALU counts (where do the spills sit?):
    python tools/asm_blocks.py file.s [min instructions]"""
import re
import sys


def main():
    t = open(sys.argv[1]).read()
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for m in re.finditer(r"; -- Begin function (\S+)\n(.*?); -- End function", t, re.S):
        print("#", m.group(1)[:90])
        blocks, cur = [], None
        for ln in m.group(2).splitlines():
            mm = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
            if mm:
                cur = [mm.group(1), mm.group(2) or "", 0, 0, 0, 0]
                blocks.append(cur)
                continue
            if cur is None:
                cur = ["entry", "", 0, 0, 0, 0]
                blocks.append(cur)
            im = re.match(r"^\t([a-z_0-9]+)\b", ln)
            if im and not ln.startswith("\t."):
                cur[2] += 1
                op = im.group(1)
                cur[3] += op.startswith("scratch_")
                cur[4] += op.startswith("global_store")
                cur[5] += op.startswith("v_")
        for b in blocks:
            if b[2] >= lim or b[3] > 0:
                print(f"  {b[0]:12s} {b[2]:5d} scratch {b[3]:3d} gstore {b[4]:3d} valu {b[5]:5d} {b[1][:60]}")


if __name__ == "__main__":
    main()
