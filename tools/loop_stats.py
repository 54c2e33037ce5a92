#!/usr/bin/env python
"""Static statistics of the evaluation loop of a kernel from the compiler's assembly (`hipcc -S`): instruction
classes inside the blocks LLVM marks `in Loop: Header=<largest loop>`, plus the kernel's register / scratch
footprint.  CPU-side companion of the rocprofv3 counters: the loop of the step kernels is straight-line code with
a handful of uniform branches, so its static instruction count tracks SQ_INSTS_VALU per evaluation.

    python tools/loop_stats.py file.s [kernel substring]
"""
import collections
import re
import subprocess
import sys


def functions(text):
    for m in re.finditer(r"; -- Begin function (\S+)\n(.*?); -- End function", text, re.S):
        yield m.group(1), m.group(2)


def classify(op):
    if op.startswith("v_accvgpr") : return "accvgpr"
    if op.endswith("_dpp") or "dpp" in op: return "dpp"
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "fma64"
    if op.startswith("v_mul_f64"): return "mul64"
    if op.startswith("v_add_f64"): return "add64"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def loop_blocks(body):
    """{header: [lines]} for every loop header mentioned in block comments"""
    loops = collections.defaultdict(list)
    cur = None
    for line in body.splitlines():
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", line)
        if m:
            cur = None
            c = m.group(2) or ""
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", c)
            if h: cur = h.group(1)
            h = re.search(r"=>This .*Loop Header", c)
            if h: cur = m.group(1)[2:]
            continue
        if cur is not None:
            loops[cur].append(line)
    return loops


def stats(lines):
    c = collections.Counter()
    for line in lines:
        m = re.match(r"^\t([a-z_0-9]+)\b", line)
        if m and not line.startswith("\t."):
            op = m.group(1)
            c[classify(op)] += 1
            if "dpp" in line and classify(op) != "dpp":
                c["dpp"] += 1; c[classify(op)] -= 1
    return c


def main():
    text = open(sys.argv[1]).read()
    like = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in functions(text):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if like not in dem:
            continue
        print("#", dem[:150])
        for k in ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            m = re.search(rf"{re.escape(name)}.*?\.{k}:\s+(\d+)", text, re.S)
        loops = loop_blocks(body)
        tot = stats(body.splitlines())
        print("  whole kernel:", sum(tot.values()), dict(tot))
        for h, lines in sorted(loops.items(), key=lambda kv: -len(kv[1]))[:int(__import__("os").environ.get("LOOPS", "3"))]:
            c = stats(lines)
            valu = sum(v for k, v in c.items() if k in ("fma64", "mul64", "add64", "valu_other", "dpp", "accvgpr"))
            print(f"  loop {h}: {sum(c.values())} instr, VALU {valu}: {dict(c)}")
    # kernel descriptors (metadata at the end of the file)
    for blk in text.split("- .agpr_count:")[1:]:
        get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        nm = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
        if like in nm:
            print(f"  {nm[:60]}: vgpr {get('vgpr_count')} agpr {blk.split()[0]} vspill {get('vgpr_spill_count')} "
                  f"sspill {get('sgpr_spill_count')} scratch {get('private_segment_fixed_size')} lds {get('group_segment_fixed_size')}")


if __name__ == "__main__":
    main()
