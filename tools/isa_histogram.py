#!/usr/bin/env python
"""Opcode histogram of one kernel of a topology library, from the compiler's assembly
(hipcc --cuda-device-only -S of jm_lib.cpp for that topology).

    python tools/isa_histogram.py <model> <kernel substring> [--flags ...]  > profiles/<name>.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_asm(model_name: str, kernel_like: str, extra=(), source="jm_lib.cpp", defines=("-DJM_SPLIT_CONSTRAINT",)):
    from jiminy_amd import codegen, load_builtin
    try:
        model = load_builtin(model_name)
    except LookupError:
        from tests import robots
        model = getattr(robots, model_name)()
    hdr = codegen.write_header(model)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [codegen.HIPCC, f"--offload-arch={codegen.OFFLOAD_ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
               f"-DJM_TOPO_HEADER=\"{hdr}\"", "-Wno-unused-value", "-ffp-contract=fast", *defines, *extra,
               "--cuda-device-only", "-S", os.path.join(codegen.CSRC, source), "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    # split into functions ("; -- Begin function <mangled>" ... "; -- End function")
    for m in re.finditer(r"; -- Begin function (\S+)\n(.*?); -- End function", text, re.S):
        name, body = m.group(1), m.group(2)
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if kernel_like in demangled:
            return demangled, body
    raise LookupError(kernel_like)


def histogram(asm: str):
    ops = collections.Counter()
    for line in asm.splitlines():
        m = re.match(r"^\t([a-z_0-9]+)\b", line)
        if m and not line.startswith("\t."):
            ops[m.group(1)] += 1
    return ops


def main():
    model, like = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    source, defines = "jm_lib.cpp", ("-DJM_SPLIT_CONSTRAINT",)
    if "k_quad_con" in like:
        source, defines = "jm_lib_constraint.cpp", ("-DJM_CON_PART=2",)
    name, asm = kernel_asm(model, like, extra, source, defines)
    ops = histogram(asm)
    total = sum(ops.values())
    print(f"# {name}\n# flags: {' '.join(extra) or '(default)'}\n# instructions: {total}")
    groups = {
        "fma (v_fma/v_fmac f64)": sum(v for k, v in ops.items() if k.startswith(("v_fma_f64", "v_fmac_f64"))),
        "unfused mul/add f64": sum(v for k, v in ops.items() if k.startswith(("v_mul_f64", "v_add_f64"))),
        "v_accvgpr_*": sum(v for k, v in ops.items() if k.startswith("v_accvgpr")),
        "dpp movs": sum(v for k, v in ops.items() if k.endswith("_dpp")),
        "div sequence (v_div_scale)": ops.get("v_div_scale_f64", 0),
        "v_rcp_f64": sum(v for k, v in ops.items() if k.startswith("v_rcp_f64")),
        "s_nop": ops.get("s_nop", 0),
        "ds_read*": sum(v for k, v in ops.items() if k.startswith("ds_read")),
        "ds_write*": sum(v for k, v in ops.items() if k.startswith("ds_write")),
        "scratch_*": sum(v for k, v in ops.items() if k.startswith("scratch_")),
        "global_*": sum(v for k, v in ops.items() if k.startswith("global_")),
        "s_load*": sum(v for k, v in ops.items() if k.startswith("s_load")),
        "v_readlane/writelane": sum(v for k, v in ops.items() if k.startswith(("v_readlane", "v_writelane"))),
        "branches": sum(v for k, v in ops.items() if k.startswith(("s_cbranch", "s_branch"))),
    }
    for k, v in groups.items():
        print(f"{k:32s} {v}")
    print("# top opcodes")
    for k, v in ops.most_common(40):
        print(f"{k:32s} {v}")


if __name__ == "__main__":
    main()
