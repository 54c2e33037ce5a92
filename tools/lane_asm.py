#!/usr/bin/env python
"""Assembly + static loop statistics of the one-robot-per-lane step kernel of a test robot, compiled on its own
(seconds instead of the whole library):  python tools/lane_asm.py <robot> [con] [-- extra hipcc flags]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import robots  # noqa: E402
from jiminy_amd import codegen  # noqa: E402

argv = sys.argv[1:]
extra = []
if "--" in argv:
    i = argv.index("--")
    argv, extra = argv[:i], argv[i + 1:]
name = argv[0]
con = "con" in argv[1:]
fn = getattr(robots, name)
model = fn(False) if name == "tree_arm" else fn()
hdr = codegen.write_header(model)
src = f"/tmp/lane/{name}.hip"
with open(src, "w") as f:
    f.write('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n'
            f'#include "{hdr}"\n#include "jm_kernels.h"\n#include "jm_constraint.h"\nnamespace jm {{\n'
            + ("template __global__ void k_constrained<double, Topo>(const BatchArgs<double>, const ConArgs<double>);\n" if con else
               "template __global__ void k_batch<double, Topo>(const BatchArgs<double>);\n") + "}\n")
out = f"/tmp/lane/{name}{'_con' if con else ''}.s"
v = codegen.preferred_variant(model)
cmd = [codegen.HIPCC, f"--offload-arch={codegen.OFFLOAD_ARCH}", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
       "-I", os.environ.get("LANE_CSRC", codegen.CSRC), "-I", os.path.join(ROOT, "include"), "-Wno-unused-value", "-ffp-contract=fast"] + list(codegen.BUILD_VARIANTS[v]) + extra + [src, "-o", out]
subprocess.check_call(cmd)
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "loop_stats.py"), out, "k_"])
