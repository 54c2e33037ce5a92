#!/usr/bin/env python
"""Assembly + static loop statistics of ONE kernel instantiation of a topology, without building the library:
    python tools/kquad_asm.py <model> ['<explicit instantiation>'] [-- extra hipcc flags]
default instantiation: k_quad<double, Topo, 4>.  Writes /tmp/kq_<model>.s and prints tools/loop_stats.py of it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    name = argv[0]
    inst = argv[1] if len(argv) > 1 else "k_quad<double, Topo, 4>(const BatchArgs<double>)"
    from jiminy_amd import codegen, load_builtin
    try:
        model = load_builtin(name)
    except LookupError:
        from tests import robots
        model = getattr(robots, name)()
    hdr = codegen.write_header(model)
    src = f"/tmp/kq_{name}.cpp"
    with open(src, "w") as f:
        f.write('#include <hip/hip_runtime.h>\n#include JM_TOPO_HEADER\n#include "jm_kernels.h"\n#include "jm_constraint.h"\n'
                '#include "jm_qcon.h"\n#include "jm_adaptive.h"\n#include "jm_qdopri.h"\n'
                f'namespace jm {{ template __global__ void {inst}; }}\n')
    out = f"/tmp/kq_{name}.s"
    cmd = [codegen.HIPCC, f"--offload-arch={codegen.OFFLOAD_ARCH}", "-O3", "-std=c++17", "-x", "hip", f"-I{codegen.CSRC}",
           f"-I{os.path.join(ROOT, 'include')}", f"-DJM_TOPO_HEADER=\"{hdr}\"", "-Wno-unused-value", "-ffp-contract=fast",
           "--cuda-device-only", "-S", src, "-o", out] + extra
    subprocess.check_call(cmd)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "loop_stats.py"), out, inst.split("<")[0]])


if __name__ == "__main__":
    main()
