#!/usr/bin/env python
"""Reproducer of the toolchain hazard of DESIGN.md section 4.7 (hipcc 7.2 / AMD clang 22, gfx950).

The in-loop copy of the branch-parallel evaluation of the `crane_walker` test robot (k_quad: 512 VGPRs, SGPR ->
VGPR spills + a small scratch frame) has been seen to return non-deterministic garbage with the default flags,
while `-mllvm -disable-machine-licm` (build variant 1) of the SAME sources is right.  This script, run on an
MI355X box, builds both variants, runs the kernel self-test of the engine (two explicit-Euler steps with and
without the a(t+) refresh: the in-loop copy against the peeled copy of the evaluation) several times on each,
and prints the disagreement per run and per variant -- a healthy build stays below 1e-9 and is bitwise
repeatable.

    python tools/repro/repro_crane_walker.py            (on the GPU box, through gpurun)
    python tools/repro/repro_crane_walker.py --isa      (anywhere: writes the instruction histograms of both
                                                         variants' k_quad next to this script)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


EXPERIMENTS = (("-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"), ("-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"),
               ("-mllvm", "-disable-postra-machine-licm"), ("-mllvm", "-hoist-const-loads=0"),
               ("-mllvm", "-avoid-speculation=1"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true", help="build every variant (no GPU needed), then exit")
    ap.add_argument("--isa", action="store_true")
    ap.add_argument("--runs", type=int, default=5)
    args = ap.parse_args()
    from jiminy_amd import codegen
    from tests import robots
    model = robots.crane_walker()
    # experiments that localise the fault: spill SGPRs / VGPRs to scratch memory instead of VGPR lanes / AGPRs
    codegen.BUILD_VARIANTS = tuple(codegen.BUILD_VARIANTS) + EXPERIMENTS
    if args.build:
        for v in range(len(codegen.BUILD_VARIANTS)):
            print(v, codegen.BUILD_VARIANTS[v], codegen.build_library(model, variant=v))
        return
    if args.isa:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_histogram as ih
        for v in (0, 1):
            name, asm = ih.kernel_asm("crane_walker", "k_quad<double", list(codegen.BUILD_VARIANTS[v]))
            ops = ih.histogram(asm)
            out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"crane_walker_k_quad_variant{v}.txt")
            with open(out, "w") as f:
                f.write(f"# {name}\n# flags: {' '.join(codegen.BUILD_VARIANTS[v]) or '(default)'}\n# instructions: {sum(ops.values())}\n")
                for k, n in ops.most_common():
                    f.write(f"{k:32s} {n}\n")
            print("wrote", out)
        return
    import torch
    from jiminy_amd import engine as E
    dev = torch.device("cuda", 0)
    for v in range(len(codegen.BUILD_VARIANTS)):
        codegen.build_library(model, variant=v)
        errs = []
        for _ in range(args.runs):
            e, legs = E._library_self_test_detail(model, v, torch.float64, dev)
            errs.append(e)
        print("   last run, per solver / batch size:", {k: "%.2e" % x for k, x in legs.items()})
        print(f"variant {v} ({' '.join(codegen.BUILD_VARIANTS[v]) or 'default flags'}): in-loop vs peeled evaluation, "
              f"relative disagreement per run: {['%.2e' % e for e in errs]}")


if __name__ == "__main__":
    main()
