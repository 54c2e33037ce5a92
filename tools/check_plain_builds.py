#!/usr/bin/env python
"""Do the topologies pinned in jiminy_amd/csrc/build_variants.json still need their pins?  Runs the engine's own kernel
self-tests (step kernels, constraint kernel, variation kernels, persistent adaptive kernel) on libraries built at plain -O3
without any per-unit flag (`JIMINY_AMD_NO_PART_FLAGS=1 JIMINY_AMD_BUILD_VARIANT=1 JIMINY_AMD_LIB_TAG=plain`: variant 1 = the
compiler's default flags since round 4, variant 0 = the basic SGPR allocator) and prints one JSON line
per (robot, check).  GPU box:
    JIMINY_AMD_NO_PART_FLAGS=1 JIMINY_AMD_BUILD_VARIANT=1 JIMINY_AMD_LIB_TAG=plain python tools/check_plain_builds.py
(the variant checked is JIMINY_AMD_BUILD_VARIANT, default 0; tagged experimental libraries: any JIMINY_AMD_LIB_TAG)"""
import json
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jiminy_amd import engine as E, load_builtin  # noqa: E402
from tests import robots  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    V = int(os.environ.get("JIMINY_AMD_BUILD_VARIANT", "0"))
    cases = [("crane_walker", robots.crane_walker()), ("tree_arm", robots.tree_arm(False)), ("atlas", load_builtin("atlas"))]
    # optional arguments: robot [check index] -- one check per process (a kernel that faults takes its process with it)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    idx = int(sys.argv[2]) if len(sys.argv) > 2 else None
    for name, model in cases:
        if only and name != only:
            continue
        checks = [("step kernels", lambda: E._library_self_test(model, V, torch.float64, dev))]
        if name != "tree_arm":
            checks += [("constraint kernel", lambda: E._constraint_self_test(model, V, dev)),
                       ("variation, spring-damper", lambda: E._variation_self_test(model, V, dev, "spring_damper")),
                       ("variation, constraint", lambda: E._variation_self_test(model, V, dev, "constraint")),
                       ("persistent adaptive kernel", lambda: E._adaptive_self_test(model, V, dev))]
        for i, (what, fn) in enumerate(checks):
            if idx is not None and i != idx:
                continue
            try:
                err = float(fn())
                print(json.dumps({"robot": name, "check": what, "error": err}), flush=True)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"robot": name, "check": what, "exception": repr(e)[:300]}), flush=True)
                traceback.print_exc(file=sys.stderr)


if __name__ == "__main__":
    main()
