#!/usr/bin/env python
"""Build an experimental variant of a topology library for A/B timing runs on the GPU box:
    python tools/ab_build.py <model> <tag> [extra hipcc flags...]   ->  jiminy_amd/csrc/build/libjm_<hash>_<tag>.so
Select it at run time with JIMINY_AMD_LIB_TAG=<tag> (codegen.lib_path)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.environ["JIMINY_AMD_LIB_TAG"] = tag
    from jiminy_amd import codegen, load_builtin
    try:
        model = load_builtin(name)
    except LookupError:
        from tests import robots
        model = getattr(robots, name)()
    t = time.time()
    print(codegen.build_library(model, force=True, extra_flags=flags), f"{time.time() - t:.0f} s")


if __name__ == "__main__":
    main()
