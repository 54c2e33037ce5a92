#!/usr/bin/env python
"""Development build of a topology library that KEEPS its object files, so that only the translation units named on the
command line are recompiled:
    python tools/dev_parts.py <model> <tag> [main] [1..9] [-- extra hipcc flags]
        ->  jiminy_amd/csrc/build/libjm_<hash>_<tag>.so   (objects under jiminy_amd/csrc/build/dev_<hash>_<tag>/)
Parts that have no object yet are compiled too.  Select the library at run time with JIMINY_AMD_LIB_TAG=<tag>."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    name, tag, want = argv[0], argv[1], argv[2:]
    os.environ["JIMINY_AMD_LIB_TAG"] = tag
    from jiminy_amd import codegen, load_builtin
    tree_csrc = codegen.CSRC
    if os.environ.get("JIMINY_AMD_CSRC_DEV"):      # sources of an experiment kept outside the tree
        codegen.CSRC = os.environ["JIMINY_AMD_CSRC_DEV"]
    try:
        model = load_builtin(name)
    except LookupError:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import robots
        fn = getattr(robots, name)
        model = fn(False) if name == "tree_arm" else fn()
    hdr = codegen.write_header(model)
    v = codegen.preferred_variant(model)
    lib = codegen.lib_path(model, v)
    objdir = os.path.join(os.path.dirname(lib), f"dev_{model.topology_hash()}_{tag}")
    os.makedirs(objdir, exist_ok=True)
    common = [f"--offload-arch={codegen.OFFLOAD_ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
              f"-DJM_TOPO_HEADER=\"{hdr}\"", "-Wno-unused-value", "-ffp-contract=fast"] + list(codegen.BUILD_VARIANTS[v]) + extra
    parts = ([1, 2, 3, 4, 5, 6] if codegen.quad_structure(model) is not None else [1]) + ([7, 8, 9, 10] if codegen.qcon_split(model) else [])
    pf = codegen.part_flags(model)
    units = {"main": [codegen.HIPCC] + common + ["-DJM_SPLIT_CONSTRAINT", "-c", os.path.join(codegen.CSRC, "jm_lib.cpp")]}
    for p in parts:
        units[str(p)] = [codegen.HIPCC] + common + pf.get(str(p), []) + [f"-DJM_CON_PART={p}", "-c",
                                                                        os.path.join(codegen.CSRC, "jm_lib_constraint.cpp")]
    t = time.time()
    procs = []
    for key, cmd in units.items():
        obj = os.path.join(objdir, f"{key}.o")
        if key in want or not os.path.exists(obj):
            procs.append((key, subprocess.Popen(cmd + ["-o", obj])))
    bad = [key for key, p in procs if p.wait()]
    if bad:
        raise SystemExit(f"failed: {bad}")
    subprocess.check_call([codegen.HIPCC, f"--offload-arch={codegen.OFFLOAD_ARCH}", "-fPIC", "-shared",
                           *[os.path.join(objdir, f"{k}.o") for k in units], "-o", lib])
    codegen.CSRC = tree_csrc     # (the record says "built from the tree's sources": an experiment is loaded, not rebuilt)
    with open(lib + ".src", "w") as f:
        f.write(codegen.source_digest(model, v, None) + "\n")
    print(lib, [k for k, _ in procs], f"{time.time() - t:.0f} s")


if __name__ == "__main__":
    main()
