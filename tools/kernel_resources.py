#!/usr/bin/env python
"""Print the register / scratch / LDS footprint of every kernel in a built topology library
(llvm-objdump --offloading + llvm-readelf --notes on the gfx950 code objects)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def resources(lib: str):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        link = os.path.join(tmp, os.path.basename(lib))
        os.symlink(os.path.abspath(lib), link)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], cwd=tmp, capture_output=True)
        for co in sorted(glob.glob(os.path.join(tmp, "*gfx950*"))):
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
                name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
                out.append({"kernel": re.sub(r"\(.*", "", name), "vgpr": get("vgpr_count"), "agpr": blk.split()[0],
                            "vgpr_spill": get("vgpr_spill_count"), "sgpr_spill": get("sgpr_spill_count"),
                            "scratch_B": get("private_segment_fixed_size"), "lds_B": get("group_segment_fixed_size")})
    return out


if __name__ == "__main__":
    for lib in sys.argv[1:]:
        print(lib)
        for r in resources(lib):
            if any(k in r["kernel"] for k in ("k_quad", "k_qcon", "k_constrained", "k_batch")):
                print("  ", r)
