"""Build the topology libraries of every authored test robot that runs the one-robot-per-lane kernels (a quick partial
rebuild while iterating on jm_kernels.h / jm_constraint.h: ~2 min instead of the full build)."""
import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import robots
from jiminy_amd import codegen
from concurrent.futures import ThreadPoolExecutor
ms = [m for m in robots.all_test_models() + robots.frame_constraint_models() if codegen.quad_structure(m) is None]
seen = {}
for m in ms: seen.setdefault(m.topology_hash(), m)
t = time.time()
with ThreadPoolExecutor(6) as ex:
    for lib in ex.map(codegen.build_library, seen.values()): print(lib)
print(f"{len(seen)} libraries, {time.time()-t:.0f} s")
