import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import robots
from jiminy_amd import codegen
sys.path.insert(0,'tools')
import kernel_resources
name = sys.argv[1] if len(sys.argv) > 1 else "arm7"
fn = getattr(robots, name)
m = fn(False) if name == "tree_arm" else fn()
t=time.time(); lib = codegen.build_library(m); print(lib, f"{time.time()-t:.0f} s")
for r in kernel_resources.resources(lib):
    if any(k in r["kernel"] for k in ("k_constrained", "k_batch")): print("  ", r)
