#!/usr/bin/env python
"""fp64 instruction mix of the evaluation loop of the dominant kernel, from the compiler's assembly (tools/kquad_asm.py ->
tools/loop_stats.py), written where bench.py finds it (`roofline.flops`):
    python tools/isa_mix.py [model] [tag]     ->  profiles/<tag>_<model>_isa_mix.json  (+ profiles/isa_mix_<model>_latest.json)
flops per VALU instruction of the loop = (2 fma + mul + add) / VALU; bench.py multiplies it with the MEASURED VALU
instructions per wave-launch (SQ_INSTS_VALU of the committed PMC profile) to get the flops a launch executes."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "anymal"
    tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kquad_asm.py"), model, "--", "-mllvm", "-sgpr-regalloc=basic"],
                         capture_output=True, text=True, check=True).stdout
    loops = [(int(m.group(1)), int(m.group(2)), eval(m.group(3))) for m in re.finditer(r"loop \S+: (\d+) instr, VALU (\d+): (\{.*\})", out)]
    n, valu, mix = max(loops)            # the evaluation loop is the largest one
    whole = eval(re.search(r"whole kernel: \d+ (\{.*\})", out).group(1))
    fma, mul, add = mix.get("fma64", 0), mix.get("mul64", 0), mix.get("add64", 0)
    rec = {"model": model, "kernel": "jm::k_quad<double, Topo, 4>", "source": "hipcc -S, static count of the evaluation loop (tools/loop_stats.py)",
           "loop_instructions": n, "loop_valu": valu, "loop_fma64": fma, "loop_mul64": mul, "loop_add64": add,
           "loop_dpp": mix.get("dpp", 0), "loop_valu_other": mix.get("valu_other", 0), "loop_lds": mix.get("lds", 0),
           "flops_per_valu_instruction_per_lane": (2 * fma + mul + add) / valu,
           "fp64_arith_share_of_valu": (fma + mul + add) / valu, "fused_share_of_fp64_arith": fma / (fma + mul + add),
           "whole_kernel": whole}
    for path in (os.path.join(ROOT, "profiles", f"{tag}_{model}_isa_mix.json"), os.path.join(ROOT, "profiles", f"isa_mix_{model}_latest.json")):
        with open(path, "w") as f:
            json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
