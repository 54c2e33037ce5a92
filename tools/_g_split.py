import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from jiminy_amd import load_builtin
from jiminy_amd.engine import BatchedEngine
from jiminy_amd.synthetic import sample_standing_states
model = load_builtin("anymal"); B = 65536; dev = torch.device("cuda", 0)
def run(lift, model_name, n_sub=5, tag="", tol=1e-5):
    st = sample_standing_states(model, B, seed=0, joint_noise=0.0, base_angle_max=0.0, twist_std=0.0, joint_vel_std=0.0,
                                command_fraction=0.0, out_of_bounds_fraction=0.0, depth_range=(-1e-3, -1e-3))
    st["q"][2] += lift
    eng = BatchedEngine(model, B, dtype=torch.float64, device=dev)
    dt = 1e-3
    eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": n_sub * dt,
                                 "sensorsUpdatePeriod": n_sub * dt, "tolAbs": tol, "tolRel": tol * 10}, "contacts": {"model": model_name}})
    eng.set_command(torch.zeros((12, B), dtype=torch.float64))
    eng.start(torch.from_numpy(st["q"]), torch.from_numpy(st["v"]))
    for _ in range(2): eng.step(n_sub * dt)
    torch.cuda.synchronize(); eng.enable_timing(True)
    for _ in range(4): eng.step(n_sub * dt)
    n, ms = eng.timing_summary()
    act = (eng.field("con_flags") & 1).sum(0).double().mean().item() if model_name == "constraint" else -1
    print(f"{tag}: {ms / n:.3f} ms per launch of {n_sub} Euler steps (+1 refresh) -> {ms / n / (n_sub + 1):.3f} ms per evaluation; active constraints {act:.2f}")
run(0.0, "constraint", tag="standing 4 feet (16 rows)")
run(0.0, "constraint", tag="standing, PGS stops after 1 sweep", tol=1e6)
run(0.0, "constraint", tag="standing, PGS tol 1e-9          ", tol=1e-9)
run(0.3, "constraint", tag="in the air (0 rows)     ")
import os
os.environ["JM_KERNEL_VARIANT"] = "lane"
run(0.3, "spring_damper", tag="lane kernel, spring      ")
