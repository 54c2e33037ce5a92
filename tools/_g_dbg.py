import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_constraint_model import _models, _pair, TIGHT
from tests.helpers import oracle_batch
from jiminy_amd.engine import BatchedEngine
name = sys.argv[1]
model = _models()[name]()
B = 96
ref, _ = _pair(model, B, seed=11)
eng = BatchedEngine(model, B, dtype=torch.float64, device=torch.device("cuda", 0),
                    extra_outputs=("contact_forces", "f_external"))
dt = 5e-4
eng.set_options({"stepper": {"odeSolver": "euler_explicit", "dtMax": dt, "controllerUpdatePeriod": dt,
                             "sensorsUpdatePeriod": dt, "tolAbs": TIGHT["tol_abs"], "tolRel": TIGHT["tol_rel"]},
                 "contacts": {"model": "constraint"}})
if model.nmotors:
    eng.set_command(torch.from_numpy(ref["command"]))
eng.start(torch.from_numpy(ref["q"]), torch.from_numpy(ref["v"]))
oracle_batch(model, ref, "start", constraint_options=TIGHT)
torch.cuda.synchronize()
a = eng.field("a").cpu().numpy()
err = np.abs(a - ref["a"]).max(0) / np.maximum(np.abs(ref["a"]).max(0), 1)
fl = eng.field("con_flags").cpu().numpy()
print("flags equal", np.array_equal(fl, ref["con_flags"]))
nact = (ref["con_flags"] & 1).sum(0)
for l in np.argsort(-err)[:10]:
    print("lane", l, "err %.2e" % err[l], "active", nact[l], "flags", ref["con_flags"][:, l], "status", int(eng.status.cpu().numpy().reshape(-1)[l]), ref["status"].reshape(-1)[l])
    print("   lam dev", eng.field("con_data").cpu().numpy()[:, l])
    print("   lam ref", ref["con_data"][:, l])
print("lanes with err>1e-6:", int((err > 1e-6).sum()), "of", B, " by nact:", {int(k): int(((err > 1e-6) & (nact == k)).sum()) for k in np.unique(nact)})
