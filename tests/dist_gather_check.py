"""Worker of tests/test_multi_gpu.py (one process per GPU under `torch.distributed.run`, backend nccl = RCCL): every
rank steps its contiguous shard of one seeded ANYmal batch and the asynchronous observation all-gather
(`jiminy_amd.distributed.ObservationGather`) must hand every rank the blocks of ALL ranks.  The check needs no second
collective: the lanes are seeded by their GLOBAL index, so each rank also steps the whole batch on its own GPU and the
gathered tensor must equal that single-GPU result bit for bit (shard / replica invariance of the kernels)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    from jiminy_amd import load_builtin
    from jiminy_amd.distributed import ObservationGather, shard_range
    from jiminy_amd.engine import BatchedEngine
    from jiminy_amd.synthetic import sample_states
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model = load_builtin("anymal")
    B, dt, steps = 512 * world, 1e-3, 5
    st = sample_states(model, B, seed=77)
    lo, hi = shard_range(B, rank, world)

    def run(sl):
        eng = BatchedEngine(model, sl.stop - sl.start, dtype=torch.float64, device=dev)
        eng.set_options({"stepper": {"odeSolver": "runge_kutta_4", "dtMax": dt, "controllerUpdatePeriod": dt,
                                     "sensorsUpdatePeriod": dt}, "contacts": {"model": "spring_damper"}})
        eng.set_command(torch.from_numpy(np.ascontiguousarray(st["command"][:, sl])))
        eng.start(torch.from_numpy(np.ascontiguousarray(st["q"][:, sl])), torch.from_numpy(np.ascontiguousarray(st["v"][:, sl])))
        return eng
    shard, whole = run(slice(lo, hi)), run(slice(0, B))
    fields = ("q", "v", "imu", "encoder")
    gather = ObservationGather()
    ok = True
    for _ in range(steps):
        shard.step(dt)
        whole.step(dt)
        gather.launch([shard.field(k) for k in fields])        # in flight while the next step integrates
        got = gather.result()                                   # [world][rows][B_local]
        want = torch.cat([whole.field(k) for k in fields], dim=0)       # [rows][B]
        rows = want.shape[0]
        got_lane_major = got.permute(1, 0, 2).reshape(rows, B)
        same = (got_lane_major == want) | (torch.isnan(got_lane_major) & torch.isnan(want))
        ok = ok and bool(same.all())
    gather.drain()
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"world": world, "backend": dist.get_backend(), "gathered_equals_single_gpu": bool(flag.item())}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
