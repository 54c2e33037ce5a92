"""Multi-GPU sharding of the batch (one process per GPU, `torch.distributed`, backend "nccl" = RCCL).

The physics has no cross-lane dependency: rank r of W owns lanes [r*B/W, (r+1)*B/W) and steps
them with its own engine; there is NO collective on the data path.  The only optional exchange
is an all-gather of the per-rank observation block for a single-learner topology
(BASELINE.json config 4), issued on the current stream.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous lane range of `rank`; the first `global_batch % world_size` ranks get one more."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("invalid rank / world_size")
    base, extra = divmod(int(global_batch), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_observations(fields: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None
                      ) -> torch.Tensor:
    """Concatenate SoA observation blocks `[rows_i][B]` into one `[sum rows][B]` tensor."""
    rows = sum(int(f.shape[0]) for f in fields)
    B = int(fields[0].shape[1])
    if out is None:
        out = torch.empty((rows, B), dtype=fields[0].dtype, device=fields[0].device)
    r = 0
    for f in fields:
        out[r:r + f.shape[0]].copy_(f)
        r += f.shape[0]
    return out


def all_gather_observations(fields: Sequence[torch.Tensor],
                            out: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
    """All-gather of the local observation block over the process group.

    Returns `[packed_local, gathered]` where `gathered` has shape `[world][rows][B_local]`
    (rank-major = lane-major because shards are contiguous lane ranges).  Buffers are reused
    between calls when `out` is passed back.
    """
    world = dist.get_world_size()
    packed = pack_observations(fields, out[0] if out else None)
    rows, B = int(packed.shape[0]), int(packed.shape[1])
    gathered = out[1] if out else torch.empty((world, rows, B), dtype=packed.dtype,
                                               device=packed.device)
    # concatenation along dim 0 of the flat view (layout accepted by both RCCL and gloo)
    dist.all_gather_into_tensor(gathered.view(world * rows, B), packed)
    return [packed, gathered]
