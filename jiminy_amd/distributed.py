"""Multi-GPU sharding of the batch (one process per GPU, `torch.distributed`, backend "nccl" = RCCL).

The physics has no cross-lane dependency: rank r of W owns lanes [r*B/W, (r+1)*B/W) and steps
them with its own engine; there is NO collective on the data path.  The only optional exchange
is an all-gather of the per-rank observation block for a single-learner topology
(BASELINE.json config 4).  It is issued asynchronously: RCCL runs it on the process group's own
stream behind an event recorded after the pack, so it overlaps with the next physics launches of the
compute stream; the result is awaited only where it is consumed (`ObservationGather`).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous lane range of `rank`; the first `global_batch % world_size` ranks get one more.
    (The observation all-gather needs equal shards: `global_batch % world_size == 0`.)"""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("invalid rank / world_size")
    base, extra = divmod(int(global_batch), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_observations(fields: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None,
                      dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Concatenate SoA observation blocks `[rows_i][B]` into one `[sum rows][B]` tensor.  `dtype`: the type of the
    packed block (default: the fields' own) -- float32 halves what a whole-batch learner's gather moves over xGMI;
    the conversion is fused into the pack copy."""
    rows = sum(int(f.shape[0]) for f in fields)
    B = int(fields[0].shape[1])
    if out is None:
        out = torch.empty((rows, B), dtype=dtype or fields[0].dtype, device=fields[0].device)
    r = 0
    for f in fields:
        out[r:r + f.shape[0]].copy_(f)      # (copy_ converts when the packed block is narrower than the fields)
        r += f.shape[0]
    return out


def _check_equal_shards(B_local: int) -> None:
    """`all_gather_into_tensor` needs the same shard size on every rank; a ragged split would hang or
    raise inside the collective, so it is refused up front (one tiny all-reduce, first call only)."""
    t = torch.tensor([B_local, -B_local], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise ValueError("all_gather_observations needs equal shards on every rank: make the global batch "
                         "a multiple of the world size (shard_range gives the first ranks one lane more otherwise)")


class ObservationGather:
    """Double-buffered asynchronous all-gather of the packed observation block.

    `launch(fields)` packs the local block on the current (compute) stream and starts the collective
    without waiting for it; `result()` makes the current stream wait for the most recent collective and
    returns `[world][rows][B_local]` (rank-major = lane-major because shards are contiguous lane
    ranges).  Two buffer pairs alternate, so the learner may still be reading gather k while gather
    k+1 is in flight and the physics of step k+2 runs.

    What a ring all-gather costs on xGMI is bytes per link (7/8 of world x block through ~153 GB/s links), so the two
    knobs are the bytes: `dtype=torch.float32` packs the block in single precision (a policy network consumes float32
    anyway: 34.6 -> 17.3 MB per rank and step for ANYmal's 66 float64 rows at B = 65 536), and `every=k` gathers only at
    every k-th `launch` call (a learner that acts every k physics steps: the gym step of the reference's ANYmal
    environment is 8 x 5 engine steps); the calls in between return at once and `result()` keeps handing out the last
    gathered block."""

    def __init__(self, dtype: Optional[torch.dtype] = None, every: int = 1) -> None:
        if every < 1:
            raise ValueError("every must be >= 1")
        self._bufs: List[Optional[List[torch.Tensor]]] = [None, None]
        self._work: List[Optional[object]] = [None, None]
        self._turn = 0
        self._checked = False
        self.dtype = dtype
        self.every = int(every)
        self._calls = 0
        self.launched = 0      # collectives actually started

    def launch(self, fields: Sequence[torch.Tensor]) -> bool:
        """Returns True when this call started a collective (every `every`-th call, beginning with the first)."""
        self._calls += 1
        if (self._calls - 1) % self.every != 0:
            return False
        world = dist.get_world_size()
        i = self._turn
        if self._work[i] is not None:    # the buffer pair is about to be overwritten
            self._work[i].wait()
            self._work[i] = None
        if not self._checked:
            _check_equal_shards(int(fields[0].shape[1]))
            self._checked = True
        buf = self._bufs[i]
        packed = pack_observations(fields, buf[0] if buf else None, dtype=self.dtype)
        rows, B = int(packed.shape[0]), int(packed.shape[1])
        gathered = buf[1] if buf else torch.empty((world, rows, B), dtype=packed.dtype, device=packed.device)
        self._bufs[i] = [packed, gathered]
        # concatenation along dim 0 of the flat view (layout accepted by both RCCL and gloo)
        self._work[i] = dist.all_gather_into_tensor(gathered.view(world * rows, B), packed, async_op=True)
        self._turn = 1 - i
        self.launched += 1
        return True

    @property
    def bytes_per_rank(self) -> int:
        """Size of the packed local block of the last launch (what every rank contributes to one collective)."""
        b = self._bufs[1 - self._turn]
        return 0 if b is None else int(b[0].numel() * b[0].element_size())

    def result(self) -> torch.Tensor:
        i = 1 - self._turn
        if self._bufs[i] is None:
            raise RuntimeError("no gather was launched")
        if self._work[i] is not None:
            self._work[i].wait()
            self._work[i] = None
        return self._bufs[i][1]

    def drain(self) -> None:
        for i in (0, 1):
            if self._work[i] is not None:
                self._work[i].wait()
                self._work[i] = None


def all_gather_observations(fields: Sequence[torch.Tensor],
                            out: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
    """Blocking form: all-gather of the local observation block over the process group.

    Returns `[packed_local, gathered]` where `gathered` has shape `[world][rows][B_local]`.  Buffers
    are reused between calls when `out` is passed back.  Every rank must own the same number of lanes."""
    world = dist.get_world_size()
    if out is None:
        _check_equal_shards(int(fields[0].shape[1]))
    packed = pack_observations(fields, out[0] if out else None)
    rows, B = int(packed.shape[0]), int(packed.shape[1])
    gathered = out[1] if out else torch.empty((world, rows, B), dtype=packed.dtype,
                                               device=packed.device)
    dist.all_gather_into_tensor(gathered.view(world * rows, B), packed)
    return [packed, gathered]
