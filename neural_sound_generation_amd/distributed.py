"""Single-node data parallelism: one process per GPU, model replicated, minibatch sharded by clips,
ONE all-reduce (sum) of the flat gradient bucket per step over RCCL/xGMI (torch.distributed backend
"nccl" is RCCL on ROCm), 1/world_size folded into the optimiser kernel.

The reference has no DDP wiring at all; the semantics implemented are plain
DDP's: per-rank BatchNorm statistics, gradients averaged over ranks, identical update everywhere.
Messages are small (5 MB at D=128) and latency-bound, hence one bucket, one collective.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world_size, local_rank); a no-op single-process setup when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # NSG_DIST_BACKEND: rehearsal knob (e.g. gloo with every rank on one GPU); production is RCCL
            backend = os.environ.get("NSG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def broadcast_flat(flat: torch.Tensor, src: int = 0, group=None):
    """Make every replica start from rank `src`'s parameters."""
    if world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def state_outside(model, opt) -> list:
    """Tensors of `model`'s state that do NOT live in `opt`'s flat parameter bucket: buffers (BatchNorm running statistics and
    counters, the EMA codebook's ema_count / ema_sum) and parameters the optimiser skips (requires_grad=False: an EMA-trained
    codebook).  state_dict order, so every rank lists the same tensors."""
    lo = opt.flat_param.data_ptr()
    hi = lo + opt.flat_param.numel() * opt.flat_param.element_size()
    return [t for t in model.state_dict(keep_vars=True).values() if not (lo <= t.data_ptr() < hi)]


def broadcast_tensors_packed(tensors, src: int = 0, group=None):
    """Broadcast rank `src`'s values of `tensors` (any dtypes) as ONE flat message: each is widened to float64 (exact for fp32
    and for the int64 step counters below 2^53), sent together, and copied back in place."""
    if world_size(group) == 1 or not tensors:
        return
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).to(torch.float64) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.detach().copy_(flat[off:off + n].view(t.shape).to(t.dtype))
            off += n


def allreduce_sum_(flat: torch.Tensor, group=None, async_op: bool = False):
    """In-place sum over ranks of one flat bucket; returns the work handle when async."""
    if world_size(group) == 1:
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class TwoPartAllReduce:
    """Sum-all-reduce of one flat buffer in two collectives: start_back() sends [split:] as soon as that part is final (it runs
    on the collective's own stream / thread beside whatever still fills [:split]); finish() sends [:split] and joins both.
    Without a start_back() since the last finish(), finish() is one collective over the whole buffer.  Same sums either way."""

    def __init__(self, flat: torch.Tensor, split: int, group=None):
        self.flat, self.split, self.group = flat, int(split), group
        self.work = None

    def start_back(self):
        if world_size(self.group) > 1 and 0 < self.split < self.flat.numel() and self.work is None:
            self.work = allreduce_sum_(self.flat[self.split:], self.group, async_op=True)

    def finish(self):
        if world_size(self.group) == 1:
            return
        work, self.work = self.work, None
        if work is None:
            allreduce_sum_(self.flat, self.group)
        else:
            allreduce_sum_(self.flat[:self.split], self.group)
            work.wait()


def shard_batch(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank r takes clips [r*B/world, (r+1)*B/world) of the global batch."""
    if batch.size(0) % world != 0:
        raise ValueError(f"global batch {batch.size(0)} is not divisible by world size {world}")
    per = batch.size(0) // world
    return batch[rank * per:(rank + 1) * per]
