"""Single-node data parallelism: one process per GPU, RCCL over xGMI.

New capability (the reference has no multi-GPU code).  Each rank
holds a full replica and a B/N slice of the minibatch; the only exchange per
optimizer step is the all-reduce (sum) of the flat gradient buffer plus one
scalar all-reduce of the target-token count so that every rank back-propagates
sum_local(xent) / sum_global(mask).  L1/L2 terms, per-tensor
clip_by_norm and Adam run after the reduction on identical data on every rank.

The gradient lives in ONE contiguous buffer (variables.py), so the exchange is
a handful of large collectives (``bucket_bytes`` each) instead of one per
tensor; on the fully connected 8-GPU xGMI mesh large messages are what lets
RCCL drive all seven links per GPU.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist

_CURRENT: Optional["DataParallel"] = None


class DataParallel:
    def __init__(self, bucket_bytes: int = 64 << 20) -> None:
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.rank = dist.get_rank()
        self.world_size = dist.get_world_size()
        self.bucket_elems = max(1, bucket_bytes // 4)
        # Host scalars (the global target-token count of a step) travel over a gloo side group:
        # reading an RCCL result back would synchronise the device every step and stop the host
        # from enqueueing ahead of the GPU.
        self._host_group = None
        if self.world_size > 1 and dist.get_backend() == "nccl":
            try:
                self._host_group = dist.new_group(backend="gloo")
            except Exception:           # pylint: disable=broad-except
                self._host_group = None  # fall back to a device all-reduce + readback

    def all_reduce_scalar(self, value: float) -> float:
        """Sum of a host scalar over ranks (global target-token count)."""
        if self.world_size == 1:
            return value
        if dist.get_backend() != "nccl" or self._host_group is not None:
            t = torch.tensor([value], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._host_group)
            return float(t.item())
        t = torch.tensor([value], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def all_reduce_gradients(self, store) -> None:
        """In-place sum of the flat gradient buffer over ranks, in large buckets."""
        if self.world_size == 1:
            return
        grad = store.ensure_grad()
        handles = []
        for start in range(0, grad.numel(), self.bucket_elems):
            chunk = grad[start:start + self.bucket_elems]
            handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
        for hnd in handles:
            hnd.wait()

    def broadcast_parameters(self, store, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s variables."""
        if self.world_size > 1:
            dist.broadcast(store.theta, src=src)

    def shard(self, dataset):
        """This rank's contiguous B/N rows of a batch (SURVEY 8e partitioning)."""
        n = len(dataset)
        per = (n + self.world_size - 1) // self.world_size
        return dataset.subset(self.rank * per, max(0, min(per, n - self.rank * per)))


def init_from_env(backend: Optional[str] = None) -> Optional[DataParallel]:
    """Initialise from torchrun's RANK / WORLD_SIZE / MASTER_* variables."""
    global _CURRENT
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not dist.is_initialized():
        _CURRENT = None
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    _CURRENT = DataParallel()
    return _CURRENT


def current() -> Optional[DataParallel]:
    return _CURRENT


def shutdown() -> None:
    global _CURRENT
    _CURRENT = None
    if dist.is_initialized():
        dist.destroy_process_group()
