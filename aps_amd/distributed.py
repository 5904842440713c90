"""
Process-group plumbing for the data-parallel path (mirror of aps/distributed/backend.py:33-129,
torch.distributed branch only; Horovod is out of scope).  One process per GPU; on ROCm the "nccl"
backend IS RCCL, which runs over xGMI inside a node.

The forward hot path has NO collective: utterances are independent, so ranks only agree on which
utterances they own (`shard_indices`, the DistributedSampler rule of aps/loader/se/chunk.py:275-280
and aps/loader/am/utils.py:17-37) and reduce timings / counts at the end (`reduce_max`,
`all_reduce`).
"""
import os
from typing import List

import torch as th
import torch.distributed as dist

BACKENDS = ["torch", "none"]


def init(backend: str = "torch", device_backend: str = "nccl") -> None:
    """init_process_group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)"""
    if backend not in BACKENDS:
        raise ValueError(f"Unsupported distributed backend: {backend}")
    if backend == "none" or dist.is_initialized():
        return
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group(backend=device_backend, init_method="env://")


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def rank() -> int:
    return dist.get_rank() if is_initialized() else 0


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", 0))


def world_size() -> int:
    return dist.get_world_size() if is_initialized() else 1


def shard_indices(num_items: int, rank_: int, world: int) -> List[int]:
    """rank-strided ownership of utterances / batches: indices[rank::world]"""
    return list(range(rank_, num_items, world))


def all_reduce(tensor: th.Tensor) -> th.Tensor:
    """sum over ranks, divided by world size (aps/distributed/backend.py:118-129)"""
    if not is_initialized():
        return tensor
    out = tensor.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out / world_size()


def reduce_max(value: float, device) -> float:
    if not is_initialized():
        return value
    t = th.tensor([value], dtype=th.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value: float, device) -> float:
    if not is_initialized():
        return value
    t = th.tensor([value], dtype=th.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier() -> None:
    if is_initialized():
        dist.barrier()
