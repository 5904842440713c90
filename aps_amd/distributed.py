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


def local_world_size() -> int:
    """ranks on this node (torchrun exports LOCAL_WORLD_SIZE; one node otherwise: the world size)"""
    return int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", 1)))


def rank_cores(local_rank_: int, local_world: int, cores: List[int]) -> List[int]:
    """the host cores of rank `local_rank_` of `local_world` ranks on a node whose usable cores are
    `cores`: a contiguous, equal share (8 ranks on a 256-thread host: 32 each), so that the ranks' Python
    launch threads, their torch CPU pools and the pinned-buffer copies do not migrate across each other's
    caches -- the "host-side feeding problem" SURVEY.md 8(e) names as the limit of 1 -> 8 GPU scaling"""
    cores = sorted(cores)
    if local_world <= 1 or len(cores) < local_world:
        return cores
    per = len(cores) // local_world
    return cores[local_rank_ * per:(local_rank_ + 1) * per]


def bind_rank_to_cores(max_threads: int = 8) -> List[int]:
    """pin this process to its share of the node's cores (`rank_cores`) and cap torch's CPU thread pool;
    returns the cores (empty where the platform has no affinity call).  Called once per rank before any
    work; a no-op for a single rank."""
    if local_world_size() <= 1 or not hasattr(os, "sched_setaffinity"):
        return []
    mine = rank_cores(local_rank(), local_world_size(), list(os.sched_getaffinity(0)))
    if mine:
        os.sched_setaffinity(0, mine)
        th.set_num_threads(max(1, min(max_threads, len(mine))))
    return mine


class PinnedStager(object):
    """Host -> device staging of input batches for one rank: `depth` page-locked host buffers, as many device
    buffers and a copy stream, so that the copy of batch k + 1 over PCIe runs BESIDE the kernels of batch k
    (131 MB of 4-channel waveforms per 128 utterances: 2.1 ms at 63 GB/s if not overlapped).

        slot = stager.stage(first)                  # host memcpy into pinned memory + async H2D on the copy stream
        for nxt in batches:
            ahead = stager.stage(nxt)               # queued BEFORE this step's kernels: overlaps them
            x = stager.use(slot)                    # the compute stream waits for THIS slot's copy only
            step(x)
            stager.release(slot)                    # an event behind the step: the slot's buffers may be refilled
            slot = ahead

    What orders what: a slot's copy waits for the event `release` recorded behind the LAST consumer of its device
    buffer -- not for the head of the compute stream, which would put it behind the kernels it is meant to run
    beside -- and `use` makes the current stream wait for that slot's copy event only.  (Rounds 3-4's `put` waited
    for the compute stream's head before the copy and for the copy right after it: fully serialised.)  `put(batch)`
    = release + stage + use in one call: correct, no overlap -- the form for callers that do not pipeline.  Without a GPU (the CPU tests) the buffers are ordinary memory."""

    def __init__(self, shape, dtype=th.float32, device=None, depth: int = 2) -> None:
        self.device = th.device("cpu") if device is None else th.device(device)
        cuda = self.device.type == "cuda"
        self.host = [th.empty(shape, dtype=dtype, pin_memory=cuda) for _ in range(depth)]
        self.dev = [th.empty(shape, dtype=dtype, device=self.device) for _ in range(depth)] if cuda else None
        self.stream = th.cuda.Stream(device=self.device) if cuda else None
        self.copied = [None] * depth   # the slot's H2D copy has finished (recorded on the copy stream)
        self.freed = [None] * depth    # the slot's last consumer has been enqueued (recorded on its stream)
        self.k = 0

    def stage(self, batch: th.Tensor) -> int:
        """copy `batch` (CPU) into the next slot and start its transfer; returns the slot"""
        i = self.k % len(self.host)
        self.k += 1
        if self.dev is None:
            self.host[i].copy_(batch)
            return i
        if self.copied[i] is not None:
            self.copied[i].synchronize()  # the slot's previous transfer has left the host memory
        self.host[i].copy_(batch)
        if self.freed[i] is not None:
            self.stream.wait_event(self.freed[i])  # the device buffer's last reader is behind us
        with th.cuda.stream(self.stream):
            self.dev[i].copy_(self.host[i], non_blocking=True)
            ev = th.cuda.Event()
            ev.record(self.stream)
        self.copied[i] = ev
        self.freed[i] = None
        return i

    def use(self, slot: int) -> th.Tensor:
        """the slot's device tensor, safe to read on the CURRENT stream from here on"""
        if self.dev is None:
            return self.host[slot]
        th.cuda.current_stream(self.device).wait_event(self.copied[slot])
        return self.dev[slot]

    def release(self, slot: int) -> None:
        """everything enqueued so far on the current stream was the slot's last reader"""
        if self.dev is None:
            return
        ev = th.cuda.Event()
        ev.record(th.cuda.current_stream(self.device))
        self.freed[slot] = ev

    def put(self, batch: th.Tensor) -> th.Tensor:
        """stage + use in one call: the slot about to be refilled is released against everything enqueued so far
        (whoever still reads its device buffer is in front of the copy), so the call is safe without `release`
        calls and overlaps nothing -- see the class for the pipelined form"""
        self.release(self.k % len(self.host))
        return self.use(self.stage(batch))


def ddp_kwargs(bucket_cap_mb: int = 32) -> dict:
    """DistributedDataParallel settings of the training path (aps/trainer/ddp.py:97-110 wraps the task
    with the defaults): gradients are views of the reducer's buckets (no copy into them, none back),
    the graph is static (the joint model runs the same autograd graph every step, so the reducer may
    pre-order its buckets by the first step's arrival order), and the buckets are `bucket_cap_mb` MB:
    a ring all-reduce over the node's point-to-point xGMI links moves 2 (N - 1) / N x the bucket per rank
    over ONE link per hop (~ 153 GB/s peak, ~ 75 % achievable): ~ 0.5 ms per 32 MB bucket at N = 8, ~ 3.9 ms
    for the 231 MB of fp32 gradients of the joint model (8 buckets) against ~ 17 ms of backward, so only
    the last bucket (the front end's and mask estimator's 10 MB) is exposed.  DESIGN.md section 6 has
    the budget."""
    return dict(bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, static_graph=True)


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
