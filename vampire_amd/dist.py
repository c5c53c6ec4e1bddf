"""Data-parallel plumbing for the hot path: one process per GPU, samples sharded over ranks,
``torch.distributed`` with the "nccl" backend (= RCCL over xGMI on ROCm) on GPUs and "gloo" in
CPU tests.  The lift+render path has no exchange step: the only collective is
the DDP all-reduce of the gradient of the path's one parameter (the density ``beta``)."""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) as set by torch.distributed.run."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    """Join the process group when WORLD_SIZE > 1; returns (rank, world)."""
    rank, _, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_seed(base_seed: int, rank: int) -> int:
    """Per-rank data seed: every rank generates (or would load) a different shard of samples."""
    return base_seed + 7919 * rank


def wrap_ddp(module, device=None):
    """DistributedDataParallel around the step module when running multi-rank, else identity."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return module
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    return DDP(module, device_ids=ids)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(seconds: float, device="cpu") -> float:
    """The timing contract: elapsed time of the slowest rank."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
