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


def init(backend: str, device=None, force: bool = False):
    """Join the process group when WORLD_SIZE > 1 (or when `force`d: a world-size-1 group, which
    the single-GPU RCCL smoke test uses to exercise the real backend); returns (rank, world)."""
    rank, _, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_seed(base_seed: int, rank: int) -> int:
    """Per-rank data seed: every rank generates (or would load) a different shard of samples."""
    return base_seed + 7919 * rank


class GradSync(torch.nn.Module):
    """Minimal overlapped data-parallel gradient averaging: the parameters are grouped, in reverse
    registration order (about the order in which backward completes them), into buckets of at most
    `bucket_bytes`; a post-accumulate hook counts the gradients of a bucket and, when the last one is
    there, flattens the bucket and launches ONE asynchronous all-reduce for it -- for the hot path's
    own parameter (`beta`, a bucket of one) that is the end of the render backward, so the collective
    runs under the lift backward; for the layered step (step.LayeredStep, 777 111 parameters) three
    1 MiB buckets go out while the backward is still running.  `finish()` (called by `train_step`
    after backward) makes the compute stream wait for the collectives and scatters the averaged
    buckets back.  Same result as DistributedDataParallel (mean over ranks) without its per-step
    bookkeeping, which costs 45 us of a 1.07 ms step on one MI355X (tools/ddp_overhead.py).
    `VAMP_GRAD_SYNC=ddp` selects DDP.

    Like DDP it broadcasts parameters and buffers from rank 0 at construction, and it joins its
    collectives by itself at the end of every backward pass (an autograd-engine callback queued by
    the first hook of the pass), so a plain `loss.backward(); optimizer.step()` is correct too;
    `finish()` stays as an explicit, idempotent join.  Every rank must produce a gradient for every
    parameter in every step (as with DDP's find_unused_parameters=False, base_cli.py:72)."""

    def __init__(self, module, group=None, bucket_bytes: int = 1 << 20):
        super().__init__()
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group)
        self._pending = []
        self._joining = False
        # False: the hooks do nothing (a caller that replays the step from a HIP graph captures it
        # without the collective and averages the gradients itself after each replay)
        self.enabled = True
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        # NCCL / RCCL has a native average; gloo (CPU tests) sums and the division follows
        self._avg = dist.get_backend(group) == "nccl"
        params = [p for p in module.parameters() if p.requires_grad][::-1]
        self._buckets, cur, size = [], [], 0
        for p in params:
            nb = p.numel() * p.element_size()
            if cur and (size + nb > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nb
        if cur:
            self._buckets.append(cur)
        self._bucket_of = {p: i for i, b in enumerate(self._buckets) for p in b}
        self._ready = [0] * len(self._buckets)
        for p in params:
            p.register_post_accumulate_grad_hook(self._launch)

    def _launch(self, p):
        if not self.enabled:
            return
        if not self._joining:
            # join at the end of this backward pass, whoever called it
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
            self._joining = True
        i = self._bucket_of[p]
        self._ready[i] += 1
        if self._ready[i] == len(self._buckets[i]):
            self._reduce(i)

    def _reduce(self, i):
        bucket = self._buckets[i]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        flat = bucket[0].grad if len(bucket) == 1 else torch.cat([q.grad.reshape(-1) for q in bucket])
        self._pending.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), flat, bucket))
        self._ready[i] = 0

    def forward(self, *a, **kw):
        # a backward pass that raised after its first hook leaves stale handles and the flag behind (the
        # engine drops its callbacks): start every step clean
        if self._pending or self._joining or any(self._ready):
            for work, _, _ in self._pending:
                work.wait()
            self._pending.clear()
            self._joining = False
            self._ready = [0] * len(self._buckets)
        return self.module(*a, **kw)

    def finish(self):
        """Order the caller's stream after the collectives launched during backward and scatter the
        averaged buckets back into the gradients."""
        for i, n in enumerate(self._ready):
            if not (n and self.enabled):
                continue
            # a bucket some of whose parameters got a gradient in this pass and some did not: reducing
            # nothing would leave the ones that did rank-local and the ranks would drift apart silently
            # (DistributedDataParallel raises here too, find_unused_parameters=False)
            missing = [k for k, q in enumerate(self._buckets[i]) if q.grad is None]
            if missing:
                names = {id(q): nm for nm, q in self.module.named_parameters()}
                self._pending.clear()
                self._joining = False
                self._ready = [0] * len(self._buckets)
                raise RuntimeError(
                    "GradSync: parameter(s) %s of gradient bucket %d received no gradient in this backward pass while "
                    "others of the bucket did; every rank must produce a gradient for every parameter in every step"
                    % (", ".join(names.get(id(self._buckets[i][k]), "?") for k in missing), i))
            self._reduce(i)          # complete gradients whose hook count was short (accumulated .grad kept from before)
        for work, flat, bucket in self._pending:
            work.wait()
            if not self._avg:
                flat.div_(self.world)
            if len(bucket) > 1:
                off = 0
                for q in bucket:
                    q.grad.copy_(flat[off:off + q.numel()].view_as(q.grad))
                    off += q.numel()
        self._pending.clear()
        self._joining = False
        self._ready = [0] * len(self._buckets)


def wrap_ddp(module, device=None, force: bool = False):
    """Gradient averaging around the step module when running multi-rank (GradSync, or
    DistributedDataParallel with VAMP_GRAD_SYNC=ddp), else identity (`force` wraps a world-size-1
    group too)."""
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or force)):
        return module
    if os.environ.get("VAMP_GRAD_SYNC", "hook") != "ddp":
        return GradSync(module)
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    return DDP(module, device_ids=ids)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds: float, device="cpu") -> float:
    """The timing contract: elapsed time of the slowest rank."""
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
