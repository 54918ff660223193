"""View-sharded data parallelism: one process per GPU, every rank renders its own camera view(s),
ONE flat-bucket all-reduce of the ~400k fp32 gradients per step (1.6 MB: latency-bound on xGMI, so a single bucket
and no overlap machinery), identical Adam step on every rank.  backend "nccl" is RCCL on ROCm; "gloo" is used by
the CPU tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # AVC_DIST_BACKEND=gloo: development aid (two ranks on ONE GPU to exercise the multi-rank path where RCCL
            # refuses duplicate devices); the product default on GPUs is "nccl" = RCCL
            backend = os.environ.get("AVC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_params(params, src=0):
    for p in params:
        dist.broadcast(p.data, src=src)


def allreduce_grads(params, world=None):
    """Average the gradients of `params` over all ranks with one flat bucket.  Parameters whose grad is None on this
    rank contribute zeros (the bucket layout must be identical on every rank)."""
    if world is None:
        world = rank_world()[1]
    if world <= 1:
        return
    grads = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
