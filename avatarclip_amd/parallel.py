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
    # under a launcher (torch.distributed.run sets RANK and WORLD_SIZE) the process group is created for ONE rank too: the
    # single-rank job then runs the same broadcast / flat-bucket all-reduce path over RCCL as the 8-rank one
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world > 1:
        pin_host_threads(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if (world > 1 or launched) and not dist.is_initialized():
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


def pin_host_threads(local_rank, local_world):
    """N launch-heavy host processes on one node (639 launches per step each at 512^2): give every rank its own slice of the cores
    (intra-op torch threads = the slice, CPU affinity = the slice) instead of N x all-cores thread pools fighting each other.
    AVC_PIN_THREADS=0 leaves both alone."""
    if os.environ.get("AVC_PIN_THREADS", "1") == "0" or local_world <= 1:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:          # not on this platform
        return None
    per = max(1, len(cores) // local_world)
    mine = cores[(local_rank % local_world) * per:(local_rank % local_world + 1) * per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(len(mine), 16)))
    return mine


def is_on():
    """a process group exists (possibly of one rank): parameters are broadcast and gradients go through the flat bucket"""
    return dist.is_available() and dist.is_initialized()


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_params(params, src=0):
    for p in params:
        dist.broadcast(p.data, src=src)


def broadcast_tensor(t, src=0):
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(t, src=src)
    return t


class GradBucket:
    """One persistent flat fp32 buffer that backs the .grad of every trainable tensor (a single
    bucket of 400 065 floats = 1.6 MB for the full nets).  The gradients are views into the bucket, so the per-step
    collective is exactly one all-reduce on memory autograd already wrote -- no per-step cat / copy-back of 28 tensors.
    A parameter that received no gradient this step contributes zeros (identical layout on every rank)."""

    def __init__(self, params):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.attach()

    def attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def allreduce_mean(self):
        world = rank_world()[1]
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:            # optimizer.zero_grad(set_to_none=True) and no gradient this step
                v.zero_()
            elif g.data_ptr() != v.data_ptr():   # autograd replaced the tensor: bring it home (first step only)
                v.copy_(g)
            p.grad = v
        if is_on():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if world > 1:
                self.flat.div_(world)


def allreduce_grads(params, world=None):
    """Average the gradients of `params` over all ranks (stateless form: one temporary flat bucket; Runner keeps a
    persistent GradBucket instead)."""
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
