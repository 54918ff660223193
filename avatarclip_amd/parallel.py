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
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        pin_host_threads(local_rank, local_world)
        check_queue_oversubscription(local_world)
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


def _parse_cpulist(text):
    """'0-63,128-191' -> [0..63, 128..191] (the kernel's cpulist format)"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every amdgpu device in HIP's default enumeration order (PCI bus order of the render-capable amdgpu functions
    under /sys/bus/pci/drivers/amdgpu; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES index lists are applied on top).  -1 = unknown.
    Returns [] where the topology is not readable (containers without /sys, other platforms)."""
    drv = os.path.join(sysfs, "bus", "pci", "drivers", "amdgpu")
    try:
        devs = sorted(d for d in os.listdir(drv) if d.count(":") == 2)
    except OSError:
        return []
    nodes = []
    for d in devs:
        try:
            with open(os.path.join(drv, d, "numa_node")) as f:
                nodes.append(int(f.read().strip()))
        except (OSError, ValueError):
            nodes.append(-1)
    # ROCR_VISIBLE_DEVICES filters what the runtime enumerates; HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is its alias) then indexes
    # into THAT list -- the two compose, in this order
    for var in ("ROCR_VISIBLE_DEVICES", ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        vis = os.environ.get(var) if isinstance(var, str) else (os.environ.get(var[0]) or os.environ.get(var[1]))
        if vis and all(t.strip().isdigit() for t in vis.split(",")):
            idx = [int(t) for t in vis.split(",")]
            if all(i < len(nodes) for i in idx):
                nodes = [nodes[i] for i in idx]
    return nodes


def numa_cpus(node, sysfs="/sys"):
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError):
        return []


def core_slice(local_rank, local_world, allowed, gpu_nodes=None, node_cpus=None):
    """The host cores of rank `local_rank`: the cores of the NUMA node its GPU hangs off, divided evenly among the ranks whose GPUs
    share that node (a launch-heavy host process on the far socket pays a cross-socket hop on every doorbell and every pinned-buffer
    write); without a readable topology -- or when the node's cores are not in this process's affinity mask -- the r-th of
    `local_world` equal slices of the allowed cores, by index.  Pure function of its arguments (tests/test_parallel_gloo.py runs it on
    fake topologies): gpu_nodes[r] = NUMA node of rank r's device or -1, node_cpus[n] = cores of node n."""
    allowed = sorted(allowed)
    r = local_rank % max(1, local_world)
    if gpu_nodes and node_cpus and r < len(gpu_nodes) and gpu_nodes[r] >= 0:
        node = gpu_nodes[r]
        cand = [c for c in node_cpus.get(node, []) if c in set(allowed)]
        peers = [q for q in range(min(local_world, len(gpu_nodes))) if gpu_nodes[q] == node]
        if cand and r in peers:
            per = len(cand) // len(peers)
            if per >= 1:
                k = peers.index(r)
                return cand[k * per:(k + 1) * per]
    per = max(1, len(allowed) // max(1, local_world))
    return allowed[r * per:(r + 1) * per] or allowed


def pin_host_threads(local_rank, local_world):
    """N launch-heavy host processes on one node (~245 launches per step each at 512^2): give every rank its own slice of the cores
    (intra-op torch threads = the slice, CPU affinity = the slice) instead of N x all-cores thread pools fighting each other -- the
    slice taken from the NUMA node of the rank's GPU (core_slice).  AVC_PIN_THREADS=0 leaves both alone."""
    if os.environ.get("AVC_PIN_THREADS", "1") == "0" or local_world <= 1:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:          # not on this platform
        return None
    nodes = gpu_numa_nodes()
    mine = core_slice(local_rank, local_world, cores, nodes, {n: numa_cpus(n) for n in set(nodes) if n >= 0})
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(len(mine), 16)))
    return mine


def check_queue_oversubscription(local_world, n_devices=None, warn=None):
    """More ranks than devices on a node (a shared node, or the single-device rig of the tests): every process opens GPU_MAX_HW_QUEUES
    hardware queues (default 4) on the device it shares; beyond the device's queue slots the scheduler time-slices wavefronts by
    context save / restore, and under that a launch died with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in 2 of 5 runs with 8 processes on
    one MI355X (never with <= 6, never with GPU_MAX_HW_QUEUES <= 2).  Returns the warning text (and emits it) when
    the process is in that regime and has not capped its queues; None otherwise.  One process per GPU -- the product's layout -- never is."""
    if n_devices is None:
        n_devices = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_devices <= 0 or local_world <= n_devices:
        return None
    per_dev = -(-local_world // n_devices)
    try:
        q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        q = 4
    if per_dev * q <= 16:
        return None
    msg = ("avatarclip_amd: %d ranks share %d device(s) (%d processes x %d hardware queues per device): wavefront preemption under "
           "queue oversubscription has crashed launches on MI355X -- set GPU_MAX_HW_QUEUES=2 (or run one rank per GPU)"
           % (local_world, n_devices, per_dev, q))
    if warn is None:
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    else:
        warn(msg)
    return msg


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
