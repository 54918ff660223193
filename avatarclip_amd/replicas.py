"""BASELINE config 4 -- "batch of 8 independent text prompts, one avatar per GPU": N INDEPENDENT runs, no process group, no
collective (replicas only; the reference has no multi-GPU mode at all, main.py:958,963).

    python -m avatarclip_amd.replicas --confs confs/a.conf confs/b.conf ... [--gpus 0,1,..] [--mode train_clip] [--log_dir DIR] [-- extra main.py flags]

starts one `python -m avatarclip_amd.main --mode MODE --conf X` per conf, replica i on GPU gpus[i % len(gpus)] (made the process's
only visible device), with its own slice of the host cores taken from the NUMA node of that GPU (parallel.core_slice: the same rule
the view-sharded ranks use), stdout / stderr in <log_dir>/replica_<i>.log, and waits for all of them.  Exit status = the number of
replicas that failed.  More replicas than devices is allowed (they then share a device and its memory); the hardware-queue cap that
this needs on MI355X (parallel.check_queue_oversubscription) is put into the children's environment.
"""
import argparse
import os
import subprocess
import sys

from . import parallel


def plan(n, gpus, allowed_cores, gpu_nodes=None, node_cpus=None):
    """[(gpu, cores)] for n replicas on the device list `gpus`: replica i -> gpus[i % len(gpus)], the cores of that GPU's NUMA node
    divided among the replicas that land on devices of the same node (index slices when the topology is unknown).  Pure function."""
    if n <= 0:
        return []
    if not gpus:
        raise ValueError("no devices")
    dev = [gpus[i % len(gpus)] for i in range(n)]
    # core_slice's view: "rank" i of n, its device's node
    nodes = None
    if gpu_nodes:
        nodes = [gpu_nodes[d] if 0 <= d < len(gpu_nodes) else -1 for d in dev]
    return [(dev[i], parallel.core_slice(i, n, allowed_cores, nodes, node_cpus)) for i in range(n)]


def child_visibility(gpu, env):
    """The device-visibility variables of the child that owns ordinal `gpu` OF THE PARENT'S VISIBLE SET (what torch.cuda.device_count()
    counts and parallel.gpu_numa_nodes() describes).  HIP_VISIBLE_DEVICES (alias CUDA_VISIBLE_DEVICES) indexes into what
    ROCR_VISIBLE_DEVICES leaves, so the parent's ROCR list is KEPT and the parent's HIP list is translated: under a scheduler's
    ROCR_VISIBLE_DEVICES=4,5,6,7 or HIP_VISIBLE_DEVICES=2,3 replica 0 lands on physical GPU 4 / 2, not on somebody else's GPU 0.
    Pure function: returns (updates, names to remove)."""
    vis = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
    if vis:
        toks = [t.strip() for t in vis.split(",") if t.strip()]
        if not 0 <= gpu < len(toks):
            raise ValueError("replica device ordinal %d is outside the parent's visible devices %r" % (gpu, vis))
        return {"HIP_VISIBLE_DEVICES": toks[gpu]}, ("CUDA_VISIBLE_DEVICES",)
    return {"HIP_VISIBLE_DEVICES": str(gpu)}, ("CUDA_VISIBLE_DEVICES",)


def launch_commands(commands, gpus=None, log_dir=None, env=None, wait=True):
    """Start one child per command (argv lists), replica i pinned to plan()[i].  Returns the list of return codes (wait=True) or the
    Popen objects."""
    import torch
    n = len(commands)
    if gpus is None:
        gpus = list(range(max(1, torch.cuda.device_count())))
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    nodes = parallel.gpu_numa_nodes()
    layout = plan(n, gpus, allowed, nodes, {k: parallel.numa_cpus(k) for k in set(nodes) if k >= 0})
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        base.pop(k, None)          # replicas are NOT ranks: no process group is created in the children
    if n > len(set(gpus)):
        per_dev = -(-n // len(set(gpus)))
        if per_dev * int(base.get("GPU_MAX_HW_QUEUES", "4")) > 16:
            base["GPU_MAX_HW_QUEUES"] = "2"      # queue oversubscription crashed launches with 8 processes per device
    if log_dir:
        os.makedirs(log_dir, exist_ok=True)
    procs = []
    for i, (cmd, (gpu, cores)) in enumerate(zip(commands, layout)):
        e = dict(base, AVC_REPLICA=str(i), OMP_NUM_THREADS=str(max(1, min(len(cores), 16))))
        upd, drop = child_visibility(gpu, base)
        e.update(upd)
        for k in drop:
            e.pop(k, None)
        out = open(os.path.join(log_dir, "replica_%d.log" % i), "w") if log_dir else None

        def pin(cores=cores):
            try:
                os.sched_setaffinity(0, cores)
            except (AttributeError, OSError):
                pass
        procs.append(subprocess.Popen(cmd, env=e, stdout=out, stderr=subprocess.STDOUT if out else None, preexec_fn=pin))
        if out:
            out.close()
    if not wait:
        return procs
    return [p.wait() for p in procs]


def launch(confs, gpus=None, mode="train_clip", extra_args=(), log_dir=None, env=None, wait=True):
    cmds = [[sys.executable, "-m", "avatarclip_amd.main", "--mode", mode, "--conf", c, "--gpu", "0"] + list(extra_args) for c in confs]
    return launch_commands(cmds, gpus, log_dir, env, wait)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--confs", nargs="+", required=True, help="one conf per replica (one prompt / avatar each)")
    ap.add_argument("--gpus", type=str, default=None, help="comma-separated device ordinals within this process's visible devices (default: all of them)")
    ap.add_argument("--mode", type=str, default="train_clip")
    ap.add_argument("--log_dir", type=str, default="./replica_logs")
    ap.add_argument("extra", nargs=argparse.REMAINDER, help="flags passed through to avatarclip_amd.main after `--`")
    args = ap.parse_args(argv)
    gpus = [int(t) for t in args.gpus.split(",")] if args.gpus else None
    extra = [a for a in args.extra if a != "--"]
    codes = launch(args.confs, gpus, args.mode, extra, args.log_dir)
    for i, (c, rc) in enumerate(zip(args.confs, codes)):
        print("replica %d  %s  exit %d  log %s" % (i, c, rc, os.path.join(args.log_dir, "replica_%d.log" % i)))
    return sum(1 for rc in codes if rc != 0)


if __name__ == "__main__":
    sys.exit(main())
