"""world_size-2 gloo test (CPU) of the view-sharded data-parallel glue: the flat-bucket all-reduce must equal the
mean of the per-rank gradients, and parameters must be identical on every rank after broadcast + step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from avatarclip_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.rank_world() == (rank, world)
    torch.manual_seed(100 + rank)            # different init per rank on purpose
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Softplus(beta=100), torch.nn.Linear(7, 3))
    params = list(net.parameters())
    parallel.broadcast_params(params)        # -> rank 0's weights everywhere
    opt = torch.optim.Adam(params, lr=1e-2)
    g = torch.Generator().manual_seed(7 + rank)   # each rank has its own "view"
    x = torch.randn(16, 5, generator=g)
    loss = net(x).pow(2).mean()
    opt.zero_grad()
    loss.backward()
    local = [p.grad.clone() for p in params]
    params[1].grad = None if rank == 1 else params[1].grad   # a rank without a grad contributes zeros
    parallel.allreduce_grads(params, world)
    opt.step()
    t = parallel.max_over_ranks(float(rank), torch.device("cpu"))
    out[rank] = dict(local=local, reduced=[p.grad.clone() for p in params], params=[p.detach().clone() for p in params], mx=t)
    parallel.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    for i in range(len(a["local"])):
        lb = torch.zeros_like(b["local"][i]) if i == 1 else b["local"][i]
        mean = (a["local"][i] + lb) / 2
        assert torch.allclose(a["reduced"][i], mean, atol=1e-7) and torch.allclose(b["reduced"][i], mean, atol=1e-7)
        assert torch.equal(a["params"][i], b["params"][i])
    assert a["mx"] == 1.0 and b["mx"] == 1.0


# ---------------------------------------------------------------------------------------------------------------------
# host-core slices of the ranks of one node (parallel.core_slice / gpu_numa_nodes / numa_cpus) on FAKE topologies: the product reads
# /sys, the rule itself is a pure function
def _fake_sysfs(tmp_path, gpu_nodes, node_cpulists):
    drv = tmp_path / "bus" / "pci" / "drivers" / "amdgpu"
    for i, node in enumerate(gpu_nodes):
        d = drv / ("0000:%02x:00.0" % (0x10 + 0x10 * i))
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    (drv / "module").mkdir()          # non-device entries of the driver directory are ignored
    for n, text in node_cpulists.items():
        nd = tmp_path / "devices" / "system" / "node" / ("node%d" % n)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(text + "\n")
    return str(tmp_path)


def test_core_slices_follow_the_numa_node_of_each_rank_s_gpu(tmp_path, monkeypatch):
    from avatarclip_amd import parallel
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    # an 8-GPU, two-socket node: GPUs 0-3 on node 0, 4-7 on node 1; node 0 = cores 0-63 + SMT siblings 128-191, node 1 = the rest
    root = _fake_sysfs(tmp_path, [0, 0, 0, 0, 1, 1, 1, 1], {0: "0-63,128-191", 1: "64-127,192-255"})
    nodes = parallel.gpu_numa_nodes(root)
    assert nodes == [0, 0, 0, 0, 1, 1, 1, 1]
    cpus = {n: parallel.numa_cpus(n, root) for n in (0, 1)}
    assert len(cpus[0]) == 128 and cpus[0][:2] == [0, 1] and cpus[0][64] == 128
    allowed = list(range(256))
    slices = [parallel.core_slice(r, 8, allowed, nodes, cpus) for r in range(8)]
    assert all(len(s) == 32 for s in slices)
    assert len(set(c for s in slices for c in s)) == 256, "the slices partition the node's cores"
    for r, s in enumerate(slices):
        assert set(s) <= set(cpus[0 if r < 4 else 1]), "rank %d left the socket of its GPU" % r
    # the index rule would have put rank 2 on cores 64-95 (socket 1) although its GPU hangs off socket 0
    assert parallel.core_slice(2, 8, allowed)[0] == 64 and slices[2][0] in cpus[0]
    # interleaved enumeration (GPU i on node i % 2), 4 ranks
    nodes2 = [0, 1, 0, 1]
    s2 = [parallel.core_slice(r, 4, allowed, nodes2, cpus) for r in range(4)]
    assert set(s2[0]) | set(s2[2]) == set(cpus[0]) and set(s2[1]) | set(s2[3]) == set(cpus[1])
    # a restricted affinity mask (cgroup): only cores the process may use are handed out
    s3 = parallel.core_slice(5, 8, list(range(64, 96)), nodes, cpus)
    assert set(s3) <= set(range(64, 96)) and len(s3) == 8
    # unknown topology (-1), missing /sys, or a node none of whose cores is allowed: the index slice
    assert parallel.core_slice(3, 8, allowed, [-1] * 8, cpus) == list(range(96, 128))
    assert parallel.gpu_numa_nodes(str(tmp_path / "nowhere")) == []
    assert parallel.core_slice(1, 2, list(range(8)), [], {}) == [4, 5, 6, 7]
    assert parallel.core_slice(0, 8, list(range(200, 208)), nodes, cpus) == [200]
    # HIP_VISIBLE_DEVICES re-indexes the devices a process sees
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5")
    assert parallel.gpu_numa_nodes(root) == [1, 1]
    # ... and composes with the runtime-level filter it indexes into (ROCR first): devices 2..7 -> of those, 1 and 3 = physical 3 and 5
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,3,4,5,6,7")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,3")
    assert parallel.gpu_numa_nodes(root) == [0, 1]


def test_queue_oversubscription_is_reported_only_when_ranks_share_a_device(monkeypatch):
    from avatarclip_amd import parallel
    said = []
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert parallel.check_queue_oversubscription(8, 8, said.append) is None          # one process per GPU: the product's layout
    assert parallel.check_queue_oversubscription(4, 1, said.append) is None          # 4 x 4 queues fit
    msg = parallel.check_queue_oversubscription(8, 1, said.append)                  # the round-4 rig: 8 x 4 queues on one device
    assert msg and "GPU_MAX_HW_QUEUES=2" in msg and said == [msg]
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    assert parallel.check_queue_oversubscription(8, 1, said.append) is None
    assert parallel.check_queue_oversubscription(8, 0, said.append) is None          # no device visible (CPU tests)


def test_replica_layout_and_environment(tmp_path, monkeypatch):
    """avatarclip_amd.replicas (BASELINE config 4): device round-robin + NUMA core slices (pure), and the children's environment -- one
    visible device each, no rank variables, the hardware-queue cap when replicas share a device, logs, exit codes."""
    import json
    import sys
    from avatarclip_amd import replicas
    cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    lay = replicas.plan(8, list(range(8)), list(range(128)), [0, 0, 0, 0, 1, 1, 1, 1], cpus)
    assert [g for g, _ in lay] == list(range(8))
    assert all(len(c) == 16 for _, c in lay) and set(lay[5][1]) <= set(cpus[1]) and set(lay[2][1]) <= set(cpus[0])
    lay = replicas.plan(4, [2, 3], list(range(128)), [0, 0, 1, 1], cpus)              # four replicas on two devices
    assert [g for g, _ in lay] == [2, 3, 2, 3] and all(set(c) <= set(cpus[1]) for _, c in lay) and len(set(sum((c for _, c in lay), []))) == 64
    assert replicas.plan(0, [0], [0]) == []
    prog = "import os, json, sys; json.dump({k: os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'RANK', 'WORLD_SIZE', 'AVC_REPLICA', 'GPU_MAX_HW_QUEUES')}, open(sys.argv[1], 'w')); sys.exit(int(sys.argv[2]))"
    outs = [str(tmp_path / ("r%d.json" % i)) for i in range(9)]
    cmds = [[sys.executable, "-c", prog, outs[i], "3" if i == 4 else "0"] for i in range(9)]
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    rcs = replicas.launch_commands(cmds, gpus=[0, 1], log_dir=str(tmp_path / "logs"), env=dict(os.environ, RANK="3", WORLD_SIZE="8"))
    assert rcs == [0, 0, 0, 0, 3, 0, 0, 0, 0]
    for i in range(9):
        e = json.load(open(outs[i]))
        assert e["HIP_VISIBLE_DEVICES"] == str(i % 2) and e["RANK"] is None and e["WORLD_SIZE"] is None and e["AVC_REPLICA"] == str(i)
        assert e["GPU_MAX_HW_QUEUES"] == "2"          # 5 processes x 4 queues on one device would oversubscribe it
        assert os.path.exists(str(tmp_path / "logs" / ("replica_%d.log" % i)))
    # a restricted parent (a scheduler's HIP_VISIBLE_DEVICES=2,3 / ROCR_VISIBLE_DEVICES=4,5,6,7): replica ordinals are relative to the
    # parent's visible set, so the children get the parent's entries, and the runtime-level filter stays in place
    assert replicas.child_visibility(1, {"HIP_VISIBLE_DEVICES": "2,3"}) == ({"HIP_VISIBLE_DEVICES": "3"}, ("CUDA_VISIBLE_DEVICES",))
    assert replicas.child_visibility(0, {"CUDA_VISIBLE_DEVICES": "5, 6"})[0] == {"HIP_VISIBLE_DEVICES": "5"}
    assert replicas.child_visibility(2, {"ROCR_VISIBLE_DEVICES": "4,5,6,7"})[0] == {"HIP_VISIBLE_DEVICES": "2"}
    with pytest.raises(ValueError):
        replicas.child_visibility(2, {"HIP_VISIBLE_DEVICES": "2,3"})
    prog2 = "import os, json, sys; json.dump({k: os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')}, open(sys.argv[1], 'w'))"
    cmds = [[sys.executable, "-c", prog2, outs[i]] for i in range(2)]
    assert replicas.launch_commands(cmds, gpus=[0, 1], env=dict(os.environ, ROCR_VISIBLE_DEVICES="4,5,6,7", HIP_VISIBLE_DEVICES="2,3", CUDA_VISIBLE_DEVICES="2,3")) == [0, 0]
    for i in range(2):
        assert json.load(open(outs[i])) == {"HIP_VISIBLE_DEVICES": str(2 + i), "ROCR_VISIBLE_DEVICES": "4,5,6,7", "CUDA_VISIBLE_DEVICES": None}
    assert replicas.main(["--confs", "/nonexistent/a.conf", "--gpus", "0", "--log_dir", str(tmp_path / "l2"), "--mode", "train"]) == 1
