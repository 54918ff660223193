"""SURVEY.md section 8e parity check on the HIP path.  On a node with >= 2 devices: RCCL, one rank per device (the tests below that
skip themselves on a 1-GPU box).  Everywhere: two / eight ranks (gloo collectives, all on GPU 0 -- a development aid, RCCL
refuses duplicate devices) each run Runner.train_clip_iteration on their own view; the all-reduced gradient must equal the
mean of the same two views rendered one after the other by a single process, up to fp32 re-association of the flat bucket."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.dp_common import free_port, single_process_accumulation, worker

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("world", [2, 8])
def test_ranks_on_the_hip_path_equal_single_gpu_accumulation(world, tmp_path):
    """world = 8 is the size BASELINE config 3 runs at: eight processes, eight libavc loads and engines on one device, eight different
    cameras, one flat bucket whose mean must equal one process accumulating the same eight views (SURVEY 8e's parity statement)."""
    import os
    res, spp = 32, 32
    mp.spawn(worker, args=(world, free_port(), str(tmp_path), "cuda", res, spp), nprocs=world, join=True)
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % k), weights_only=False) for k in range(world)]
    a = rs[0]
    eyes = np.stack([r["eye"] for r in rs])
    d = np.abs(eyes[:, None] - eyes[None]).max(-1) + np.eye(world)
    assert d.min() > 1e-3, "two ranks drew the same camera"
    assert len({r["data_seed"] for r in rs}) == world
    for b in rs:
        assert b["bucket_is_grad"]
        for wa, wb, ga, gb, pa, pb in zip(a["w0"], b["w0"], a["grads"], b["grads"], a["params"], b["params"]):
            assert torch.equal(wa, wb) and torch.equal(ga, gb) and torch.equal(pa, pb)
    w0, mean_grads, losses = single_process_accumulation("cuda", world, res, spp, [r["data_seed"] for r in rs])
    for k in range(world):
        assert abs(losses[k] - rs[k]["loss"]) < 1e-5, (k, losses, rs[k]["loss"])
    gn = torch.cat([g.reshape(-1) for g in mean_grads]).norm()
    for g1, g2 in zip(mean_grads, a["grads"]):
        assert (g1 - g2).norm() <= 1e-4 * (g1.norm() + 1e-3 * gn)


@gpu
@pytest.mark.parametrize("world", [2, 8])
def test_bench_spawns_its_own_ranks_and_reports_the_whole_job(world):
    """`python bench.py --gpus N` with no WORLD_SIZE starts the N ranks itself (the driver's contract); all ranks share device 0 here
    (AVC_SINGLE_DEVICE + gloo, the development path for a 1-GPU box).  N = 8 = the driver's largest run: eight host processes, eight
    CLIP graph captures, eight hipFuncSetAttribute paths at once.  The JSON line must say n_gpus = N and count the rays of ALL ranks;
    without AVC_SINGLE_DEVICE the same command must refuse to run on a box with one device."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVC_SINGLE_DEVICE="1", AVC_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--res", "64", "--small", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["steps"] == 2
    assert out["config"]["collective"] == "gloo, %d rank(s)" % world
    assert abs(out["value"] - world * 64 * 64 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    if torch.cuda.device_count() < world and world == 2:
        env2 = dict(os.environ); env2.pop("AVC_SINGLE_DEVICE", None); env2.pop("WORLD_SIZE", None); env2.pop("RANK", None)
        r2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=120, cwd=root)
        assert r2.returncode != 0 and "device" in (r2.stderr + r2.stdout)


def _n_devices():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@gpu
@pytest.mark.skipif(_n_devices() < 2, reason="RCCL refuses two ranks on one device: needs a node with >= 2 MI355X (runs by itself on one)")
def test_rccl_ranks_one_device_each_equal_single_gpu_accumulation(tmp_path):
    """SURVEY 8e's parity statement on the PRODUCT's layout: world = min(8, devices) ranks, one MI355X each, backend "nccl" = RCCL over
    xGMI -- broadcast of the weights, per-rank views, ONE all-reduce of the flat gradient bucket (small nets here) -- must give
    every rank the mean of the gradients ONE process computes for the same views on device 0.  The assertions are the gloo test's;
    nothing here is specific to a device count, so the first multi-GPU lease produces the evidence without a code change."""
    import os
    world = min(8, _n_devices())
    res, spp = 32, 32
    mp.spawn(worker, args=(world, free_port(), str(tmp_path), "cuda", res, spp, "nccl"), nprocs=world, join=True)
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % k), weights_only=False) for k in range(world)]
    assert [r["device"] for r in rs] == ["cuda:%d" % k for k in range(world)]
    assert all(r["backend"] == "nccl" and r["ranks_seen"] == world and r["bucket_is_grad"] for r in rs)
    assert len({r["data_seed"] for r in rs}) == world
    a = rs[0]
    for b in rs:
        for wa, wb, ga, gb, pa, pb in zip(a["w0"], b["w0"], a["grads"], b["grads"], a["params"], b["params"]):
            assert torch.equal(wa, wb) and torch.equal(ga, gb) and torch.equal(pa, pb)
    w0, mean_grads, losses = single_process_accumulation("cuda", world, res, spp, [r["data_seed"] for r in rs])
    for k in range(world):
        assert abs(losses[k] - rs[k]["loss"]) < 1e-5, (k, losses, rs[k]["loss"])
    gn = torch.cat([g.reshape(-1) for g in mean_grads]).norm()
    for g1, g2 in zip(mean_grads, a["grads"]):
        assert (g1 - g2).norm() <= 1e-4 * (g1.norm() + 1e-3 * gn)
    print("RCCL world %d: all-reduced bucket == single-process mean" % world)


@gpu
@pytest.mark.skipif(_n_devices() < 2, reason="needs a node with >= 2 MI355X (runs by itself on one)")
def test_bench_over_rccl_on_every_device_of_the_node():
    """`python bench.py --gpus N` on a real node, N = min(8, devices): the line must say that RCCL ran with N ranks on N DIFFERENT
    devices (`rccl_ranks_seen` and `process_group` come from the process group, not from the flags) and count the rays of all of them."""
    import json
    import os
    import subprocess
    import sys
    world = min(8, _n_devices())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AVC_DIST_BACKEND", "AVC_SINGLE_DEVICE"):
        env.pop(k, None)
    env["AVC_ASSERT_DIST"] = "nccl"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "2", "--res", "128", "--small", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == world and out["config"]["collective"] == "nccl (RCCL), %d rank(s)" % world
    assert out["rccl_ranks_seen"] == world and sorted(out["process_group"]["devices"]) == list(range(world))
    assert abs(out["value"] - world * 128 * 128 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]


@gpu
def test_pinned_upload_ring_survives_wraparound():
    """h2d.upload: more uploads than the ring has rows, values checked AFTER all of them were enqueued (a row must not be reused
    before its copy has completed)"""
    import numpy as np
    import torch
    from avatarclip_amd import h2d
    outs = [h2d.upload(np.full((4, 4), float(i)), "cuda") for i in range(300)]
    torch.cuda.synchronize()
    assert all(float(o[1, 2]) == float(i) and o.shape == (4, 4) for i, o in enumerate(outs))
    big = h2d.upload(np.arange(1000.0), "cuda")          # larger than a ring row: plain path
    assert float(big[999]) == 999.0


def _rccl_single_rank_worker(proc_index, port, out):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.pop("AVC_DIST_BACKEND", None)
    import torch
    import torch.distributed as dist
    from avatarclip_amd import parallel
    rank, world, local_rank = parallel.init_from_env(backend="nccl")
    assert (rank, world, local_rank) == (0, 1, 0) and parallel.is_on() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    params = [torch.nn.Parameter(torch.randn(257, 256, device=dev)), torch.nn.Parameter(torch.randn(400065 - 257 * 256, device=dev))]
    before = [p.detach().clone() for p in params]
    parallel.broadcast_params(params)                       # ncclBroadcast on the parameter storage
    bucket = parallel.GradBucket(params)                    # one flat fp32 buffer of 400 065 floats = the full nets' gradient
    for p in params:
        p.grad.copy_(torch.arange(p.numel(), device=dev, dtype=torch.float32).view_as(p) * 1e-3)
    want = bucket.flat.clone()
    bucket.allreduce_mean()                                 # ncclAllReduce(SUM) over the bucket, one rank: identity
    t = parallel.max_over_ranks(12.5, dev)
    parallel.barrier()
    torch.cuda.synchronize()
    out["same_params"] = all(torch.equal(a, b.detach()) for a, b in zip(before, params))
    out["same_bucket"] = bool(torch.equal(want, bucket.flat))
    out["bucket_numel"] = int(bucket.flat.numel())
    out["max"] = t
    out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    dist.destroy_process_group()
    out["clean_exit"] = True


@gpu
def test_rccl_path_with_a_single_rank():
    """backend "nccl" IS RCCL on ROCm.  Every multi-rank test in this suite runs on gloo (RCCL refuses two ranks on one device), so
    this one executes the product's collective calls -- init, broadcast of the parameters, the flat-bucket all-reduce, the MAX
    reduction of the timing, barrier, teardown -- on RCCL itself with the one rank a 1-GPU box allows."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_single_rank_worker, args=(free_port(), out), nprocs=1, join=True)
    assert out["same_params"] and out["same_bucket"] and out["bucket_numel"] == 400065 and out["max"] == 12.5 and out["clean_exit"]
    print("RCCL", out["nccl_version"])


@gpu
def test_bench_single_rank_under_the_launcher_runs_the_rccl_path():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launcher's environment makes the Runner broadcast
    its weights and all-reduce its gradient bucket over RCCL (one rank), i.e. the 8-GPU code path end to end on one GPU"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AVC_DIST_BACKEND", "AVC_SINGLE_DEVICE"):
        env.pop(k, None)
    env["AVC_ASSERT_DIST"] = "nccl"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--res", "64", "--small", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["config"]["collective"] == "nccl (RCCL), 1 rank(s)"
    assert out["rccl_ranks_seen"] == 1 and out["process_group"] == {"backend": "nccl", "ranks": 1, "devices": [0]}


@gpu
def test_two_replicas_equal_their_solo_runs(tmp_path):
    """BASELINE config 4 (replicas only): two independent avatars -- their own seeds, hence their own prompt stand-ins, cameras
    and jitter -- started side by side by avatarclip_amd.replicas (no process group; both on device 0 here, one per GPU on a node) must
    produce exactly the losses each produces when it runs alone: nothing is shared but the device."""
    import json
    import os
    import sys
    from avatarclip_amd import replicas
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = os.path.join(root, "tests", "replica_child.py")

    def cmds(tag):
        return [[sys.executable, child, str(seed), "3", os.path.join(str(tmp_path), "%s_%d.json" % (tag, seed))] for seed in (1, 2)]

    env = dict(os.environ, RANK="0", WORLD_SIZE="1")        # a launcher's leftovers must not turn replicas into ranks
    rcs = replicas.launch_commands(cmds("pair"), gpus=[0], log_dir=str(tmp_path / "logs"), env=env)
    assert rcs == [0, 0], [open(os.path.join(str(tmp_path), "logs", "replica_%d.log" % i)).read()[-1500:] for i in range(2)]
    for c in cmds("solo"):
        assert replicas.launch_commands([c], gpus=[0], log_dir=str(tmp_path / "logs_solo"))[0] == 0
    for seed in (1, 2):
        pair = json.load(open(os.path.join(str(tmp_path), "pair_%d.json" % seed)))
        solo = json.load(open(os.path.join(str(tmp_path), "solo_%d.json" % seed)))
        print(seed, pair, solo)
        assert pair["visible"] == "0" and pair["replica"] == str(seed - 1)
        assert all(np.isfinite(pair["losses"])) and len(pair["losses"]) == 3
        assert np.allclose(pair["losses"], solo["losses"], rtol=1e-6, atol=0), (pair["losses"], solo["losses"])
    a = json.load(open(os.path.join(str(tmp_path), "pair_1.json")))["losses"]
    b = json.load(open(os.path.join(str(tmp_path), "pair_2.json")))["losses"]
    assert abs(a[0] - b[0]) > 1e-6, "the two replicas are supposed to be different avatars"
