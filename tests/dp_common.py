"""Shared worker of the view-sharded data-parallel Runner tests: N ranks, each running ONE
Runner.train_clip_iteration on its own camera view, vs one process accumulating the same N views."""
import os
import socket

import numpy as np
import torch


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build_runner(device, res, spp, use_oracle_kernels):
    """small nets, seeded (train.seed = 0 in bench.make_conf); on the CPU the renderer / perceptor are the oracle, so that the
    distributed logic can be tested without a GPU"""
    import bench
    from avatarclip_amd.runner import Runner, EllipsoidPrior, clip_vit_random_state_dict
    conf = bench.make_conf(res, spp, small=True)
    conf.put("train.warm_up_end", 0)
    r = Runner(None, mode="train_clip", conf=conf, device=device)
    clip_sd = clip_vit_random_state_dict(0)
    if use_oracle_kernels:
        from oracle import clip_vit_oracle as C
        from oracle import neus_oracle as O

        class OraclePerceptor:
            def encode_image(self, x):
                return C.encode_image(clip_sd, x)
        r.init_clip(perceptor=OraclePerceptor())

        def oracle_render(rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0):
            jitter = torch.rand(rays_o.shape[0], 1)           # torch's per-rank CPU stream (renderer.py:318)
            return O.render(dict(r.sdf_network.named_parameters()), dict(r.color_network.named_parameters()),
                            r.deviation_network.variance, rays_o, rays_d, near, far, spp // 2, spp // 2, 4, jitter, background_rgb,
                            cos_anneal_ratio)
        r.renderer.render = oracle_render
    else:
        r.init_clip(clip_state_dict=clip_sd)
    r.init_smpl(EllipsoidPrior(device=device))
    r.update_learning_rate()
    return r


def worker(rank, world, port, out, device_kind, res, spp, backend="gloo"):
    """backend "gloo": collectives on the host, every rank on device 0 (1-GPU boxes, CPU tests); backend "nccl" (= RCCL): one device per
    rank, the product's layout on a node"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "gloo":
        os.environ["AVC_DIST_BACKEND"] = "gloo"
        if device_kind == "cuda":   # all ranks share device 0: keep the hardware queues from being oversubscribed (see bench.py's spawner)
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    else:
        os.environ.pop("AVC_DIST_BACKEND", None)
        os.environ.pop("AVC_SINGLE_DEVICE", None)
    from avatarclip_amd import parallel
    parallel.init_from_env(backend=backend)
    if backend == "nccl":
        assert device_kind == "cuda" and torch.distributed.get_backend() == "nccl" and torch.cuda.device_count() >= world
        device = torch.device("cuda", rank)
        torch.cuda.set_device(device)
    else:
        device = torch.device("cuda", 0) if device_kind == "cuda" else torch.device("cpu")   # all ranks share device 0 (development aid)
    r = build_runner(device, res, spp, use_oracle_kernels=(device_kind != "cuda"))
    assert r.world == world and r.rank == rank and r.grad_bucket is not None
    w0 = [p.detach().cpu().clone() for p in r.params_to_train]
    loss = r.train_clip_iteration(0)
    v = r.last_view
    res_ = dict(eye=np.asarray(v.eye), at=np.asarray(v.at), loss=float(loss), data_seed=r.data_seed, w0=w0,
                grads=[p.grad.detach().cpu().clone() for p in r.params_to_train],
                params=[p.detach().cpu().clone() for p in r.params_to_train],
                bucket_is_grad=all(p.grad.data_ptr() == g.data_ptr() for p, g in zip(r.params_to_train, r.grad_bucket.views)),
                device=str(device), backend=torch.distributed.get_backend(), ranks_seen=torch.distributed.get_world_size())
    if isinstance(out, str):    # a directory: one file per rank (eight ranks answering a Manager at once overran its listener)
        torch.save(res_, os.path.join(out, "rank%d.pt" % rank))
    else:
        out[rank] = res_
    parallel.barrier()
    torch.distributed.destroy_process_group()


def single_process_accumulation(device_kind, world, res, spp, seeds):
    """the same `world` views rendered one after the other by ONE process, gradients averaged (SURVEY.md 8e parity check)"""
    import random
    device = torch.device("cuda", 0) if device_kind == "cuda" else torch.device("cpu")
    r = build_runner(device, res, spp, use_oracle_kernels=(device_kind != "cuda"))
    w0 = [p.detach().cpu().clone() for p in r.params_to_train]
    acc, losses = None, []
    for rank in range(world):
        if rank > 0:   # what Runner.seed_data_rngs does on rank r > 0 (rank 0 keeps the stream the constructor seeded)
            np.random.seed(seeds[rank]); random.seed(seeds[rank]); torch.manual_seed(seeds[rank])
            if device_kind == "cuda":
                torch.cuda.manual_seed(seeds[rank])
        loss, _ = r.clip_loss(0)
        r.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        g = [torch.zeros_like(p).cpu() if p.grad is None else p.grad.detach().cpu().clone() for p in r.params_to_train]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        losses.append(float(loss))
    return w0, [a / world for a in acc], losses
