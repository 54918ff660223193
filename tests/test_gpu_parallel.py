"""SURVEY.md section 8e parity check on the HIP path: two ranks (gloo collectives, both on GPU 0 -- a development aid, RCCL
refuses duplicate devices) each run Runner.train_clip_iteration on their own view; the all-reduced gradient must equal the
mean of the same two views rendered one after the other by a single process, up to fp32 re-association of the flat bucket."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.dp_common import free_port, single_process_accumulation, worker

gpu = pytest.mark.gpu


@gpu
def test_two_ranks_on_the_hip_path_equal_single_gpu_accumulation():
    world, res, spp = 2, 32, 32
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), out, "cuda", res, spp), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert np.abs(a["eye"] - b["eye"]).max() > 1e-3, "both ranks drew the same camera"
    assert a["bucket_is_grad"] and b["bucket_is_grad"]
    for wa, wb, ga, gb, pa, pb in zip(a["w0"], b["w0"], a["grads"], b["grads"], a["params"], b["params"]):
        assert torch.equal(wa, wb) and torch.equal(ga, gb) and torch.equal(pa, pb)
    w0, mean_grads, losses = single_process_accumulation("cuda", world, res, spp, [a["data_seed"], b["data_seed"]])
    assert abs(losses[0] - a["loss"]) < 1e-5 and abs(losses[1] - b["loss"]) < 1e-5, (losses, a["loss"], b["loss"])
    gn = torch.cat([g.reshape(-1) for g in mean_grads]).norm()
    for g1, g2 in zip(mean_grads, a["grads"]):
        assert (g1 - g2).norm() <= 1e-4 * (g1.norm() + 1e-3 * gn)
