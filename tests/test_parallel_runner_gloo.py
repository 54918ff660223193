"""View-sharded data parallelism of the Runner, world size 2, gloo, on the CPU (renderer and
perceptor replaced by the CPU oracle -- the distributed logic is what is under test):
  * identical initial weights on every rank, DIFFERENT cameras / jitter / lights per rank (with the reference's
    single seed every rank would render the same view);
  * the all-reduced gradient == the mean over the same two views rendered by one process (fp32 re-association only);
  * the gradients live in ONE persistent flat bucket; weights stay identical after the step."""
import numpy as np
import torch
import torch.multiprocessing as mp

from tests.dp_common import free_port, single_process_accumulation, worker


def test_runner_two_ranks_equal_single_process_accumulation():
    world, res, spp = 2, 8, 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), out, "cpu", res, spp), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["data_seed"] + 1 == b["data_seed"]
    assert np.abs(a["eye"] - b["eye"]).max() > 1e-3, "both ranks drew the same camera"
    assert abs(a["loss"] - b["loss"]) > 1e-7
    for wa, wb in zip(a["w0"], b["w0"]):
        assert torch.equal(wa, wb)
    assert a["bucket_is_grad"] and b["bucket_is_grad"]
    for ga, gb, pa, pb in zip(a["grads"], b["grads"], a["params"], b["params"]):
        assert torch.equal(ga, gb) and torch.equal(pa, pb)
    w0, mean_grads, losses = single_process_accumulation("cpu", world, res, spp, [a["data_seed"], b["data_seed"]])
    for w, wa in zip(w0, a["w0"]):
        assert torch.equal(w, wa)
    assert abs(losses[0] - a["loss"]) < 1e-6 and abs(losses[1] - b["loss"]) < 1e-6
    gn = torch.cat([g.reshape(-1) for g in mean_grads]).norm()
    for g1, g2 in zip(mean_grads, a["grads"]):
        assert (g1 - g2).norm() <= 1e-5 * (g1.norm() + 1e-3 * gn)
