"""world_size-2 gloo test (CPU) of the view-sharded data-parallel glue: the flat-bucket all-reduce must equal the
mean of the per-rank gradients, and parameters must be identical on every rank after broadcast + step."""
import os
import socket

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
