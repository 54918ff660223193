"""Whole-iteration parity: Runner.train_clip_iteration (main.py:345-566) on the HIP path vs the SAME host glue driving the
CPU oracle (oracle/neus_oracle.render + oracle/clip_vit_oracle.encode_image) with identical seeds, cameras, jitter and
weights.  Covers rows a1-a18 of SURVEY section 8 in one pass: rays, hierarchical sampling, the MLPs, compositing,
shading, CLIP, losses, backward, Adam.  BASELINE config 1 geometry (64 x 64 rays would be the CPU-runnable case; 32 x 32
rays x 32 spp here keeps the CPU leg at a few seconds), small nets (confs/examples_small).

Tolerances: per-iteration loss within 2e-3 (the loss is O(1): L1 + eikonal + BCE + 2 x (1 - cos)); per-tensor gradient of
the first iteration: relative L2 <= 5e-2 (against max(own norm, 1e-4 of the whole gradient)) and cosine >= 0.995 (bf16
gradient operands; includes the CLIP backward to pixels); the scalar d loss / d (sdf bias) is ill-conditioned and has its
own 10 % bound (see the comment in the test).
"""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


def _make_runner(device, res, spp):
    import bench
    from avatarclip_amd.runner import Runner
    conf = bench.make_conf(res, spp, small=True)
    conf.put("train.use_bg_aug", False)
    conf.put("train.warm_up_end", 0)
    torch.manual_seed(0)
    r = Runner(None, mode="train_clip", conf=conf, device=device)
    return r


@gpu
def test_train_clip_iteration_matches_oracle_driven_iteration():
    from oracle import clip_vit_oracle as C
    from oracle import neus_oracle as O
    from avatarclip_amd.runner import clip_vit_random_state_dict, EllipsoidPrior
    res, spp, iters = 32, 32, 2
    clip_sd = clip_vit_random_state_dict(0)
    jit = [torch.rand(res * res, 1, generator=torch.Generator().manual_seed(100 + i)) for i in range(iters)]

    # ---------------- product path
    dev = torch.device("cuda")
    a = _make_runner(dev, res, spp)
    a.init_clip(clip_state_dict=clip_sd)
    a.init_smpl()
    a.update_learning_rate()
    step = {"i": 0}
    a_render = a.renderer.render
    a.renderer.render = lambda *args, **kw: a_render(*args, jitter=jit[step["i"]].to(dev), **kw)

    # ---------------- the same glue on the CPU, driving the oracle
    cpu = torch.device("cpu")
    b = _make_runner(cpu, res, spp)

    class OraclePerceptor:
        def encode_image(self, x):
            return C.encode_image(clip_sd, x)

    b.init_clip(perceptor=OraclePerceptor())
    b.init_smpl(EllipsoidPrior(device=cpu))
    b.update_learning_rate()

    def oracle_render(rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0):
        sd_s = dict(b.sdf_network.named_parameters())
        sd_c = dict(b.color_network.named_parameters())
        out = O.render(sd_s, sd_c, b.deviation_network.variance, rays_o, rays_d, near, far, spp // 2, spp // 2, 4,
                       jit[step["i"]], background_rgb, cos_anneal_ratio)
        return out
    b.renderer.render = oracle_render
    # identical initial weights (both were built under torch.manual_seed(0); make it explicit)
    for pa, pb in zip(a.params_to_train, b.params_to_train):
        assert torch.equal(pa.detach().cpu(), pb.detach())

    grads_a, grads_b, losses = None, None, []
    for i in range(iters):
        step["i"] = i
        np.random.seed(1234 + i)
        la = a.train_clip_iteration(i)
        if i == 0:
            grads_a = [p.grad.detach().cpu().clone() for p in a.params_to_train]
        a.update_learning_rate()
        np.random.seed(1234 + i)
        lb = b.train_clip_iteration(i)
        if i == 0:
            grads_b = [p.grad.detach().clone() for p in b.params_to_train]
        b.update_learning_rate()
        losses.append((la.item(), lb.item()))
    print("losses (hip, oracle):", losses)
    for la, lb in losses:
        assert abs(la - lb) < 2e-3 * max(1.0, abs(lb))
    names = [n for net in (a.sdf_network, a.deviation_network, a.color_network) for n, _ in net.named_parameters()]
    worst = 0.0
    sdf_last_bias = [n for n, _ in a.sdf_network.named_parameters() if n.endswith(".bias")][-1]
    seen_sdf = False
    gnorm = torch.cat([g.reshape(-1) for g in grads_b]).double().norm().item()
    for n, ga, gb in zip(names, grads_a, grads_b):
        if gb.abs().max() < 1e-7:
            continue
        if n == sdf_last_bias and not seen_sdf:
            # d loss / d (sdf bias) = sum over all points of d loss / d sdf: the eikonal and the alpha terms pull both ways and
            # cancel to ~1e-3 of sum |d_sdf|, so 1e-3-level forward differences (f16 operands) move this ONE scalar by a few
            # percent.  It is checked on its own with a wider bound; the other 128 entries go through the common check.
            seen_sdf = True
            assert abs(ga.reshape(-1)[0] - gb.reshape(-1)[0]) < 0.1 * abs(gb.reshape(-1)[0]) + 1e-4
            ga, gb = ga.reshape(-1)[1:], gb.reshape(-1)[1:]
        # tensors whose gradient is below 1e-4 of the whole gradient (the 3-entry weight_g of the extra colour head, fed only
        # by the CLIP term) are compared against that floor, not against their own norm
        rel = ((ga - gb).double().norm() / (gb.double().norm() + 1e-4 * gnorm)).item()
        cos = torch.nn.functional.cosine_similarity(ga.reshape(1, -1).double(), gb.reshape(1, -1).double()).item()
        if gb.double().norm() < 1e-3 * gnorm:
            cos = 1.0
        worst = max(worst, rel)
        if rel >= 5e-2:
            d = (ga - gb).reshape(-1)
            k = d.abs().argmax().item()
            print("DIAG", n, "rel", rel, "argmax", k, ga.reshape(-1)[k].item(), gb.reshape(-1)[k].item(), "norms", ga.norm().item(), gb.norm().item(),
                  "without it", ((d.double().norm() ** 2 - d[k].double() ** 2).sqrt() / gb.double().norm()).item())
        assert rel < 5e-2 and cos > 0.995, (n, rel, cos)
    print("worst per-tensor relative gradient error", worst)
