"""Trajectory parity: 300 consecutive `train_clip` iterations (main.py:345-566) on the HIP path against 300 iterations of
the INDEPENDENT CPU oracle (oracle/iteration_oracle.py: oracle render + shading + losses + ViT oracle + torch Adam), both legs fed the same
recorded draws -- cameras (main.py:348-359, drawn once with the reference's own sampler order), jitter, light directions, ambience, prior
renders.  One-step gates (tests/test_gpu_iteration.py) bound every tensor's gradient error per iteration; they cannot see an error that is
small per step and has the SAME sign on every step (the bf16 rounding of the weights in the gradient sweeps is the one
error that does not average out over the points).  Such a bias would show here as the two loss curves drifting apart.

The dynamics are chaotic (Adam divides by the gradient's own magnitude; the hierarchical sampler is discontinuous in the weights), so the two
weight vectors do NOT stay equal entry by entry -- after a few dozen steps they are two samples of the same optimisation.  What must agree:
  * the loss averaged over 20-iteration windows around iterations 50 / 100 / 200 / 290: |HIP - oracle| <= 3 % of the oracle's window mean
    (the windows see identical cameras, backgrounds and lights; measured: see the printed table and profiles/r06_trajectory.md);
  * both curves fall, and by the same amount: (first window - last window) agrees to 10 %;
  * a held-out view rendered from either leg's final weights by the SAME renderer (the oracle's): PSNR on the CLIP colours and on the silhouette
    (weight sum) >= 35 dB OR within 3 dB of what a SECOND oracle leg, started from weights moved by 1e-6, reaches against the first (the
    yardstick for "the same avatar": two fp32 runs of one implementation drift apart as well), and >= 30 dB in any case;
  * the HIP renderer on ITS final weights against the oracle renderer on the same weights: the one-step forward gate (5e-3) still holds at
    the end of the run (weights that have left the initialisation: inv_s has grown, the surface has sharpened).
Small nets (confs/examples_small), 32 x 32 full-frame rays, 16 + 16 samples per ray, lr warm-up off, 300 steps; the two CPU legs (~60 s each at 16 threads) run in a spawned process beside the HIP leg.
"""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu

N_ITERS = 300
WINDOWS = (50, 100, 200, 290)


def _psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


def _oracle_leg(p):
    """One oracle leg in a process of its own (spawned: legs A and B run side by side, beside the HIP leg).  Everything heavy is rebuilt here
    from seeds -- the stand-in CLIP weights, the procedural prior -- so that only the initial MLP weights and the recorded draws travel."""
    import os
    import time
    t_start = time.time()
    try:        # the two legs on DISJOINT cores (both would otherwise start their 16 threads on the same ones and halve each other)
        cores = sorted(os.sched_getaffinity(0))
        if len(cores) >= 64:
            mine = cores[16 + 24 * p["slot"]: 16 + 24 * p["slot"] + 16]     # (the first cores are left to the parent's launch thread; measured on the
                                                                             # 2 x 64-core box: 145 s per leg like this, 166 s on different sockets, 178 s unpinned, ~80 s alone)
            os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))     # torch's default oversubscribes the many small ops of this path on a 256-core host
    from oracle import iteration_oracle as IT
    from oracle import neus_oracle as O
    from avatarclip_amd.runner import clip_vit_random_state_dict, EllipsoidPrior
    clip_sd = clip_vit_random_state_dict(0)
    prior = EllipsoidPrior(device="cpu")
    g = torch.Generator().manual_seed(31)
    mv = lambda t: (t * (1 + p["perturb"] * torch.randn(t.shape, generator=g))).clone().requires_grad_()
    st = IT.OracleState({n: mv(t) for n, t in p["sdf"].items()}, {n: mv(t) for n, t in p["color"].items()}, mv(p["variance"]), lr0=p["lr0"],
                        alpha=p["alpha"], warm_up_end=p["warm_up_end"], end_iter=p["end_iter"])
    R = p["rays"]
    losses = []
    for i, (eye, at, theta, phi, is_front) in enumerate(p["cams"]):
        rs = np.random.RandomState(4321 + i)
        light = O.sphere_coord(theta + rs.uniform(-np.pi / 4, np.pi / 4), phi + rs.uniform(-np.pi / 4, np.pi / 4))
        amb = float(rs.uniform(0, 0.2))
        dr = IT.Draws(eye=eye, at=at, theta=theta, phi=phi, is_front=is_front, prior_rgb=prior(eye, at),
                      jitter=torch.rand(R, 1, generator=torch.Generator().manual_seed(100 + i)), choice_i=3, light_dir=light, ambience=amb)
        losses.append(float(IT.train_clip_iteration(st, p["oconf"], dr, clip_sd, p["texts"], i)["loss"]))
    print("oracle leg %d: %.1f s" % (p["slot"], time.time() - t_start), flush=True)
    return dict(losses=np.asarray(losses), seconds=time.time() - t_start, sdf={k: t.detach() for k, t in st.sdf.items()}, color={k: t.detach() for k, t in st.color.items()},
                variance=st.variance.detach(), lr=st.opt.param_groups[0]["lr"], iter_step=st.iter_step)


@gpu
def test_300_iterations_track_the_independent_oracle():
    from oracle import iteration_oracle as IT
    from oracle import neus_oracle as O
    from avatarclip_amd.runner import clip_vit_random_state_dict, EllipsoidPrior
    from tests.test_gpu_iteration import _make_runner, _oracle_conf
    res, spp = 32, 32
    dev = torch.device("cuda")
    clip_sd = clip_vit_random_state_dict(0)
    a = _make_runner(dev, res, spp, True)
    a.init_clip(clip_state_dict=clip_sd)
    prior = EllipsoidPrior(device="cpu")
    np.random.seed(2024)
    cams = [a.sample_camera(i) for i in range(N_ITERS)]            # the reference's sampler and draw order (face views every 4th iteration)
    priors = {}

    def prior_of(i):
        if i not in priors:
            priors[i] = prior(cams[i][0], cams[i][1])
        return priors[i]
    step = {"i": 0}
    a.init_smpl(prior_renderer=lambda eye, at: prior_of(step["i"]).to(dev))
    a.update_learning_rate()
    init = ({n: p.detach().cpu().clone() for n, p in a.sdf_network.named_parameters()}, {n: p.detach().cpu().clone() for n, p in a.color_network.named_parameters()},
            a.deviation_network.variance.detach().cpu().clone())

    texts = dict(prompt=a.encoded_text.cpu(), face_prompt=a.encoded_face_text.cpu(), back_prompt=a.encoded_back_text.cpu())
    oconf = _oracle_conf(a, a.dataset.H)
    R = res * res
    jitter = lambda i: torch.rand(R, 1, generator=torch.Generator().manual_seed(100 + i))
    # ---------------- oracle legs (CPU), started first: two spawned processes beside the HIP leg.  A: from the product's initial weights.  B: the SAME
    # oracle from weights moved by 1e-6 (relative): the optimisation is chaotic, so two fp32 runs of one implementation drift apart too, and B measures
    # by how much -- the yardstick for "the HIP leg arrives at the same avatar" below.
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    payload = dict(sdf=init[0], color=init[1], variance=init[2], lr0=a.learning_rate, alpha=a.learning_rate_alpha, warm_up_end=a.warm_up_end,
                   end_iter=a.end_iter, rays=R, cams=cams, oconf=oconf, texts=texts)
    pool = ProcessPoolExecutor(1, mp_context=mp.get_context("spawn"))     # ONE worker: two legs side by side halve each other on the 2 x 64-core box (39 s per 100
                                                                          # iterations each against 20 s alone, whatever the core placement): one after the other, beside the HIP leg
    legs = [pool.submit(_oracle_leg, dict(payload, perturb=pt, slot=k)) for k, pt in enumerate((0.0, 1e-6))]
    # ---------------- HIP leg
    a_render = a.renderer.render
    jit = {}
    a.renderer.render = lambda *args, **kw: a_render(*args, jitter=jit["t"], **kw)
    loss_hip = []
    for i in range(N_ITERS):
        step["i"] = i
        jit["t"] = jitter(i).to(dev)
        np.random.seed(4321 + i)                       # the product draws: light angles (2), ambience (1)  (main.py:433,440)
        loss_hip.append(a.train_clip_iteration(i, camera=cams[i]))
        a.update_learning_rate()
    loss_hip = torch.stack(loss_hip).cpu().double().numpy()
    assert np.isfinite(loss_hip).all()
    leg_a, leg_b = [f.result(timeout=1500) for f in legs]
    pool.shutdown()
    loss_or, loss_or_b = leg_a["losses"], leg_b["losses"]
    print("oracle legs: %.1f s / %.1f s" % (leg_a["seconds"], leg_b["seconds"]))
    assert abs(a.optimizer.param_groups[0]["lr"] - leg_a["lr"]) < 1e-12 and a.iter_step == leg_a["iter_step"] == N_ITERS
    # ---------------- the curves
    print("iter   hip      oracle   (single iterations)")
    for i in (0, 1, 2, 5, 10, 20, 50, 100, 200, 299):
        print("%4d  %8.5f %8.5f" % (i, loss_hip[i], loss_or[i]))
    assert abs(loss_hip[0] - loss_or[0]) < 2e-3 * max(1.0, abs(loss_or[0]))          # the one-step gate, for reference
    win = lambda x, c: float(np.mean(x[max(0, c - 10):c + 10]))
    print("window  hip      oracle A oracle B   |hip - A| / A   |B - A| / A")
    worst = 0.0
    for c in (10,) + WINDOWS:
        h, o, ob = win(loss_hip, c), win(loss_or, c), win(loss_or_b, c)
        worst = max(worst, abs(h - o) / abs(o))
        print("%4d   %8.5f %8.5f %8.5f   %.4f          %.4f" % (c, h, o, ob, abs(h - o) / abs(o), abs(ob - o) / abs(o)))
        assert abs(h - o) <= 0.03 * abs(o), (c, h, o)
    drop_h, drop_o = win(loss_hip, 10) - win(loss_hip, 290), win(loss_or, 10) - win(loss_or, 290)
    print("drop first -> last window: hip %.5f oracle %.5f; worst window difference %.4f" % (drop_h, drop_o, worst))
    assert drop_o > 0 and drop_h > 0 and abs(drop_h - drop_o) <= 0.10 * drop_o
    rms = float(np.sqrt(np.mean((loss_hip - loss_or) ** 2)) / np.mean(np.abs(loss_or)))
    print("rms per-iteration loss difference / mean loss: %.4f" % rms)
    # ---------------- the avatars the two legs arrived at, held-out view, the SAME (oracle) renderer on both weight sets
    eye, at = np.array([0.55, 0.15, 1.35], np.float32), np.array([0.0, 0.02, 0.0], np.float32)
    pose = torch.from_numpy(O.lookat(eye, at, np.array([0., 1, 0]))).float()
    assert a.dataset.W == res
    o, v = O.gen_rays_pose(pose, res, res, a.dataset.focal)
    ro, rd = o.reshape(-1, 3).contiguous(), v.reshape(-1, 3).contiguous()
    near, far = O.near_far_from_sphere(ro, rd)
    jt = jitter(9999)
    bg = torch.zeros(1, 3)
    hip_s = {n: p.detach().cpu() for n, p in a.sdf_network.named_parameters()}
    hip_c = {n: p.detach().cpu() for n, p in a.color_network.named_parameters()}
    hip_v = a.deviation_network.variance.detach().cpu()
    r_or = O.render(leg_a["sdf"], leg_a["color"], leg_a["variance"], ro, rd, near, far, spp // 2, spp // 2, 4, jt, bg, 1.0)
    r_hw = O.render(hip_s, hip_c, hip_v, ro, rd, near, far, spp // 2, spp // 2, 4, jt, bg, 1.0)
    r_ob = O.render(leg_b["sdf"], leg_b["color"], leg_b["variance"], ro, rd, near, far, spp // 2, spp // 2, 4, jt, bg, 1.0)
    p_col = _psnr(r_hw["extra_color_fine"].detach(), r_or["extra_color_fine"].detach())
    p_sil = _psnr(r_hw["weight_sum"].detach(), r_or["weight_sum"].detach())
    q_col = _psnr(r_ob["extra_color_fine"].detach(), r_or["extra_color_fine"].detach())
    q_sil = _psnr(r_ob["weight_sum"].detach(), r_or["weight_sum"].detach())
    print("held-out view under ONE renderer (the oracle's), final weights:  HIP vs oracle A: colour %.2f dB, silhouette %.2f dB   |   oracle B vs oracle A (the "
          "optimisation's own chaos, start moved by 1e-6): colour %.2f dB, silhouette %.2f dB;   inv_s %.3f / %.3f / %.3f"
          % (p_col, p_sil, q_col, q_sil, float(torch.exp(hip_v * 10)), float(torch.exp(leg_a["variance"] * 10)), float(torch.exp(leg_b["variance"] * 10))))
    # the HIP leg is as close to the oracle as the oracle is to itself (3 dB of slack), and never worse than 30 dB; two runs measured 38.9 / 36.3 and
    # 36.5 / 33.9 dB against oracle legs that differed only in their thread count (profiles/r06_trajectory.md)
    assert p_col >= min(35.0, q_col - 3.0) and p_sil >= min(35.0, q_sil - 3.0) and min(p_col, p_sil) >= 30.0
    # the HIP renderer on its own final weights vs the oracle renderer on the same weights and depths
    out = a_render(ro.to(dev), rd.to(dev), near.to(dev), far.to(dev), background_rgb=bg.to(dev), cos_anneal_ratio=1.0,
                   z_vals=r_hw["z_vals"].detach().to(dev))
    e = (out["extra_color_fine"].detach().cpu() - r_hw["extra_color_fine"].detach()).abs().max().item()
    print("HIP vs oracle renderer on the final weights: max |rgb diff| %.3e" % e)
    assert e < 5e-3
