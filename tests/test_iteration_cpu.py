"""CPU check of the loop logic around the kernels: Runner.train_clip_iteration with its renderer / perceptor replaced by the
CPU oracle (so only the Runner's own orchestration is under test: view set-up, background, glue, loss assembly, prompt
switching, Adam, learning-rate schedule) against the independent oracle iteration (oracle/iteration_oracle.py).
Both legs are fp32 on the CPU: losses agree to 1e-5, weights after two steps to 1e-6."""
import numpy as np
import torch


def test_runner_loop_matches_independent_oracle_iteration_on_cpu():
    import bench
    from oracle import clip_vit_oracle as C
    from oracle import iteration_oracle as IT
    from oracle import neus_oracle as O
    from avatarclip_amd.runner import Runner, EllipsoidPrior, clip_vit_random_state_dict
    res, spp, iters = 16, 16, 2
    conf = bench.make_conf(res, spp, small=True)
    conf.put("train.use_bg_aug", False)
    conf.put("train.warm_up_end", 3)       # exercises the warm-up branch of the schedule (main.py:578-579)
    torch.manual_seed(0)
    cpu = torch.device("cpu")
    r = Runner(None, mode="train_clip", conf=conf, device=cpu)
    clip_sd = clip_vit_random_state_dict(0)

    class OraclePerceptor:
        def encode_image(self, x):
            return C.encode_image(clip_sd, x)
    r.init_clip(perceptor=OraclePerceptor())
    prior = EllipsoidPrior(device=cpu)
    r.init_smpl(prior)
    r.update_learning_rate()
    jit = {}

    def oracle_render(rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0):
        return O.render(dict(r.sdf_network.named_parameters()), dict(r.color_network.named_parameters()),
                        r.deviation_network.variance, rays_o, rays_d, near, far, spp // 2, spp // 2, 4, jit["t"], background_rgb,
                        cos_anneal_ratio)
    r.renderer.render = oracle_render
    sd_s = {n: p.detach().clone().requires_grad_() for n, p in r.sdf_network.named_parameters()}
    sd_c = {n: p.detach().clone().requires_grad_() for n, p in r.color_network.named_parameters()}
    var = r.deviation_network.variance.detach().clone().requires_grad_()
    st = IT.OracleState(sd_s, sd_c, var, lr0=r.learning_rate, alpha=r.learning_rate_alpha, warm_up_end=r.warm_up_end, end_iter=r.end_iter)
    texts = dict(prompt=r.encoded_text, face_prompt=r.encoded_face_text, back_prompt=r.encoded_back_text)
    oconf = dict(H=res, W=res, focal=r.dataset.focal, n_samples=spp // 2, n_importance=spp // 2, up_sample_steps=4,
                 full_frame_resolution_level=r.full_frame_resolution_level, mask_weight=r.mask_weight, igr_weight=r.igr_weight,
                 clip_weight=r.clip_weight, add_no_texture=True, texture_cast_light=True, use_face_prompt=True, use_back_prompt=True,
                 use_silhouettes=False, cos_anneal_ratio=1.0)
    cams = [(np.array([0.35, 0.25, 1.45], np.float32), np.array([0.02, -0.03, 0.01], np.float32), 0.3, 1.2, 1),
            (np.array([-1.1, 0.1, -0.9], np.float32), np.array([0.0, 0.05, 0.0], np.float32), 2.8, 4.0, 0)]
    for i in range(iters):
        jit["t"] = torch.rand(res * res, 1, generator=torch.Generator().manual_seed(100 + i))
        np.random.seed(99 + i)
        la = r.train_clip_iteration(i, camera=cams[i])
        r.update_learning_rate()
        rs = np.random.RandomState(99 + i)
        eye, at, theta, phi, is_front = cams[i]
        light = O.sphere_coord(theta + rs.uniform(-np.pi / 4, np.pi / 4), phi + rs.uniform(-np.pi / 4, np.pi / 4))
        dr = IT.Draws(eye=eye, at=at, theta=theta, phi=phi, is_front=is_front, prior_rgb=prior(eye, at), jitter=jit["t"],
                      light_dir=light, ambience=float(rs.uniform(0, 0.2)))
        ob = IT.train_clip_iteration(st, oconf, dr, clip_sd, texts, i)
        assert abs(la.item() - ob["loss"].item()) < 1e-5 * max(1.0, abs(ob["loss"].item())), (i, la.item(), ob["loss"].item())
        assert abs(r.optimizer.param_groups[0]["lr"] - st.opt.param_groups[0]["lr"]) < 1e-12
    for pa, pb in zip(r.params_to_train, st.params()):
        assert (pa.detach() - pb.detach()).abs().max() < 1e-6
