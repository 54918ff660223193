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
import os

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


@gpu
@pytest.mark.parametrize("extra_color", [True, False])
def test_runner_train_neus_init_and_checkpoint_round_trip(tmp_path, extra_color):
    """Runner.train (main.py:180-256, BASELINE config 1) on a synthetic 4-image dataset in the reference's on-disk format
    (transforms_train.json + PNGs): the loss falls, the checkpoint has the reference's keys (main.py:622-629) and resumes."""
    import json
    from PIL import Image
    import bench
    from oracle import neus_oracle as O
    from avatarclip_amd.runner import Runner
    res = 32
    data = tmp_path / "data"
    (data / "img").mkdir(parents=True)
    frames = []
    yy, xx = np.mgrid[0:res, 0:res]
    for k in range(4):
        ang = 2 * np.pi * k / 4
        eye = np.array([1.5 * np.sin(ang), 0.2, 1.5 * np.cos(ang)])
        pose = O.lookat(eye, np.zeros(3), np.array([0., 1, 0]))
        disk = ((xx - res / 2 + 0.5) ** 2 + (yy - res / 2 + 0.5) ** 2) < (0.3 * res) ** 2     # a sphere seen from anywhere
        img = np.zeros((res, res, 3), np.uint8)
        img[disk] = (200, 150 + 20 * k, 100)
        Image.fromarray(img).save(str(data / "img" / ("%04d.png" % k)))
        frames.append({"file_path": "img/%04d" % k, "transform_matrix": np.asarray(pose).tolist()})
    with open(data / "transforms_train.json", "w") as fp:
        json.dump({"camera_angle_x": float(np.pi / 3), "frames": frames}, fp)
    conf = bench.make_conf(res, 32, small=True)
    conf.put("general.base_exp_dir", str(tmp_path / "exp"))
    conf.put("dataset.data_dir", str(data))
    conf.put("train.end_iter", 30)
    conf.put("train.batch_size", 256)
    conf.put("train.warm_up_end", 0)
    conf.put("train.save_freq", 30)
    # confs/base_models/*.conf (the NeuS-init stage) have no extra colour head; confs/examples/*.conf do
    conf.put("model.rendering_network.extra_color", extra_color)
    conf.put("model.neus_renderer.extra_color", extra_color)
    torch.manual_seed(0)
    np.random.seed(0)
    dev = torch.device("cuda")
    runner = Runner(None, mode="train", conf=conf, device=dev)
    assert runner.dataset.n_images == 4
    assert ("extra_lin.weight_v" in runner.color_network.state_dict()) == extra_color
    first = runner.train_iteration(runner.dataset.gen_random_rays_at(0, 256)).item()
    runner.train()
    last = runner.train_iteration(runner.dataset.gen_random_rays_at(0, 256)).item()
    print("NeuS-init loss", first, "->", last)
    assert np.isfinite(first) and np.isfinite(last) and last < first
    ck = tmp_path / "exp" / "checkpoints" / "ckpt_000030.pth"
    assert ck.exists()
    blob = torch.load(str(ck), map_location="cpu", weights_only=False)
    assert set(blob.keys()) == {"sdf_network_fine", "variance_network_fine", "color_network_fine", "optimizer", "iter_step"}
    assert any(k.endswith("weight_g") for k in blob["sdf_network_fine"]) and "variance" in blob["variance_network_fine"]
    r2 = Runner(None, mode="train", conf=conf, device=dev)
    r2.load_checkpoint("ckpt_000030.pth")
    assert r2.iter_step == 30
    sd_now = {k: v.clone() for k, v in blob["sdf_network_fine"].items()}
    for k, v in r2.sdf_network.state_dict().items():
        assert torch.equal(v.cpu(), sd_now[k].cpu())
    # --is_continue picks the newest checkpoint not past end_iter (main.py:160-170); train_clip's --pretrain path loads the
    # three networks of a NeuS-init checkpoint into the CLIP stage (main.py:611-620, strict=False for the colour head)
    r3 = Runner(None, mode="train", conf=conf, device=dev, is_continue=True)
    assert r3.iter_step == 30
    conf.put("train.pretrain", str(ck))
    r4 = Runner(None, mode="train_clip", conf=conf, device=dev)
    for k, v in r4.sdf_network.state_dict().items():
        assert torch.equal(v.cpu(), sd_now[k].cpu())
    # validate_image (main.py:741-820): one view rendered in ray chunks
    img = r4.validate_image(idx=0, resolution_level=2)
    assert img.shape == (16, 16, 3) and torch.isfinite(img).all() and img.min() >= 0 and img.max() <= 1


@gpu
def test_train_clip_iteration_with_silhouette_rays_and_background_augmentation():
    """the reference's default sampling mode (use_silhouettes = True, main.py:360-375): a ragged, per-iteration ray set inside the
    dilated SMPL silhouette, scattered back into the image for CLIP, with every background augmentation (main.py:387-415)"""
    import bench
    from avatarclip_amd.runner import Runner
    conf = bench.make_conf(256, 32, small=True)
    conf.put("train.use_silhouettes", True)
    conf.put("train.max_ray_num", 3000)
    conf.put("train.warm_up_end", 0)
    torch.manual_seed(0)
    np.random.seed(3)
    r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
    r.init_clip()
    r.init_smpl()
    r.update_learning_rate()
    seen = set()
    for i in range(8):
        loss = r.train_clip_iteration(i)
        assert torch.isfinite(loss), i
        seen.add(int(r.last_stats["rays"]))
        r.update_learning_rate()
    assert len(seen) > 1 and max(seen) <= 3000 * 1.3      # ragged ray counts that respect the budget
    assert all(torch.isfinite(p).all() for p in r.params_to_train)


@gpu
@pytest.mark.parametrize("flags", [dict(add_no_texture=False, texture_cast_light=False, use_face_prompt=False, use_back_prompt=False),
                                   dict(add_no_texture=False, texture_cast_light=True, use_face_prompt=True, use_back_prompt=False),
                                   dict(add_no_texture=True, texture_cast_light=False, use_face_prompt=False, use_back_prompt=True)])
def test_train_clip_ablation_switches(flags):
    """the switch combinations of confs/ablation/*.conf (main.py:426-534): every branch of the shading / CLIP-loss glue runs"""
    import bench
    from avatarclip_amd.runner import Runner
    conf = bench.make_conf(64, 32, small=True)
    for k, v in flags.items():
        conf.put("train." + k, v)
    conf.put("train.use_bg_aug", False)
    torch.manual_seed(0)
    np.random.seed(5)
    r = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
    r.init_clip()
    r.init_smpl()
    r.update_learning_rate()
    for i in range(5):          # i = 0 and 4 take the face-prompt camera when enabled (iter_i % 4 == 0)
        loss = r.train_clip_iteration(i)
        assert torch.isfinite(loss), (flags, i)
        r.update_learning_rate()
    assert all(torch.isfinite(p).all() for p in r.params_to_train)


@gpu
def test_cli_train_clip_from_a_conf_file(tmp_path):
    """`python -m avatarclip_amd.main --mode train_clip --conf X` (the reference's entry point, main.py:947-980) end to end:
    conf file on disk (reference syntax), the train_clip loop, report, checkpoint.  (`--case` is accepted and, as in the
    reference's Runner.__init__ (main.py:31-43), not substituted into the conf.)"""
    import subprocess
    import sys
    import bench
    conf = bench.make_conf(32, 32, small=True)
    # write the conf back out in the reference's HOCON style
    def dump(node, ind=0):
        out = []
        for k, v in node.items():
            if hasattr(v, "items"):
                out.append(" " * ind + "%s {" % k)
                out += dump(v, ind + 4)
                out.append(" " * ind + "}")
            else:
                out.append(" " * ind + "%s = %s" % (k, v))
        return out
    conf.put("general.base_exp_dir", str(tmp_path / "exp" / "smoke"))
    conf.put("train.end_iter", 3)
    conf.put("train.save_freq", 3)
    conf.put("train.report_freq", 1)
    conf.put("train.warm_up_end", 0)
    path = tmp_path / "case.conf"
    path.write_text("\n".join(dump(conf)) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "avatarclip_amd.main", "--mode", "train_clip", "--conf", str(path), "--case", "smoke"],
                       cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "loss" in p.stdout
    assert (tmp_path / "exp" / "smoke" / "checkpoints" / "ckpt_000003.pth").exists()
