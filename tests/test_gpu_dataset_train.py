"""Runner.train (main.py:180-256, the NeuS-init stage = BASELINE config 1) on the reference's OWN dataset: the 108 views of
data/zero_beta_standpose_render (tests/golden/smpl_views.npz, packed by oracle/gen_golden_views.py), written back to the
reference's on-disk layout (transforms_train.json + img/*.png), with the networks and loss weights of
confs/base_models/astrongman.conf.  Checks: the first iteration's loss equals the CPU oracle's on the same ray batch and
jitter; the loss falls over 60 iterations; the loss curve goes to gpurun_out/ (committed as profiles/r02_train_standpose_loss.json)."""
import json
import os

import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unpack_dataset(tag, dst):
    from PIL import Image
    z = np.load(os.path.join(ROOT, "tests", "golden", "smpl_views.npz"))
    os.makedirs(os.path.join(dst, "img"), exist_ok=True)
    frames = []
    for k, (im, pose) in enumerate(zip(z[tag + "_images"], z[tag + "_poses"])):
        Image.fromarray(np.repeat(im[..., None], 3, axis=2)).save(os.path.join(dst, "img", "%04d.png" % k))
        frames.append({"file_path": "img/%04d" % k, "transform_matrix": pose.tolist()})
    with open(os.path.join(dst, "transforms_train.json"), "w") as fp:
        json.dump({"camera_angle_x": float(z[tag + "_camera_angle_x"]), "frames": frames}, fp)
    return len(frames)


@gpu
def test_train_on_the_reference_standpose_dataset(tmp_path):
    import bench
    from oracle import neus_oracle as O
    from avatarclip_amd.runner import Runner
    n = unpack_dataset("stand", str(tmp_path / "data"))
    assert n == 108
    conf = bench.make_conf(256, 64, small=False)          # full-size nets = confs/base_models/astrongman.conf
    conf.put("general.base_exp_dir", str(tmp_path / "exp"))
    conf.put("dataset.data_dir", str(tmp_path / "data"))
    conf.put("train.batch_size", 5120)
    conf.put("train.mask_weight", 0.5)
    conf.put("train.warm_up_end", 0)      # the conf warms up over 5000 iterations (lr = 0 at step 0); a 60-step test starts at full lr
    conf.put("train.end_iter", 60)
    conf.put("model.rendering_network.extra_color", False)
    conf.put("model.neus_renderer.extra_color", False)
    dev = torch.device("cuda")
    r = Runner(None, mode="train", conf=conf, device=dev)
    assert r.dataset.n_images == 108 and r.dataset.H == 256 and abs(r.dataset.focal - 0.5 * 256 / np.tan(np.pi / 6)) < 1e-3
    # ---- the device-side batch look-up (device copy of the images, pinned index upload) = the reference's CPU look-up on the same draws
    from avatarclip_amd.dataset import SMPL_Dataset
    ds_cpu = SMPL_Dataset(conf["dataset"], device=torch.device("cpu"))
    for bs in (512, 5120):
        torch.manual_seed(11)
        a = r.dataset.gen_random_rays_at(torch.tensor(7), bs).cpu()
        torch.manual_seed(11)
        b = ds_cpu.gen_random_rays_at(torch.tensor(7), bs)
        assert torch.equal(a[:, 6:], b[:, 6:]) and (a[:, :6] - b[:, :6]).abs().max() < 1e-5
    # ---- first iteration vs the oracle on the same batch
    torch.manual_seed(3)
    data = r.dataset.gen_random_rays_at(58, 1024)
    jitter = torch.rand(1024, 1, generator=torch.Generator().manual_seed(5))
    sd_s = {k: v.detach().cpu().clone() for k, v in r.sdf_network.named_parameters()}
    sd_c = {k: v.detach().cpu().clone() for k, v in r.color_network.named_parameters()}
    var = r.deviation_network.variance.detach().cpu().clone()
    ren = r.renderer.render
    r.renderer.render = lambda *a, **k: ren(*a, jitter=jitter.to(dev), **k)
    r.update_learning_rate()
    loss0 = r.train_iteration(data).item()
    r.renderer.render = ren
    d = data.cpu()
    ro, rd, rgb, mask = d[:, :3], d[:, 3:6], d[:, 6:9], d[:, 9:10]
    near, far = O.near_far_from_sphere(ro, rd)
    out = O.render(sd_s, sd_c, var, ro, rd, near, far, 32, 32, 4, jitter, None, 1.0, extra_color=False)
    ref, closs, eik, mloss = O.neus_losses(out, rgb, (mask > 0.5).float(), 0.1, 0.5)
    print("first iteration: hip %.6f  oracle %.6f  (colour %.4f eikonal %.4f mask %.4f)" % (loss0, ref.item(), closs.item(), eik.item(), mloss.item()))
    assert abs(loss0 - ref.item()) < 2e-3 * max(1.0, abs(ref.item()))
    # ---- the reference's loop
    curve = [loss0]
    r.iter_step = 0
    image_perm = r.get_image_perm()
    for _ in range(60):
        batch = r.dataset.gen_random_rays_at(image_perm[r.iter_step % len(image_perm)], r.batch_size)
        curve.append(r.train_iteration(batch).item())
        r.update_learning_rate()
    print("loss curve (every 10th):", [round(c, 4) for c in curve[::10]])
    assert np.isfinite(curve).all() and np.mean(curve[-10:]) < 0.8 * np.mean(curve[1:6])
    # ---- Runner.train() itself: 5 more steps with the scalar log of main.py:230-238 (Loss/*, Statistics/s_val, cdf, weight_max, psnr)
    conf.put("train.end_iter", r.iter_step + 5)
    conf.put("train.report_freq", 5)
    r.end_iter, r.report_freq = r.iter_step + 5, 5
    r.train()
    rows = [json.loads(l) for l in open(os.path.join(str(tmp_path / "exp"), "logs", "scalars.jsonl"))]
    tags = {x["tag"] for x in rows}
    assert {"Loss/loss", "Loss/color_loss", "Loss/eikonal_loss", "Statistics/s_val", "Statistics/cdf", "Statistics/weight_max",
            "Statistics/psnr"} <= tags and len(rows) == 5 * 7
    psnr = [x["value"] for x in rows if x["tag"] == "Statistics/psnr"]
    sval = [x["value"] for x in rows if x["tag"] == "Statistics/s_val"]
    assert all(np.isfinite(psnr)) and all(0.0 < p < 60.0 for p in psnr) and all(0.0 < v < 1.0 for v in sval)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "r03_train_standpose_loss.json"), "w") as fp:
        json.dump({"what": "Runner.train on data/zero_beta_standpose_render (108 views), confs/base_models networks, batch 5120, lr 5e-4 "
                           "without warm-up, 60 iterations on the HIP path", "first_iteration_hip": loss0, "first_iteration_oracle": ref.item(),
                   "loss": curve}, fp)
