"""Pins oracle/neus_oracle.py (the CPU restatement) against fixtures produced by running the reference's own
modules (oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import neus_oracle as O
from tests.helpers import load_case, relerr, GOLDEN
import os


@pytest.mark.parametrize("name,has_jitter", [("neus_small.npz", False), ("neus_full.npz", True), ("neus_full_128.npz", True)])
def test_render_matches_reference(name, has_jitter):
    """neus_full_128.npz (oracle/gen_golden_128.py): BASELINE config 3's regime -- 64 + 64 samples, steps of 16, per-ray grey background"""
    rec, sd_sdf, sd_col, variance = load_case(name)
    n0, m = rec["up0_z_in"].shape[1], rec["up0_new_z"].shape[1]
    jitter = rec["jitter"] if has_jitter else None
    bg = rec["bg"] if rec["bg"].numel() else None
    # the up-sampling chain is chaotic in fp32 (1e-7 sdf noise moves inverse-CDF samples), so every step is
    # checked from the reference's own inputs of that step
    for i in range(4):
        z_in, sdf_in = rec["up%d_z_in" % i], rec["up%d_sdf_in" % i]
        new_z = O.up_sample(rec["rays_o"], rec["rays_d"], z_in, sdf_in, m, 64 * 2 ** i)
        assert torch.allclose(new_z, rec["up%d_new_z" % i], atol=1e-6), i
        z2, sdf2 = O.cat_z_vals(sd_sdf, rec["rays_o"], rec["rays_d"], z_in, rec["up%d_new_z" % i], sdf_in, last=(i == 3))
        if i < 3:
            assert torch.equal(z2, rec["up%d_z_in" % (i + 1)])
            assert torch.allclose(sdf2, rec["up%d_sdf_in" % (i + 1)], atol=5e-6)
        else:
            assert torch.equal(z2, rec["z_final"])
    z0 = O.coarse_z_vals(rec["near"], rec["far"], n0, jitter)
    assert torch.allclose(z0, rec["up0_z_in"], atol=1e-6)
    # render on the reference's z so the comparison is not chaotic in z
    out = O.render(sd_sdf, sd_col, variance, rec["rays_o"], rec["rays_d"], rec["near"], rec["far"], n0, n0,
                   background_rgb=bg, cos_anneal_ratio=float(rec["cos_anneal"]), z_vals=rec["z_final"])
    for k in ("color_fine", "extra_color_fine", "weight_sum", "weight_max", "weights", "mid_z_vals",
              "inside_sphere", "cdf_fine", "s_val"):
        assert torch.allclose(out[k].detach(), rec["out_" + k], atol=3e-5), k
    assert torch.allclose(out["gradients"].detach(), rec["out_gradients"], atol=2e-4, rtol=1e-4)
    assert abs(out["gradient_error"].item() - rec["out_gradient_error"].item()) < 1e-5


@pytest.mark.parametrize("name", ["neus_small.npz", "neus_full.npz", "neus_full_128.npz"])
def test_parameter_gradients_match_reference(name):
    from oracle.gen_golden import scalar_loss
    rec, sd_sdf, sd_col, variance = load_case(name)
    bg = rec["bg"] if rec["bg"].numel() else None
    leaves = {}
    def req(sd, pfx):
        out = {}
        for k, v in sd.items():
            out[k] = v.clone().requires_grad_(True)
            leaves[pfx + k] = out[k]
        return out
    s, c = req(sd_sdf, "sdf."), req(sd_col, "col.")
    var = variance.clone().requires_grad_(True)
    leaves["var.variance"] = var
    n0 = rec["up0_z_in"].shape[1]
    out = O.render(s, c, var, rec["rays_o"], rec["rays_d"], rec["near"], rec["far"], n0, n0, background_rgb=bg,
                   cos_anneal_ratio=float(rec["cos_anneal"]), z_vals=rec["z_final"])
    coef = {k[5:]: v for k, v in rec.items() if k.startswith("coef_")}
    loss = scalar_loss(out, coef)
    assert abs(loss.item() - rec["loss"].item()) < 1e-3 * max(1.0, abs(rec["loss"].item()))
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    for n, g in zip(names, grads):
        ref = rec["grad_" + n]
        if g is None:
            assert ref.abs().max() == 0
            continue
        assert relerr(g, ref) < 2e-3 or (g - ref).abs().max() < 1e-5, (n, relerr(g, ref))


def test_sample_pdf_kat():
    z = np.load(os.path.join(GOLDEN, "sampling_kat.npz"))
    s = O.sample_pdf(torch.from_numpy(z["bins"]), torch.from_numpy(z["weights"]), 8)
    assert torch.allclose(s, torch.from_numpy(z["samples"]), atol=1e-6)


def test_rays_and_cameras():
    z = np.load(os.path.join(GOLDEN, "rays_cam.npz"))
    pose = torch.from_numpy(z["pose58"])
    for lvl in (4, 2.25):
        o, v = O.gen_rays_pose(pose, 256, 256, float(z["focal"]), lvl)
        assert np.allclose(o.numpy(), z["rays_o_l%s" % lvl], atol=1e-6)
        assert np.allclose(v.numpy(), z["rays_v_l%s" % lvl], atol=1e-6)
    o, v = O.gen_rays_pose(pose, 256, 256, float(z["focal"]), 4)
    near, far = O.near_far_from_sphere(o.reshape(-1, 3), v.reshape(-1, 3))
    assert np.allclose(near.numpy(), z["near_l4"], atol=1e-6) and np.allclose(far.numpy(), z["far_l4"], atol=1e-6)
    for eye, at, pose in zip(z["cam_eye"], z["cam_at"], z["cam_pose"]):
        assert np.allclose(O.lookat(eye, at, np.array([0, 1, 0])), pose, atol=1e-6)
    for i in range(6):
        assert np.allclose(O.sphere_coord(0.3 * i, 0.7 * i), z["sphere_coord"][i])


def test_gen_rays_silhouettes_matches_reference_golden():
    """SMPL_Dataset.gen_rays_silhouettes (dataset.py:252-275) with the dilation done as a 21x21 max filter on the
    device instead of scipy on the host: same rays, same grid size, same mask as the reference's own function."""
    import os
    from avatarclip_amd.dataset import SMPL_Dataset
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "rays_sil.npz"))
    ds = SMPL_Dataset(None, device=torch.device("cpu"), load_images=False)
    assert abs(ds.focal - float(z["focal"])) < 1e-9
    pose = torch.from_numpy(z["pose"])
    for i in range(2):
        o, v, Wn, sel = ds.gen_rays_silhouettes(pose, int(z["max_ray_num%d" % i]), torch.from_numpy(z["mask%d" % i]))
        assert Wn == int(z["W%d" % i])
        assert torch.equal(sel.cpu(), torch.from_numpy(z["sel%d" % i]))
        assert o.shape == z["rays_o%d" % i].shape
        assert np.allclose(o.numpy(), z["rays_o%d" % i], atol=1e-6) and np.allclose(v.numpy(), z["rays_v%d" % i], atol=1e-6)
    # empty mask: the reference falls back to the level-4 full frame
    o, v = ds.gen_rays_silhouettes(pose, 1000, torch.zeros(256, 256))
    assert o.shape == (64, 64, 3)
